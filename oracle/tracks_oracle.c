/* tracks_oracle.c -- CPU restatement of the grouping in tracking.create_tracks_manager.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Follows opensfm/tracking.py:82-98 and
 * opensfm/unionfind.py:67-103:
 *   uf = UnionFind(); for (im1, im2) in matches: for (f1, f2) in matches[im1, im2]: uf.union((im1,f1),(im2,f2))
 *   sets keyed by root, filled by iterating uf (a dict: insertion order), so the sets are listed in the
 *   order of their first-inserted member and every set lists its members in insertion order;
 *   tracks = [t for t in sets.values() if _good_track(t, min_length)]   (tracking.py:238-244:
 *   len(t) >= min_length and no image twice);  track_id = index in that list.
 * Which member ends up being the root (weights, tuple tie-break) does not influence the output.
 * Parity: pinned by tests/test_oracle_tracks.py against a literal Python transcription of the two
 * reference functions run on random match graphs (the reference itself needs pymap / networkx).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int find_root(int32_t *parent, int v) {
  int r = v;
  while (parent[r] != r) r = parent[r];
  while (parent[v] != r) {
    const int n = parent[v];
    parent[v] = r;
    v = n;
  }
  return r;
}

/* edges: node ids (node = node_offsets[image] + feature) in the reference's union order.
 * Outputs (capacity = number of nodes): obs_track / obs_image / obs_feature in (track, insertion) order.
 * Returns the number of observations; *n_tracks_out = number of tracks. */
int64_t oracle_tracks(const int32_t *ea, const int32_t *eb, int64_t n_edges, const int64_t *node_offsets, int32_t n_images,
                      int32_t min_length, int32_t *obs_track, int32_t *obs_image, int32_t *obs_feature, int64_t *n_tracks_out) {
  const int64_t N = node_offsets[n_images];
  int32_t *parent = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N + 1));
  int32_t *weight = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N + 1));
  int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N + 1)); /* insertion order -> node */
  int32_t *image = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N + 1));
  int64_t n_ins = 0;
  for (int64_t v = 0; v < N; v++) parent[v] = -1;
  {
    int im = 0;
    for (int64_t v = 0; v < N; v++) {
      while (im + 1 < n_images && node_offsets[im + 1] <= v) im++;
      image[v] = im;
    }
  }
  for (int64_t e = 0; e < n_edges; e++) {
    const int pair[2] = {ea[e], eb[e]};
    int roots[2];
    for (int k = 0; k < 2; k++) { /* self[x]: unknown objects become singletons, in argument order */
      if (parent[pair[k]] < 0) {
        parent[pair[k]] = pair[k];
        weight[pair[k]] = 1;
        order[n_ins++] = pair[k];
      }
      roots[k] = find_root(parent, pair[k]);
    }
    if (roots[0] != roots[1]) { /* heaviest = max((weight, root)); the tie-break does not matter for the sets */
      const int h = weight[roots[0]] >= weight[roots[1]] ? roots[0] : roots[1];
      const int l = h == roots[0] ? roots[1] : roots[0];
      weight[h] += weight[l];
      parent[l] = h;
    }
  }
  /* sets in order of first-inserted member, members in insertion order */
  int32_t *set_of_root = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N + 1));
  int32_t *set_len = (int32_t *)calloc((size_t)(n_ins + 1), sizeof(int32_t));
  int32_t *member_set = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_ins + 1));
  for (int64_t v = 0; v < N; v++) set_of_root[v] = -1;
  int64_t n_sets = 0;
  for (int64_t k = 0; k < n_ins; k++) {
    const int r = find_root(parent, order[k]);
    if (set_of_root[r] < 0) set_of_root[r] = (int32_t)n_sets++;
    member_set[k] = set_of_root[r];
    set_len[member_set[k]]++;
  }
  /* _good_track: length and no repeated image */
  uint8_t *bad = (uint8_t *)calloc((size_t)(n_sets + 1), 1);
  int64_t *start = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_sets + 2));
  start[0] = 0;
  for (int64_t s = 0; s < n_sets; s++) start[s + 1] = start[s] + set_len[s];
  int32_t *members = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_ins + 1));
  int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_sets + 1));
  memcpy(fill, start, sizeof(int64_t) * (size_t)n_sets);
  for (int64_t k = 0; k < n_ins; k++) members[fill[member_set[k]]++] = order[k];
  int32_t *seen = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n_images + 1));
  for (int i = 0; i < n_images; i++) seen[i] = -1;
  for (int64_t s = 0; s < n_sets; s++) {
    if (set_len[s] < min_length) bad[s] = 1;
    for (int64_t q = start[s]; q < start[s + 1]; q++) {
      const int im = image[members[q]];
      if (seen[im] == (int32_t)s) bad[s] = 1;
      seen[im] = (int32_t)s;
    }
  }
  int64_t n_tracks = 0, n_obs = 0;
  for (int64_t s = 0; s < n_sets; s++) {
    if (bad[s]) continue;
    for (int64_t q = start[s]; q < start[s + 1]; q++) {
      const int v = members[q];
      obs_track[n_obs] = (int32_t)n_tracks;
      obs_image[n_obs] = image[v];
      obs_feature[n_obs] = (int32_t)(v - node_offsets[image[v]]);
      n_obs++;
    }
    n_tracks++;
  }
  *n_tracks_out = n_tracks;
  free(parent); free(weight); free(order); free(image); free(set_of_root); free(set_len); free(member_set);
  free(bad); free(start); free(members); free(fill); free(seen);
  return n_obs;
}
