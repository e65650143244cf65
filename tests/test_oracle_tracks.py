"""The tracks oracle against a literal Python transcription of the reference's two functions
(opensfm/unionfind.py:67-103 and opensfm/tracking.py:82-98,238-244), which cannot be imported here
(opensfm.tracking needs pymap and networkx)."""
import numpy as np


class _UnionFind:  # opensfm/unionfind.py:67-103, transcribed
    def __init__(self):
        self.weights, self.parents = {}, {}

    def __getitem__(self, obj):
        if obj not in self.parents:
            self.parents[obj] = obj
            self.weights[obj] = 1
            return obj
        path = [obj]
        root = self.parents[obj]
        while root != path[-1]:
            path.append(root)
            root = self.parents[root]
        for ancestor in path:
            self.parents[ancestor] = root
        return root

    def __iter__(self):
        return iter(self.parents)

    def union(self, *objects):
        roots = [self[x] for x in objects]
        heaviest = max((self.weights[r], r) for r in roots)[1]
        for r in roots:
            if r != heaviest:
                self.weights[heaviest] += self.weights[r]
                self.parents[r] = heaviest


def reference_tracks(matches, min_length):  # tracking.py:82-98 + _good_track (238-244)
    uf = _UnionFind()
    for im1, im2 in matches:
        for f1, f2 in matches[im1, im2]:
            uf.union((im1, f1), (im2, f2))
    sets = {}
    for i in uf:
        p = uf[i]
        sets.setdefault(p, []).append(i)

    def good(track):
        if len(track) < min_length:
            return False
        ims = [f[0] for f in track]
        return len(ims) == len(set(ims))

    return [t for t in sets.values() if good(t)]


def random_match_graph(rng, n_images, n_feat, n_pairs, per_pair, p_conflict=0.1):
    matches = {}
    for _ in range(n_pairs):
        a, b = sorted(rng.choice(n_images, 2, replace=False))
        if (a, b) in matches:
            continue
        f1 = rng.choice(n_feat, per_pair, replace=False)
        # mostly "same scene point" links (f -> f) with some conflicting links that create bad tracks
        f2 = np.where(rng.random(per_pair) < p_conflict, rng.integers(0, n_feat, per_pair), f1)
        matches[int(a), int(b)] = [(int(x), int(y)) for x, y in zip(f1, f2)]
    return matches


def to_edges(matches, n_images, n_feat):
    off = np.arange(n_images + 1, dtype=np.int64) * n_feat
    ea = [off[a] + f1 for (a, b), m in matches.items() for f1, f2 in m]
    eb = [off[b] + f2 for (a, b), m in matches.items() for f1, f2 in m]
    return np.array(ea, np.int32), np.array(eb, np.int32), off


def check_equal(ref, got):
    nt, ot, oi, of = got
    assert nt == len(ref)
    flat = [(k, im, f) for k, t in enumerate(ref) for im, f in t]
    assert len(flat) == len(ot)
    assert np.array_equal(np.array(flat, np.int64).reshape(-1, 3), np.stack([ot, oi, of], axis=1))


def test_tracks_oracle_equals_reference_transcription(oracle_lib):
    rng = np.random.default_rng(3)
    for n_images, n_feat, n_pairs, per_pair, ml in ((6, 40, 10, 15, 2), (12, 100, 40, 30, 2), (20, 60, 120, 25, 3), (5, 10, 9, 8, 2)):
        matches = random_match_graph(rng, n_images, n_feat, n_pairs, per_pair)
        ea, eb, off = to_edges(matches, n_images, n_feat)
        check_equal(reference_tracks(matches, ml), oracle_lib.tracks(ea, eb, off, ml))


def test_tracks_oracle_edge_cases(oracle_lib):
    off = np.array([0, 5, 10, 15], np.int64)
    nt, ot, oi, of = oracle_lib.tracks(np.zeros(0, np.int32), np.zeros(0, np.int32), off, 2)
    assert nt == 0 and len(ot) == 0
    # a chain 0:1 - 1:2 - 2:3 is one track of length 3; adding 0:4 - 2:3 puts image 0 twice -> dropped
    nt, ot, oi, of = oracle_lib.tracks(np.array([1, 7], np.int32), np.array([7, 13], np.int32), off, 2)
    assert nt == 1 and list(oi) == [0, 1, 2] and list(of) == [1, 2, 3]
    nt, *_ = oracle_lib.tracks(np.array([1, 7, 4], np.int32), np.array([7, 13, 13], np.int32), off, 2)
    assert nt == 0
    nt, *_ = oracle_lib.tracks(np.array([1], np.int32), np.array([7], np.int32), off, 3)
    assert nt == 0
