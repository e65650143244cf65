"""The bag-of-words side of matching on the MI355X: ``match_words`` / ``match_words_symmetric`` (``opensfm/matching.py:637-680`` over
``pyfeatures.match_using_words``, ``opensfm/src/features/src/matching.cc:24-88``), batched over a pair list, and the VLAD descriptor /
distance calls of ``opensfm/vlad.py`` (``pyfeatures.compute_vlad_descriptor`` / ``compute_vlad_distances``, ``matching.cc:93-152``).
Thin ctypes glue over ``csrc/words.hip``; there is no CPU fallback."""
import ctypes as C
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import check, default_context


def _fp(a: np.ndarray, t):
    return a.ctypes.data_as(C.POINTER(t))


class WordsStore:
    """Descriptors (float32, 128-D) and the closest vocabulary words of every feature of a set of images, resident in HBM."""

    def __init__(self, descriptors: Sequence[np.ndarray], words: Sequence[np.ndarray], ctx=None):
        if len(descriptors) != len(words):
            raise ValueError("one word array per descriptor array")
        self.ctx = ctx or default_context()
        counts = np.array([len(d) for d in descriptors], np.int32)
        nw = None
        for d, w in zip(descriptors, words):
            w = np.asarray(w)
            if len(w) != len(d):
                raise ValueError("words and descriptors of an image differ in length")
            if len(d):
                k = w.reshape(len(d), -1).shape[1]
                if nw is not None and k != nw:
                    raise ValueError("images carry different numbers of words per feature")
                nw = k
        nw = nw or 1
        desc = np.ascontiguousarray(np.concatenate([np.asarray(d, np.float32).reshape(-1, 128) for d in descriptors] or [np.zeros((0, 128), np.float32)]))
        wd = np.ascontiguousarray(np.concatenate([np.asarray(w, np.int32).reshape(len(d), -1) if len(d) else np.zeros((0, nw), np.int32)
                                                  for d, w in zip(descriptors, words)] or [np.zeros((0, nw), np.int32)]))
        h = C.c_void_p()
        check(_lib.load().osfm_words_store_create(self.ctx.handle, len(counts), _fp(counts, C.c_int32), 128, nw, _fp(desc, C.c_float),
                                                  _fp(wd, C.c_int32), C.byref(h)), "osfm_words_store_create")
        self.handle = h
        self.n_images = len(counts)
        self.counts = counts
        self.max_count = int(_lib.load().osfm_words_store_max_count(h))

    def close(self):
        if getattr(self, "handle", None):
            _lib.load().osfm_words_store_destroy(self.handle)
            self.handle = None

    __del__ = close


def match_words_pairs(store: WordsStore, pairs, config: Optional[Dict[str, Any]] = None, symmetric: bool = True) -> Tuple[List[np.ndarray], float]:
    """``match_words[_symmetric]`` for every pair (i, j) of image indices of ``store``: per pair an (m, 2) int32 array of feature index
    pairs ordered by the first image's feature; and the kernel milliseconds."""
    cfg = config or {}
    ratio, checks = float(cfg.get("lowes_ratio", 0.8)), int(cfg.get("bow_num_checks", 20))
    pairs = np.ascontiguousarray(np.asarray(pairs, np.int32).reshape(-1, 2))
    n = len(pairs)
    counts = np.zeros(max(n, 1), np.int32)
    matches = np.zeros((max(n, 1), store.max_count, 2), np.int32)
    ms = C.c_double(0.0)
    check(_lib.load().osfm_match_words_pairs(store.ctx.handle, store.handle, _fp(pairs, C.c_int32), n, C.c_float(ratio), checks, int(bool(symmetric)),
                                             _fp(counts, C.c_int32), _fp(matches, C.c_int32), C.byref(ms)), "osfm_match_words_pairs")
    return [matches[p, : counts[p]].copy() for p in range(n)], float(ms.value)


def match_words(f1: np.ndarray, words1: np.ndarray, f2: np.ndarray, words2: np.ndarray, config: Dict[str, Any]) -> np.ndarray:
    """Drop-in for ``matching.match_words`` (matching.py:637-656): (m, 2) array of (index in f1, index in f2)."""
    store = WordsStore([f1, f2], [words1, np.asarray(words2).reshape(len(f2), -1)])
    try:
        return match_words_pairs(store, [(0, 1)], config, symmetric=False)[0][0]
    finally:
        store.close()


def match_words_symmetric(f1: np.ndarray, words1: np.ndarray, f2: np.ndarray, words2: np.ndarray, config: Dict[str, Any]) -> List[Tuple[int, int]]:
    """Drop-in for ``matching.match_words_symmetric`` (matching.py:659-680).  The reference returns ``list(set & set)``, i.e. no
    defined order; here the pairs come ordered by the first index."""
    store = WordsStore([f1, f2], [words1, words2])
    try:
        return [(int(a), int(b)) for a, b in match_words_pairs(store, [(0, 1)], config, symmetric=True)[0][0]]
    finally:
        store.close()


# ---- VLAD (opensfm/vlad.py) ----
def unnormalized_vlad(features: np.ndarray, centers: np.ndarray, ctx=None) -> Optional[np.ndarray]:
    """``pyfeatures.compute_vlad_descriptor`` as ``vlad.unnormalized_vlad`` calls it (vlad.py:27-41)."""
    if np.ndim(centers) != 2 or np.ndim(features) != 2 or centers.shape[1] != features.shape[1] or centers.dtype != features.dtype:
        return None  # vlad.unnormalized_vlad's own guard (vlad.py:17-21)
    ctx = ctx or default_context()
    features = np.ascontiguousarray(features, np.float32)
    centers = np.ascontiguousarray(centers, np.float32)
    if centers.size == 0:
        raise RuntimeError("Zero VLAD centers or zero length VLAD words.")
    out = np.zeros(centers.size, np.float32)
    check(_lib.load().osfm_vlad_descriptor(ctx.handle, _fp(features, C.c_float), len(features), _fp(centers, C.c_float), len(centers),
                                           centers.shape[1], _fp(out, C.c_float)), "osfm_vlad_descriptor")
    return out


def signed_square_root_normalize(v: np.ndarray) -> np.ndarray:
    """``vlad.signed_square_root_normalize`` (vlad.py:44-54): host arithmetic on one 8192-vector, kept in numpy as in the reference."""
    v = np.sign(v) * np.sqrt(np.abs(v))
    v /= np.linalg.norm(v)
    return v


def vlad_distances(image: str, other_images: Iterable[str], histograms: Dict[str, np.ndarray], ctx=None) -> Tuple[str, List[float], List[str]]:
    """``vlad.vlad_distances`` (vlad.py:57-72 over compute_vlad_distances, matching.cc:126-152): L2 distances between the descriptor of
    ``image`` and those of the other images that have one (``image`` itself skipped), in sorted-name order (the reference iterates
    a std::set)."""
    if image not in histograms:
        return image, [], []
    others = [o for o in sorted(set(other_images)) if o != image and o in histograms]  # the reference walks a std::set: sorted names
    if not others:
        return image, [], []
    ctx = ctx or default_context()
    ref = np.ascontiguousarray(histograms[image], np.float32)
    mat = np.ascontiguousarray(np.stack([np.asarray(histograms[o], np.float32) for o in others]))
    out = np.zeros(len(others), np.float64)
    check(_lib.load().osfm_vlad_distances(ctx.handle, _fp(ref, C.c_float), _fp(mat, C.c_float), len(others), len(ref), _fp(out, C.c_double)),
          "osfm_vlad_distances")
    return image, list(out), others


def bow_distances(image: str, other_images: Iterable[str], histograms: Dict[str, np.ndarray], ctx=None) -> Tuple[str, List[float], List[str]]:
    """``pairs_selection.bow_distances`` (pairs_selection.py:690-708): L1 distances between the BoW histogram of ``image`` and those of the
    other images that have one, in the order given (the reference iterates ``other_images`` as is); ``osfm_bow_distances`` sums in
    numpy's order, so the values are the reference's bit for bit."""
    if image not in histograms:
        return image, [], []
    others = [o for o in other_images if o != image and o in histograms]
    if not others:
        return image, [], []
    ctx = ctx or default_context()
    ref = np.ascontiguousarray(histograms[image], np.float64)
    mat = np.ascontiguousarray(np.stack([np.asarray(histograms[o], np.float64) for o in others]))
    out = np.zeros(len(others), np.float64)
    check(_lib.load().osfm_bow_distances(ctx.handle, _fp(ref, C.c_double), _fp(mat, C.c_double), len(others), len(ref), _fp(out, C.c_double)),
          "osfm_bow_distances")
    return image, list(out), others


def bow_histogram(words_of_image: np.ndarray, n_words: int, weights: np.ndarray) -> np.ndarray:
    """``BagOfWords.histogram`` (bow.py:34-36): weighted, normalised word counts (host arithmetic, as in the reference)"""
    h = np.bincount(np.asarray(words_of_image).astype(np.int64), minlength=n_words) * weights
    return h / h.sum()
