"""``pyfeatures`` (opensfm/src/features/python/pybind.cc:55-63): ``hahog``, ``match_using_words``, ``compute_vlad_descriptor``,
``compute_vlad_distances`` with the reference's arguments and return values.  ``match_using_words`` is a functional shim: every call
uploads both images into a resident ``WordsStore`` and tears it down again; batch users go through ``matching.match_images_with_pairs``
(``matcher_type: WORDS``), which keeps the store resident over the whole pair list."""
from typing import Dict, List, Tuple

import numpy as np

from .. import features as _features
from .. import words as _words


def hahog(image, peak_threshold: float = 0.003, edge_threshold: float = 10, target_num_features: int = 0):
    """features::hahog (hahog.cc:125-206): (points n x 4, descriptors n x 128) float32, or None for an empty image"""
    return _features.hahog(np.asarray(image), peak_threshold, edge_threshold, target_num_features)


def match_using_words(features1, words1, features2, words2, lowes_ratio: float, max_checks: int):
    """features::match_using_words (matching.cc:73-88): (m, 2) int array of (index in features1, index in features2)"""
    f1, f2 = np.asarray(features1, np.float32), np.asarray(features2, np.float32)
    if len(f1) == 0 or len(f2) == 0:  # masked features can leave an image empty (matching.match_words): no matches, as the reference
        return np.zeros((0, 2), np.int32)
    # the queries (image 1) come with their n closest words, the indexed image with ONE word per feature (matching.py:637-656 passes
    # words2[:, 0]); the resident store keeps one width per call and indexes an image by its features' first word
    w1 = np.asarray(words1).reshape(len(f1), -1)
    w2 = np.asarray(words2).reshape(len(f2), -1)
    if w2.shape[1] != 1:
        raise ValueError("words2 holds one word per feature of the second image (the reference passes words2[:, 0])")
    w2 = np.repeat(w2, w1.shape[1], axis=1)
    store = _words.WordsStore([f1, f2], [w1, w2])
    try:
        found, _ = _words.match_words_pairs(store, [(0, 1)], {"lowes_ratio": lowes_ratio, "bow_num_checks": max_checks}, symmetric=False)
    finally:
        store.close()
    return np.asarray(found[0], np.int32).reshape(-1, 2)


def compute_vlad_descriptor(features, vlad_centers):
    """features::compute_vlad_descriptor (matching.cc:90-124): the unnormalised descriptor, centres x dimension floats"""
    centers = np.asarray(vlad_centers, np.float32)
    if centers.ndim != 2 or centers.shape[0] == 0 or centers.shape[1] == 0:
        raise RuntimeError("Zero VLAD centers or zero length VLAD words.")  # matching.cc:95-97
    v = _words.unnormalized_vlad(np.asarray(features, np.float32), centers)
    return np.zeros(centers.size, np.float32) if v is None else v


def compute_vlad_distances(vlad_descriptors: Dict[str, np.ndarray], image: str, other_images) -> Tuple[List[float], List[str]]:
    """features::compute_vlad_distances (matching.cc:126-152): L2 distances of ``image``'s descriptor to the others that have one"""
    _, distances, others = _words.vlad_distances(image, list(other_images), vlad_descriptors)
    return distances, others
