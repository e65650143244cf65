"""Generates tests/golden/tracks_golden.json by RUNNING the reference's own tracks code.

opensfm/tracking.py and opensfm/unionfind.py are pure Python, but `import opensfm` pulls in the compiled pybind modules, which
are not built in this image.  This script loads the two files from /root/reference under stand-in modules for what they import
(opensfm.pymap with a recording TracksManager / Observation, opensfm.dataset_base, networkx) and calls the reference's
`tracking.create_tracks_manager` (opensfm/tracking.py:68-140) unchanged on seeded random match graphs: union-find grouping,
`_good_track`, the numbering of the surviving tracks and the order of their observations all come from the reference's code.

Run (only where /root/reference is mounted):  python tests/golden/gen_tracks_golden.py
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference/opensfm"
HERE = os.path.dirname(os.path.abspath(__file__))


class _Observation:
    NO_SEMANTIC_VALUE = -1

    def __init__(self, x, y, s, r, g, b, featureid, segmentation, instance):
        self.id = int(featureid)
        self.depth_prior = None


class _TracksManager:
    def __init__(self):
        self.observations = []  # (image, track_id, feature id) in insertion order

    def add_observation(self, image, track_id, obs):
        self.observations.append((image, str(track_id), obs.id))


def load_reference_tracking():
    pkg = types.ModuleType("opensfm")
    pkg.__path__ = [REF]
    pymap = types.ModuleType("opensfm.pymap")
    pymap.Observation = _Observation
    pymap.TracksManager = _TracksManager
    pymap.Depth = lambda **kw: kw
    dsb = types.ModuleType("opensfm.dataset_base")
    dsb.DataSetBase = object
    nx = types.ModuleType("networkx")
    nx.Graph = object
    pkg.pymap = pymap
    sys.modules.update({"opensfm": pkg, "opensfm.pymap": pymap, "opensfm.dataset_base": dsb})
    sys.modules.setdefault("networkx", nx)
    for name in ("unionfind", "tracking"):
        spec = importlib.util.spec_from_file_location("opensfm." + name, os.path.join(REF, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["opensfm." + name] = mod
        spec.loader.exec_module(mod)
    return sys.modules["opensfm.tracking"]


def random_case(rng, n_images, n_features, p_pair, p_match, p_wrong, min_length):
    images = ["im%02d" % i for i in range(n_images)]
    matches = {}
    for a in range(n_images):
        for b in range(a + 1, n_images):
            if rng.random() > p_pair:
                continue
            m = []
            for f in range(n_features):
                if rng.random() < p_match:
                    g = f if rng.random() > p_wrong else int(rng.integers(0, n_features))  # wrong matches merge tracks -> image twice
                    m.append((f, g))
            if m:
                matches[images[a], images[b]] = m
    return {"images": images, "n_features": n_features, "min_length": min_length,
            "matches": [[a, b, m] for (a, b), m in matches.items()]}


def main():
    tracking = load_reference_tracking()
    rng = np.random.default_rng(2024)
    cases = []
    for (ni, nf, pp, pm, pw, ml) in ((4, 12, 1.0, 0.6, 0.0, 2), (6, 30, 0.7, 0.5, 0.1, 2), (8, 40, 0.5, 0.4, 0.2, 3), (5, 25, 1.0, 0.9, 0.3, 2),
                                     (3, 5, 1.0, 1.0, 0.0, 4), (10, 60, 0.4, 0.3, 0.05, 2)):
        c = random_case(rng, ni, nf, pp, pm, pw, ml)
        feats = {im: np.zeros((nf, 3)) for im in c["images"]}
        cols = {im: np.zeros((nf, 3), int) for im in c["images"]}
        matches = {(a, b): [tuple(x) for x in m] for a, b, m in c["matches"]}
        tm = tracking.create_tracks_manager(feats, cols, {}, {}, matches, c["min_length"], {})
        c["observations"] = [[im, tid, fid] for im, tid, fid in tm.observations]
        cases.append(c)
    with open(os.path.join(HERE, "tracks_golden.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    print("wrote", len(cases), "cases;", sum(len(c["observations"]) for c in cases), "observations")


if __name__ == "__main__":
    main()
