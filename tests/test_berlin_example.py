"""The reference's own Ceres-produced artefact (data/berlin/reconstruction_example.json + tracks_example.csv) through the adapter
(= BAHelpers::Bundle) with the CPU oracle as the solver: the first check of the restated residuals and LM against something Ceres
wrote.  The same case runs on the HIP solver in test_gpu_berlin.py."""
import numpy as np
import pytest

import berlin_case as case
from opensfm_amd import bundle


@pytest.fixture()
def oracle_streaming(monkeypatch, oracle_lib):
    def solve(problem, config=None, ctx=None, **overrides):
        cfg = dict(config or {})
        out = oracle_lib.ba_solve(problem, loss=cfg.get("loss_function", "SoftLOneLoss"), loss_threshold=cfg.get("loss_function_threshold", 1.0),
                                  max_iterations=cfg.get("bundle_max_iterations", 100))
        out["brief_report"] = "oracle: iterations %d" % out["iterations"]
        return out

    monkeypatch.setattr(bundle, "bundle_arrays", solve)


def test_golden_is_the_reference_artefact():
    import json
    import os

    ref = "/root/reference/data/berlin/reconstruction_example.json"
    if not os.path.exists(ref):
        pytest.skip("reference not mounted")
    r = json.load(open(ref))[0]
    g = json.load(open(case.GOLDEN))
    assert g["cameras"] == r["cameras"] and {k: v["coordinates"] for k, v in r["points"].items()} == g["points"]
    assert all(g["shots"][k]["rotation"] == v["rotation"] and g["shots"][k]["translation"] == v["translation"] for k, v in r["shots"].items())


def test_reference_solution_reprojects_its_tracks(oracle_streaming):
    _, ba, _, _, errs = case.run({"bundle_use_gcp": False, "bundle_max_iterations": 0})
    assert ba.solver == "osfm_ba_solve"
    case.check_reference_solution_reprojects_its_tracks(errs)


def test_reference_solution_is_nearly_stationary_without_the_camera_prior(oracle_streaming):
    _, ba, before, after, errs = case.run(dict(case.NO_CAMERA_PRIOR, bundle_use_gcp=False))
    case.check_stationarity(ba, before, after)
    assert np.median(errs) * 3264 < 1.0


def test_present_defaults_cannot_have_produced_the_artefact(oracle_streaming):
    _, ba, before, after, _ = case.run({"bundle_use_gcp": False})
    case.check_defaults_cannot_have_produced_it(ba, before, after)
