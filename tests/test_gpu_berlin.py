"""data/berlin's example reconstruction (see tests/berlin_case.py) on the HIP solvers: the streaming solver (osfm_ba_solve) and the
general one (osfm_bundle_solve, also with the dataset's ground control points as the dataset's config.yaml asks), against the
measurements the CPU oracle gives in tests/test_berlin_example.py."""
import numpy as np
import pytest

import berlin_case as case

pytestmark = pytest.mark.gpu


def test_reference_solution_reprojects_its_tracks(gpu_ctx):
    for general in (False, True):
        _, ba, _, _, errs = case.run({"bundle_use_gcp": False, "bundle_max_iterations": 0}, force_general=general)
        assert ba.solver == ("osfm_bundle_solve" if general else "osfm_ba_solve")
        case.check_reference_solution_reprojects_its_tracks(errs)


def test_stationarity_and_parity_with_the_oracle(gpu_ctx, oracle_lib):
    rec, ba, before, after, errs = case.run(dict(case.NO_CAMERA_PRIOR, bundle_use_gcp=False))
    case.check_stationarity(ba, before, after)
    # the same flattened problem in the CPU oracle: same cost history, same optimum (flattened from an adjuster that has not run:
    # run() writes the optimum back into the adjuster's own copies)
    _, ba0, _, _, _ = case.run(dict(case.NO_CAMERA_PRIOR, bundle_use_gcp=False, bundle_max_iterations=0))
    prob = ba0._streaming_form(ba0._problem())
    o = oracle_lib.ba_solve(prob, max_iterations=100)
    g = ba._report
    n = min(len(o["cost_history"]), len(g["cost_history"]), 10)
    assert np.allclose(g["cost_history"][:n], o["cost_history"][:n], rtol=1e-6)
    assert abs(g["final_cost"] - o["final_cost"]) < 1e-5 * o["final_cost"]
    # streaming and general solver agree on the optimum of this real-data problem
    _, ba2, _, after2, _ = case.run(dict(case.NO_CAMERA_PRIOR, bundle_use_gcp=False), force_general=True)
    assert abs(ba2._report["final_cost"] - g["final_cost"]) < 1e-4 * g["final_cost"]
    assert np.allclose(after2["cam"], after["cam"], atol=2e-3)


def test_defaults_and_control_points(gpu_ctx):
    _, ba, before, after, _ = case.run({"bundle_use_gcp": False})
    case.check_defaults_cannot_have_produced_it(ba, before, after)
    # data/berlin/config.yaml: bundle_use_gcp yes -- the three control points join as point priors + observations (general solver)
    rec, ba, before, after, errs = case.run({"bundle_use_gcp": True})
    assert ba.solver == "osfm_bundle_solve" and np.isfinite(errs).all()
    assert ba._report["final_cost"] < ba._report["initial_cost"]
