"""The product's bundle-adjustment kernels and LM driver (opensfm_amd/csrc/ba.hip, ba_general.hip) executed on the HOST by the HIP
emulation of tests/native/hipemu -- every kernel, launch, LDS exchange, wavefront shuffle and fp64 MFMA of the real sources, workgroup by
workgroup -- against the CPU oracle.  This is the `-m "not gpu"` twin of tests/test_gpu_ba.py / test_gpu_bundle_general.py at sizes the
emulation finishes in seconds: it catches an indexing slip or an uninitialised read (emulated device memory and dynamic LDS are
poisoned with NaNs) before a GPU lease is spent on it.  The emulated library is test infrastructure; the product never loads it."""
import numpy as np
import pytest

from emu_util import emulated
from opensfm_amd import synthetic

NO_TOL = dict(function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)


def _rmse_px(err):
    return float(np.sqrt((np.asarray(err)[:, :2] ** 2).sum(1).mean()) * 2000.0)


def test_streaming_solver_narrow_band_matches_oracle(oracle_lib):
    """sequence, half-width 4: matrix-core band assembly, cyclic reduction in LDS, exact camera border, PCG"""
    from opensfm_amd import bundle

    pr = synthetic.make_ba_scene(20, 300, 5, seed=11)
    with emulated() as lib:
        g = bundle.bundle_arrays(pr, {"bundle_max_iterations": 4}, **NO_TOL)
        assert lib.hipemu_launch_count() > 100
    o = oracle_lib.ba_solve(pr, max_iterations=4, **NO_TOL)
    assert g["preconditioner_bandwidth"] == g["shot_bandwidth"] and g["shot_bandwidth"] >= 2
    assert np.allclose(g["cost_history"], o["cost_history"], rtol=1e-10)
    assert abs(_rmse_px(g["reproj_err"]) - _rmse_px(o["reproj_err"])) < 1e-4
