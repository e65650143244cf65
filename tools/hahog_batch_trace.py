"""rocprofv3 driver: one batch of resident images through osfm_hahog_extract_batch"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from opensfm_amd import _lib, features
rng = np.random.default_rng(3)
im = rng.random((1536, 2048)).astype(np.float32)
k = np.ones(9, np.float32) / 9
im = np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 1, im)
im = np.ascontiguousarray(np.apply_along_axis(lambda v: np.convolve(v, k, mode="same"), 0, im), np.float32)
ctx = _lib.Context()
conc = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
features.hahog_batch([im] * conc, 1e-5, 10.0, 10000, concurrency=conc, ctx=ctx)
import time
t0 = time.perf_counter()
r = features.hahog_batch([im] * n, 1e-5, 10.0, 10000, concurrency=conc, ctx=ctx)
dt = time.perf_counter() - t0
print("features", len(r[0][0]), "images/s", n / dt)
