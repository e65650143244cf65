import ctypes, os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from opensfm_amd import _lib, features
rng = np.random.default_rng(3)
hip = ctypes.CDLL("libamdhip64.so")
ctx = _lib.Context()
for (r, c, tgt) in [(1536, 2048, 10000), (768, 1024, 10000), (1536, 2048, 1000)]:
    im = rng.random((r, c)).astype(np.float32)
    for _ in range(3):
        im = (im + np.roll(im, 1, 0) + np.roll(im, 1, 1) + np.roll(im, -1, 0) + np.roll(im, -1, 1)) / 5
    im = np.ascontiguousarray(im, np.float32)
    d = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(im.nbytes)); hip.hipMemcpy(d, ctypes.c_void_p(im.ctypes.data), ctypes.c_size_t(im.nbytes), 1)
    n = 48
    for conc in (4, 8):
        features.hahog_batch([d.value] * 8, 1e-5, 10.0, tgt, concurrency=conc, shapes=[im.shape] * 8, ctx=ctx)
        t0 = time.perf_counter(); res = features.hahog_batch([d.value] * n, 1e-5, 10.0, tgt, concurrency=conc, shapes=[im.shape] * n, ctx=ctx); dt = time.perf_counter() - t0
        print(r, c, 'target', tgt, 'features', len(res[0][0]), 'conc', conc, 'images/s', round(n / dt, 1))
    t0 = time.perf_counter()
    for _ in range(8): features._extract(im, 1e-5, 10.0, tgt, 0, ctx)
    print('   single (host image) images/s', round(8 / (time.perf_counter() - t0), 1))
