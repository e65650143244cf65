"""Test helper: run opensfm_amd's bundle-adjustment entry points on the HOST EMULATION of the kernels (tests/native/build_emu.py).
The product never does this -- `opensfm_amd._lib.load()` only ever opens the hipcc-built library; the tests swap the handle."""
import contextlib
import ctypes as C
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def _builder():
    spec = importlib.util.spec_from_file_location("build_emu", os.path.join(HERE, "native", "build_emu.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_cached = {}


def emu_library(sanitize: bool = False) -> C.CDLL:
    if sanitize not in _cached:
        from opensfm_amd import _lib

        lib = C.CDLL(_builder().build(sanitize=sanitize))
        for name, (res, args) in _lib._signatures().items():
            if hasattr(lib, name):
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
        lib.hipemu_launch_count.restype = C.c_long
        _cached[sanitize] = lib
    return _cached[sanitize]


@contextlib.contextmanager
def emulated(sanitize: bool = False):
    """inside: every opensfm_amd call that goes through _lib.load() runs on the emulated library"""
    from opensfm_amd import _lib

    lib = emu_library(sanitize)
    old_lib, old_ctx = _lib._lib, getattr(_lib._tls, "ctx", None)
    _lib._lib, _lib._tls.ctx = lib, {}
    try:
        yield lib
    finally:
        for c in _lib._tls.ctx.values():
            c.close()
        _lib._lib, _lib._tls.ctx = old_lib, old_ctx
