"""Modules under the names of the reference's pybind11 extensions -- ``pyfeatures``, ``pyrobust``, ``pybundle``, ``pysfm`` -- holding the
entries of their surfaces that lie on the hot path (``opensfm/src/*/python/pybind.cc``), served by ``libosfm_mi355.so``.

The reference imports its extensions as ``from opensfm import pyfeatures`` ...; a deployment that wants ``bin/opensfm_run_all`` untouched
puts these modules in front: ``opensfm_amd.compat.install()`` registers them in ``sys.modules`` as ``opensfm.pyfeatures`` etc. for every
name that is NOT already importable (it never shadows a compiled extension unless ``force=True``), attribute by attribute: an attribute
this package does not provide is looked up in the compiled module when there is one, and raises ``AttributeError`` naming the missing
entry otherwise -- there is no CPU fallback behind these names."""
import importlib
import sys
import types
from typing import Dict

from . import pybundle, pyfeatures, pyrobust, pysfm

MODULES = {"pyfeatures": pyfeatures, "pyrobust": pyrobust, "pybundle": pybundle, "pysfm": pysfm}


class _Overlay(types.ModuleType):
    """attributes of the GPU module first, then of the compiled module it stands in front of"""

    def __init__(self, name: str, gpu, compiled=None):
        super().__init__(name, gpu.__doc__)
        self.__dict__["_gpu"], self.__dict__["_compiled"] = gpu, compiled

    def __getattr__(self, attr):
        for m in (self.__dict__["_gpu"], self.__dict__["_compiled"]):
            if m is not None and hasattr(m, attr):
                return getattr(m, attr)
        raise AttributeError(f"{self.__name__}.{attr} is not provided by opensfm_amd.compat (GPU hot-path subset) and no compiled module is present")


def install(package: str = "opensfm", force: bool = False) -> Dict[str, types.ModuleType]:
    """register ``<package>.pyfeatures`` ... in ``sys.modules``; returns what was registered"""
    done = {}
    for name, gpu in MODULES.items():
        full = f"{package}.{name}"
        compiled = None
        try:
            compiled = importlib.import_module(full)
            if isinstance(compiled, _Overlay):  # installed before
                if not force:
                    continue
                compiled = compiled.__dict__["_compiled"]
        except Exception:  # noqa: BLE001 -- the compiled extension is absent (or its own imports fail)
            compiled = None
        if compiled is not None and not force:
            continue
        sys.modules[full] = done[name] = _Overlay(full, gpu, compiled)
        pkg = sys.modules.get(package)
        if pkg is not None:
            setattr(pkg, name, done[name])
    return done
