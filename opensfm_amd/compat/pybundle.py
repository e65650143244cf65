"""``pybundle`` (opensfm/src/bundle/python/pybind.cc:45-117): ``BundleAdjuster`` with the reference's method names, as far as
``sfm::BAHelpers`` uses them (relative / linear motion, heatmaps and scale groups raise ``NotImplementedError``)."""
from ..bundle import BundleAdjuster  # noqa: F401
