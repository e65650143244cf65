"""``pysfm.BAHelpers`` (opensfm/src/sfm/python/pybind.cc:31-39): ``bundle``, ``bundle_to_map``, ``detect_alignment_constraints``,
``add_gcp_to_bundle`` over the attributes the reference's map objects expose to Python (opensfm_amd/opensfm_adapter.py)."""
from .. import opensfm_adapter as _adapter


class BAHelpers:
    bundle = staticmethod(_adapter.bundle)
    bundle_to_map = staticmethod(_adapter.bundle_to_map)
    detect_alignment_constraints = staticmethod(_adapter.detect_alignment_constraints)
    add_gcp_to_bundle = staticmethod(_adapter.add_gcp_to_bundle)
