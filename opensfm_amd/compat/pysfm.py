"""``pysfm.BAHelpers`` (opensfm/src/sfm/python/pybind.cc:31-39): ``bundle``, ``bundle_local``, ``bundle_shot_poses``,
``shot_neighborhood_ids``, ``bundle_to_map``, ``detect_alignment_constraints``, ``add_gcp_to_bundle`` over the attributes the reference's
map objects expose to Python (opensfm_amd/opensfm_adapter.py) -- everything ``opensfm/reconstruction.py:70-149`` calls on it."""
from .. import opensfm_adapter as _adapter


class BAHelpers:
    bundle = staticmethod(_adapter.bundle)
    bundle_local = staticmethod(_adapter.bundle_local)
    bundle_shot_poses = staticmethod(_adapter.bundle_shot_poses)
    shot_neighborhood_ids = staticmethod(_adapter.shot_neighborhood_ids)
    bundle_to_map = staticmethod(_adapter.bundle_to_map)
    detect_alignment_constraints = staticmethod(_adapter.detect_alignment_constraints)
    add_gcp_to_bundle = staticmethod(_adapter.add_gcp_to_bundle)
