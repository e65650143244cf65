"""opensfm_amd.compat: the hot-path entries of the reference's pybind11 modules under the reference's names (no GPU needed: names,
defaults and the install logic; the calls themselves are exercised in tests/test_gpu_compat.py)."""
import inspect
import os
import re
import sys
import types

import pytest

from opensfm_amd import compat

REF = "/root/reference/opensfm/src"


def test_install_registers_only_what_is_missing():
    pkg = types.ModuleType("osfm_compat_probe")
    sys.modules["osfm_compat_probe"] = pkg
    try:
        done = compat.install("osfm_compat_probe")
        assert sorted(done) == ["pybundle", "pyfeatures", "pyrobust", "pysfm"]
        from osfm_compat_probe import pyfeatures, pyrobust  # noqa: F401

        assert pkg.pyfeatures is sys.modules["osfm_compat_probe.pyfeatures"]
        with pytest.raises(AttributeError, match="akaze"):
            pyfeatures.akaze
        # a second install finds its own overlays "importable" and, without force, leaves them alone
        assert compat.install("osfm_compat_probe") == {}
        # a compiled module that is present is not shadowed ...
        fake = types.ModuleType("osfm_compat_probe.pyrobust")
        fake.ransac_line = lambda *a: "compiled"
        sys.modules["osfm_compat_probe.pyrobust"] = fake
        assert "pyrobust" not in compat.install("osfm_compat_probe")
        # ... unless forced, and then its other entries stay reachable behind the GPU ones
        forced = compat.install("osfm_compat_probe", force=True)
        assert forced["pyrobust"].ransac_line() == "compiled" and forced["pyrobust"].ransac_relative_pose is compat.pyrobust.ransac_relative_pose
    finally:
        for k in [k for k in sys.modules if k.startswith("osfm_compat_probe")]:
            del sys.modules[k]


def test_signatures_follow_the_pybind_definitions():
    sig = inspect.signature(compat.pyfeatures.hahog)
    assert list(sig.parameters) == ["image", "peak_threshold", "edge_threshold", "target_num_features"]
    assert [p.default for p in sig.parameters.values()][1:] == [0.003, 10, 0]  # features/python/pybind.cc:55-57
    assert list(inspect.signature(compat.pyfeatures.match_using_words).parameters) == ["features1", "words1", "features2", "words2", "lowes_ratio", "max_checks"]
    p = compat.pyrobust.RobustEstimatorParams()
    assert (p.iterations, p.probability, p.use_local_optimization, p.use_iteration_reduction) == (100, 0.99, True, True)
    assert [m.name for m in compat.pyrobust.RansacType] == ["RANSAC", "MSAC", "LMedS"]
    for name in ("bundle", "bundle_local", "bundle_shot_poses", "shot_neighborhood_ids", "bundle_to_map", "detect_alignment_constraints", "add_gcp_to_bundle"):
        assert callable(getattr(compat.pysfm.BAHelpers, name))
    # argument order of sfm/ba_helpers.h:15-49 as pybind11 exposes it
    assert list(inspect.signature(compat.pysfm.BAHelpers.bundle_local).parameters)[:6] == ["reconstruction", "camera_priors", "rig_camera_priors", "gcp",
                                                                                           "central_shot_id", "config"]
    assert list(inspect.signature(compat.pysfm.BAHelpers.bundle_shot_poses).parameters)[:5] == ["reconstruction", "shot_ids", "camera_priors",
                                                                                                "rig_camera_priors", "config"]
    assert list(inspect.signature(compat.pysfm.BAHelpers.shot_neighborhood_ids).parameters)[:5] == ["reconstruction", "central_shot_id", "radius",
                                                                                                    "min_common_points", "max_interior_size"]


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference is not mounted")
def test_names_exist_in_the_references_bindings():
    """every name this package provides is a name the reference's pybind.cc defines (read from the file)"""
    def defined(path, pattern):
        return set(re.findall(pattern, open(os.path.join(REF, path)).read()))

    feat = defined("features/python/pybind.cc", r'm\.def\("(\w+)"')
    assert {"hahog", "match_using_words", "compute_vlad_descriptor", "compute_vlad_distances"} <= feat
    rob = open(os.path.join(REF, "robust/python/pybind.cc")).read()
    assert 'm.def("ransac_relative_pose"' in rob and '"RobustEstimatorParams"' in rob and '"RansacType"' in rob
    for field in ("iterations", "probability", "use_local_optimization", "use_iteration_reduction", "score", "model", "lo_model", "inliers_indices"):
        assert f'"{field}"' in rob
    sfm = defined("sfm/python/pybind.cc", r'def_static\("(\w+)"')
    assert {"bundle", "bundle_local", "bundle_shot_poses", "shot_neighborhood_ids", "bundle_to_map", "detect_alignment_constraints", "add_gcp_to_bundle"} <= sfm
    bundle_methods = defined("bundle/python/pybind.cc", r'\.def\("(\w+)"')
    ours = {n for n, f in inspect.getmembers(compat.pybundle.BundleAdjuster, callable) if not n.startswith("_")}
    needed = {"run", "add_camera", "get_camera", "add_rig_camera", "get_rig_camera_pose", "add_rig_instance", "get_rig_instance_pose",
              "add_rig_instance_position_prior", "add_point", "add_point_prior", "get_point", "has_point", "add_point_projection_observation",
              "add_absolute_up_vector", "add_absolute_pan", "add_absolute_tilt", "add_absolute_roll", "set_point_projection_loss_function",
              "set_internal_parameters_prior_sd", "set_compute_reprojection_errors", "set_max_num_iterations", "set_num_threads",
              "set_use_analytic_derivatives", "set_linear_solver_type", "brief_report", "full_report"}
    assert needed <= bundle_methods and needed <= ours, sorted(needed - ours)
    # what the class has beyond the Python bindings are methods sfm::BAHelpers calls on the C++ object (bundle_adjuster.h), in snake case
    header = open(os.path.join(REF, "bundle/bundle_adjuster.h")).read()
    for extra in ours - bundle_methods:
        camel = "".join(w.capitalize() for w in extra.split("_")).replace("Sd", "SD")
        assert re.search(r"\b" + camel + r"\s*\(", header), (extra, camel)
