"""Plain-Python value types with the attribute / method names of the reference's ``pygeometry`` and ``pymap`` objects that the
matching + bundle-adjustment path touches (``opensfm/src/geometry/python/pybind.cc``, ``opensfm/src/map/python/pybind.cc``).

The product never needs these: every entry point accepts ANY object with the same attributes (the reference's own compiled types
included).  They exist so that the path can be driven, tested and demonstrated where the reference's compiled modules are absent,
and they are what ``opensfm_adapter`` and the ``BundleAdjuster`` facade hand back.  Host-side glue only: no numerics of the hot
path live here.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import numpy as np


def _rodrigues(r) -> np.ndarray:
    r = np.asarray(r, float).reshape(3)
    th = float(np.linalg.norm(r))
    K = np.array([[0, -r[2], r[1]], [r[2], 0, -r[0]], [-r[1], r[0], 0]])
    if th < 1e-300:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * K @ K


def _log_rotation(R) -> np.ndarray:
    """angle-axis of a rotation matrix (ceres::RotationMatrixToAngleAxis through the quaternion)"""
    R = np.asarray(R, float).reshape(3, 3)
    q = np.empty(4)
    tr = np.trace(R)
    if tr >= 0:
        t = np.sqrt(tr + 1.0)
        q[0] = 0.5 * t
        t = 0.5 / t
        q[1:] = [(R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i + 1] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[k, j] - R[j, k]) * t
        q[j + 1] = (R[j, i] + R[i, j]) * t
        q[k + 1] = (R[k, i] + R[i, k]) * t
    s = float(np.linalg.norm(q[1:]))
    if s > 0:
        two_theta = 2.0 * (np.arctan2(-s, -q[0]) if q[0] < 0 else np.arctan2(s, q[0]))
        return q[1:] * (two_theta / s)
    return q[1:] * 2.0


class Pose:
    """``pygeometry.Pose``: world-to-camera angle-axis ``rotation`` and ``translation`` (x_cam = R x_world + t)."""

    def __init__(self, rotation=None, translation=None):
        self.rotation = np.zeros(3) if rotation is None else np.asarray(rotation, float).reshape(3).copy()
        self.translation = np.zeros(3) if translation is None else np.asarray(translation, float).reshape(3).copy()

    # -- the camera-to-world form the bundle adjustment works in (bundle/data/pose.h:34-43) --
    @classmethod
    def from_cam_to_world(cls, rotation_min, origin) -> "Pose":
        p = cls()
        p.set_from_cam_to_world(rotation_min, origin)
        return p

    def set_from_cam_to_world(self, rotation_min, origin) -> None:
        self.rotation = -np.asarray(rotation_min, float).reshape(3)
        self.translation = -_rodrigues(self.rotation) @ np.asarray(origin, float).reshape(3)

    def cam_to_world_parameters(self) -> np.ndarray:
        """[rx ry rz tx ty tz] of bundle::Pose (CAM_TO_WORLD): minus the stored rotation, the origin"""
        return np.r_[-self.rotation, self.get_origin()]

    def get_rotation_matrix(self) -> np.ndarray:
        return _rodrigues(self.rotation)

    def get_R_world_to_cam(self) -> np.ndarray:
        return _rodrigues(self.rotation)

    def get_R_cam_to_world(self) -> np.ndarray:
        return _rodrigues(self.rotation).T

    def get_R_world_to_cam_min(self) -> np.ndarray:
        return self.rotation.copy()

    def get_R_cam_to_world_min(self) -> np.ndarray:
        return -self.rotation

    def get_t_world_to_cam(self) -> np.ndarray:
        return self.translation.copy()

    def get_origin(self) -> np.ndarray:
        return -_rodrigues(self.rotation).T @ self.translation

    def get_t_cam_to_world(self) -> np.ndarray:
        return self.get_origin()

    def set_origin(self, origin) -> None:
        self.translation = -_rodrigues(self.rotation) @ np.asarray(origin, float).reshape(3)

    def set_rotation_matrix(self, R) -> None:
        self.rotation = _log_rotation(R)

    def get_world_to_cam(self) -> np.ndarray:
        T = np.eye(4)
        T[:3, :3], T[:3, 3] = self.get_rotation_matrix(), self.translation
        return T

    def get_cam_to_world(self) -> np.ndarray:
        return np.linalg.inv(self.get_world_to_cam())

    def transform(self, point) -> np.ndarray:
        return self.get_rotation_matrix() @ np.asarray(point, float) + self.translation

    def transform_inverse(self, point) -> np.ndarray:
        return self.get_rotation_matrix().T @ (np.asarray(point, float) - self.translation)

    def compose(self, other: "Pose") -> "Pose":
        """self o other: x -> self(other(x))"""
        R = self.get_rotation_matrix() @ other.get_rotation_matrix()
        t = self.get_rotation_matrix() @ other.translation + self.translation
        return Pose(_log_rotation(R), t)

    def inverse(self) -> "Pose":
        R = self.get_rotation_matrix().T
        return Pose(_log_rotation(R), -R @ self.translation)

    def relative_to(self, base: "Pose") -> "Pose":
        return self.compose(base.inverse())

    def is_identity(self) -> bool:
        return not (self.rotation.any() or self.translation.any())

    def __repr__(self) -> str:  # pragma: no cover
        return f"Pose(rotation={self.rotation}, translation={self.translation})"


class Similarity:
    """``pygeometry.Similarity``: x -> scale * R x + t (``opensfm/src/geometry/similarity.h``)"""

    def __init__(self, rotation=None, translation=None, scale: float = 1.0):
        self.rotation = np.zeros(3) if rotation is None else np.asarray(rotation, float).reshape(3).copy()
        self.translation = np.zeros(3) if translation is None else np.asarray(translation, float).reshape(3).copy()
        self.scale = float(scale)

    def transform(self, point) -> np.ndarray:
        return self.scale * (_rodrigues(self.rotation) @ np.asarray(point, float)) + self.translation

    def parameters(self) -> np.ndarray:
        """[rx ry rz tx ty tz scale] of bundle::Similarity (bias.h:10-31)"""
        return np.r_[self.rotation, self.translation, self.scale]


# projection_type -> native parameter names, [projection][distortion][affine] (camera_instances.h:127-160)
CAMERA_PARAMETERS = {
    "perspective": ("k1", "k2", "focal"),
    "fisheye": ("k1", "k2", "focal"),
    "brown": ("k1", "k2", "k3", "p1", "p2", "focal", "aspect_ratio", "cx", "cy"),
    "fisheye_opencv": ("k1", "k2", "k3", "k4", "focal", "aspect_ratio", "cx", "cy"),
    "fisheye62": ("k1", "k2", "k3", "k4", "k5", "k6", "p1", "p2", "focal", "aspect_ratio", "cx", "cy"),
    "fisheye624": ("k1", "k2", "k3", "k4", "k5", "k6", "p1", "p2", "s0", "s1", "s2", "s3", "focal", "aspect_ratio", "cx", "cy"),
    "dual": ("transition", "k1", "k2", "focal"),
    "radial": ("k1", "k2", "focal", "aspect_ratio", "cx", "cy"),
    "simple_radial": ("k1", "focal", "aspect_ratio", "cx", "cy"),
    "spherical": (),
}
CAMERA_MODEL_IDS = {"perspective": 0, "fisheye": 1, "brown": 2, "fisheye_opencv": 3, "fisheye62": 4, "fisheye624": 5, "dual": 6, "radial": 7,
                    "simple_radial": 8, "spherical": 9, "equirectangular": 9}


def camera_parameter_values(camera) -> np.ndarray:
    """the native parameter vector (16 slots) of any object with the reference's camera attributes"""
    names = CAMERA_PARAMETERS["spherical" if camera.projection_type == "equirectangular" else camera.projection_type]
    out = np.zeros(16)
    for k, n in enumerate(names):
        out[k] = float(camera.principal_point[0 if n == "cx" else 1]) if n in ("cx", "cy") else float(getattr(camera, n))
    return out


def set_camera_parameter_values(camera, values) -> None:
    names = CAMERA_PARAMETERS["spherical" if camera.projection_type == "equirectangular" else camera.projection_type]
    pp = None
    for k, n in enumerate(names):
        if n in ("cx", "cy"):
            pp = np.array(camera.principal_point, float) if pp is None else pp
            pp[0 if n == "cx" else 1] = float(values[k])
        else:
            setattr(camera, n, float(values[k]))
    if pp is not None:
        camera.principal_point = pp


class Camera:
    """``pygeometry.Camera`` as far as the path reads and writes it: ``id``, ``projection_type``, ``width`` / ``height`` and the
    named parameters (``focal``, ``k1`` ... ``aspect_ratio``, ``principal_point``, ``transition``)."""

    def __init__(self, projection_type: str = "perspective", **params):
        self.id = ""
        self.projection_type = projection_type
        self.width = self.height = 0
        self.focal = 1.0
        self.aspect_ratio = 1.0
        self.principal_point = np.zeros(2)
        self.transition = 1.0
        for n in ("k1", "k2", "k3", "k4", "k5", "k6", "p1", "p2", "s0", "s1", "s2", "s3"):
            setattr(self, n, 0.0)
        for k, v in params.items():
            setattr(self, k, v)

    @staticmethod
    def create_perspective(focal, k1, k2) -> "Camera":
        return Camera("perspective", focal=float(focal), k1=float(k1), k2=float(k2))

    @staticmethod
    def create_fisheye(focal, k1, k2) -> "Camera":
        return Camera("fisheye", focal=float(focal), k1=float(k1), k2=float(k2))

    @staticmethod
    def create_dual(transition, focal, k1, k2) -> "Camera":
        return Camera("dual", transition=float(transition), focal=float(focal), k1=float(k1), k2=float(k2))

    @staticmethod
    def create_spherical() -> "Camera":
        return Camera("spherical")

    @staticmethod
    def _with_distortion(kind, focal, aspect_ratio, principal_point, distortion, names) -> "Camera":
        c = Camera(kind, focal=float(focal), aspect_ratio=float(aspect_ratio), principal_point=np.asarray(principal_point, float).copy())
        for n, v in zip(names, np.asarray(distortion, float).reshape(-1)):
            setattr(c, n, float(v))
        return c

    @staticmethod
    def create_brown(focal, aspect_ratio, principal_point, distortion) -> "Camera":
        return Camera._with_distortion("brown", focal, aspect_ratio, principal_point, distortion, ("k1", "k2", "k3", "p1", "p2"))

    @staticmethod
    def create_fisheye_opencv(focal, aspect_ratio, principal_point, distortion) -> "Camera":
        return Camera._with_distortion("fisheye_opencv", focal, aspect_ratio, principal_point, distortion, ("k1", "k2", "k3", "k4"))

    @staticmethod
    def create_fisheye62(focal, aspect_ratio, principal_point, distortion) -> "Camera":
        return Camera._with_distortion("fisheye62", focal, aspect_ratio, principal_point, distortion, ("k1", "k2", "k3", "k4", "k5", "k6", "p1", "p2"))

    @staticmethod
    def create_fisheye624(focal, aspect_ratio, principal_point, distortion) -> "Camera":
        return Camera._with_distortion("fisheye624", focal, aspect_ratio, principal_point, distortion,
                                       ("k1", "k2", "k3", "k4", "k5", "k6", "p1", "p2", "s0", "s1", "s2", "s3"))

    @staticmethod
    def create_radial(focal, aspect_ratio, principal_point, distortion) -> "Camera":
        return Camera._with_distortion("radial", focal, aspect_ratio, principal_point, distortion, ("k1", "k2"))

    @staticmethod
    def create_simple_radial(focal, aspect_ratio, principal_point, k1) -> "Camera":
        return Camera._with_distortion("simple_radial", focal, aspect_ratio, principal_point, [k1], ("k1",))

    def get_parameters_values(self) -> np.ndarray:
        return camera_parameter_values(self)[: len(CAMERA_PARAMETERS[self.projection_type])]

    def set_parameters_values(self, values) -> None:
        set_camera_parameter_values(self, values)

    def copy(self) -> "Camera":
        import copy

        return copy.deepcopy(self)


# ---- the slice of pymap the bundle adjustment walks (opensfm/src/map) ----
class ShotMeasurements:
    def __init__(self, gps_position=None, gps_accuracy=None):
        self.gps_position = None if gps_position is None else np.asarray(gps_position, float)
        self.gps_accuracy = gps_accuracy


class Depth:
    """``map::Depth`` (map/observation.h:10-18): the depth prior of an observation"""

    def __init__(self, value: float, is_radial: bool, std_deviation: float):
        self.value, self.is_radial, self.std_deviation = float(value), bool(is_radial), float(std_deviation)


class Observation:
    def __init__(self, x: float, y: float, scale: float, depth_prior: Optional[Depth] = None):
        self.point = np.array([x, y], float)
        self.scale = float(scale)
        self.depth_prior = depth_prior  # map::Observation::depth_prior (observation.h:50)


class RigCamera:
    def __init__(self, rig_camera_id: str, pose: Optional[Pose] = None):
        self.id = rig_camera_id
        self.pose = pose or Pose()


class Shot:
    def __init__(self, shot_id: str, camera: Camera, rig_instance: "RigInstance", rig_camera: RigCamera):
        self.id = shot_id
        self.camera = camera
        self.rig_instance = rig_instance
        self.rig_camera = rig_camera
        self.metadata = ShotMeasurements()
        self.observations: Dict[str, Observation] = {}  # landmark id -> observation

    @property
    def pose(self) -> Pose:
        """world-to-camera pose of the shot: rig camera pose o rig instance pose (map/shot.h)"""
        return self.rig_camera.pose.compose(self.rig_instance.pose)

    def get_landmark_observations(self) -> Dict[str, Observation]:
        return self.observations


class RigInstance:
    def __init__(self, rig_instance_id: str, pose: Optional[Pose] = None):
        self.id = rig_instance_id
        self.pose = pose or Pose()
        self.shots: Dict[str, Shot] = {}
        self.rig_camera_ids: Dict[str, str] = {}


class Landmark:
    def __init__(self, landmark_id: str, coordinates):
        self.id = landmark_id
        self.coordinates = np.asarray(coordinates, float).copy()
        self.reprojection_errors: Dict[str, np.ndarray] = {}


class GroundControlPointObservation:
    def __init__(self, shot_id: str, projection):
        self.shot_id = shot_id
        self.projection = np.asarray(projection, float)


class GroundControlPoint:
    def __init__(self, gcp_id: str, lla: Optional[Dict[str, float]] = None, has_altitude: bool = True):
        self.id = gcp_id
        self.lla = lla or {}
        self.has_altitude = has_altitude
        self.observations: List[GroundControlPointObservation] = []


class TopocentricConverter:
    """local ENU frame around (lat, lon, alt) -- opensfm/geo.py's WGS84 math (ecef_from_lla / topocentric_from_lla)"""
    A, B = 6378137.0, 6356752.314245

    def __init__(self, lat: float = 0.0, lon: float = 0.0, alt: float = 0.0):
        self.lat, self.lon, self.alt = lat, lon, alt

    @classmethod
    def _ecef(cls, lat, lon, alt):
        a2, b2 = cls.A**2, cls.B**2
        lat, lon = np.radians(lat), np.radians(lon)
        L = 1.0 / np.sqrt(a2 * np.cos(lat) ** 2 + b2 * np.sin(lat) ** 2)
        return np.array([(a2 * L + alt) * np.cos(lat) * np.cos(lon), (a2 * L + alt) * np.cos(lat) * np.sin(lon), (b2 * L + alt) * np.sin(lat)])

    def to_topocentric(self, lat, lon, alt) -> np.ndarray:
        la, lo = np.radians(self.lat), np.radians(self.lon)
        d = self._ecef(lat, lon, alt) - self._ecef(self.lat, self.lon, self.alt)
        R = np.array([[-np.sin(lo), np.cos(lo), 0.0], [-np.sin(la) * np.cos(lo), -np.sin(la) * np.sin(lo), np.cos(la)],
                      [np.cos(la) * np.cos(lo), np.cos(la) * np.sin(lo), np.sin(la)]])
        return R @ d


class Reconstruction:
    """``types.Reconstruction`` as a bag of dicts: cameras, rig_cameras, rig_instances, shots, points, biases, reference"""

    def __init__(self):
        self.cameras: Dict[str, Camera] = {}
        self.rig_cameras: Dict[str, RigCamera] = {}
        self.rig_instances: Dict[str, RigInstance] = {}
        self.shots: Dict[str, Shot] = {}
        self.points: Dict[str, Landmark] = {}
        self.biases: Dict[str, Similarity] = {}
        self.reference = TopocentricConverter()

    @property
    def map(self) -> "Reconstruction":
        """``types.Reconstruction.map`` (the pymap.Map the reference hands to ``pysfm.BAHelpers``): the bag of dicts itself"""
        return self

    def add_camera(self, camera: Camera) -> Camera:
        self.cameras[camera.id] = camera
        self.biases.setdefault(camera.id, Similarity())
        return camera

    def add_rig_camera(self, rig_camera: RigCamera) -> RigCamera:
        self.rig_cameras[rig_camera.id] = rig_camera
        return rig_camera

    def add_rig_instance(self, rig_instance: RigInstance) -> RigInstance:
        self.rig_instances[rig_instance.id] = rig_instance
        return rig_instance

    def create_shot(self, shot_id: str, camera_id: str, pose: Optional[Pose] = None, rig_camera_id: Optional[str] = None,
                    rig_instance_id: Optional[str] = None) -> Shot:
        """like ``types.Reconstruction.create_shot``: a shot on its own rig instance with the identity rig camera unless told otherwise"""
        rc = self.rig_cameras.get(rig_camera_id or camera_id) or self.add_rig_camera(RigCamera(rig_camera_id or camera_id))
        ri = self.rig_instances.get(rig_instance_id or shot_id) or self.add_rig_instance(RigInstance(rig_instance_id or shot_id, pose))
        shot = Shot(shot_id, self.cameras[camera_id], ri, rc)
        ri.shots[shot_id] = shot
        ri.rig_camera_ids[shot_id] = rc.id
        self.shots[shot_id] = shot
        return shot

    def create_point(self, point_id: str, coordinates) -> Landmark:
        self.points[point_id] = Landmark(point_id, coordinates)
        return self.points[point_id]

    def add_observation(self, shot_id: str, point_id: str, observation: Observation) -> None:
        self.shots[shot_id].observations[point_id] = observation


def optional_value(x) -> Optional[Any]:
    """value of a reference ``OptionalValue`` (``.has_value`` / ``.value``), of a plain value, or None"""
    if x is None:
        return None
    if hasattr(x, "has_value"):
        return x.value if x.has_value else None
    return x
