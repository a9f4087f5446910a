"""Seeded synthetic LiDAR data for the scan-to-map hot path.

No external data exists for this path (the reference ships no clouds, bags or
fixtures: /root/reference/ros/open3d_slam_ros/data/.gitkeep only), so every
benchmark / parity input is generated here from fixed seeds:

* scene  : hollow box room [-30,30] x [-30,30] x [0,10] m + 12 vertical
           cylinders (r = 0.5 m, h = 10 m), seed 1234
* map    : N area-weighted uniform surface samples with analytic normals
           oriented towards free space, seed 1235
* scans  : VLP-16-like (16 rings x 4096 az = 65 536 rays) and OS-128-like
           (128 rings x 1024 az = 131 072 rays) ray casts against the scene,
           range noise N(0, 0.01 m), seed 1236 + frame

The *map frame* is the world frame shifted by (0, 0, -1.5) so that the nominal
sensor position is the map origin and "initial guess = identity" (SURVEY §8d).
All arrays are float64, C-contiguous, shape (n, 3) -- the layout of
`open3d::geometry::PointCloud::points_` (std::vector<Eigen::Vector3d>) that the
reference passes across its registration seam
(open3d_slam/include/open3d_slam/CloudRegistration.hpp:25).
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

SENSOR_HEIGHT = 1.5
ROOM_HALF = 30.0
ROOM_HEIGHT = 10.0
CYL_RADIUS = 0.5
N_CYL = 12

SEED_SCENE = 1234
SEED_MAP = 1235
SEED_SCAN = 1236


@dataclasses.dataclass
class Scene:
    """Box room + cylinders, expressed in the MAP frame (z shifted by -1.5)."""

    cyl_xy: np.ndarray  # (12, 2)
    zmin: float = -SENSOR_HEIGHT
    zmax: float = ROOM_HEIGHT - SENSOR_HEIGHT
    half: float = ROOM_HALF
    cyl_r: float = CYL_RADIUS


def make_scene(seed: int = SEED_SCENE) -> Scene:
    rng = np.random.default_rng(seed)
    pts = []
    # rejection: keep cylinders >= 4 m from the origin-centred sensor path start
    while len(pts) < N_CYL:
        xy = rng.uniform(-25.0, 25.0, size=2)
        if np.hypot(xy[0], xy[1]) < 4.0:
            continue
        pts.append(xy)
    return Scene(cyl_xy=np.asarray(pts, dtype=np.float64))


def rpy_to_R(roll: float, pitch: float, yaw: float) -> np.ndarray:
    """R = Rz(yaw) * Ry(pitch) * Rx(roll) (same Euler order Open3D's
    TransformVector6dToMatrix4d uses)."""
    cr, sr = math.cos(roll), math.sin(roll)
    cp, sp = math.cos(pitch), math.sin(pitch)
    cy, sy = math.cos(yaw), math.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]], dtype=np.float64)
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]], dtype=np.float64)
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]], dtype=np.float64)
    return Rz @ Ry @ Rx


def make_pose(t, rpy_deg) -> np.ndarray:
    T = np.eye(4)
    T[:3, :3] = rpy_to_R(*[math.radians(a) for a in rpy_deg])
    T[:3, 3] = t
    return T


def ground_truth_pose() -> np.ndarray:
    """T_gt of SURVEY §8d: trans (0.30,-0.20,0.05) m o rpy (0.5,-0.5,2.0) deg."""
    return make_pose([0.30, -0.20, 0.05], [0.5, -0.5, 2.0])


def sample_map(scene: Scene, n_points: int = 1_000_000, seed: int = SEED_MAP):
    """Area-weighted uniform samples on all scene surfaces + analytic normals."""
    rng = np.random.default_rng(seed)
    h = scene.half
    H = scene.zmax - scene.zmin
    areas = np.array(
        [4 * h * h, 4 * h * h]  # floor, ceiling
        + [2 * h * H] * 4  # walls
        + [2 * math.pi * scene.cyl_r * H] * len(scene.cyl_xy)
    )
    face = rng.choice(len(areas), size=n_points, p=areas / areas.sum())
    u = rng.uniform(0.0, 1.0, size=n_points)
    v = rng.uniform(0.0, 1.0, size=n_points)
    pts = np.empty((n_points, 3))
    nrm = np.zeros((n_points, 3))
    a = (2 * u - 1) * h
    b = (2 * v - 1) * h
    z = scene.zmin + v * H
    m = face == 0  # floor, normal +z
    pts[m] = np.stack([a[m], b[m], np.full(m.sum(), scene.zmin)], 1)
    nrm[m] = [0, 0, 1]
    m = face == 1  # ceiling, normal -z
    pts[m] = np.stack([a[m], b[m], np.full(m.sum(), scene.zmax)], 1)
    nrm[m] = [0, 0, -1]
    m = face == 2  # wall x=-h, normal +x
    pts[m] = np.stack([np.full(m.sum(), -h), a[m], z[m]], 1)
    nrm[m] = [1, 0, 0]
    m = face == 3  # wall x=+h
    pts[m] = np.stack([np.full(m.sum(), h), a[m], z[m]], 1)
    nrm[m] = [-1, 0, 0]
    m = face == 4  # wall y=-h
    pts[m] = np.stack([a[m], np.full(m.sum(), -h), z[m]], 1)
    nrm[m] = [0, 1, 0]
    m = face == 5  # wall y=+h
    pts[m] = np.stack([a[m], np.full(m.sum(), h), z[m]], 1)
    nrm[m] = [0, -1, 0]
    for c, (cx, cy) in enumerate(scene.cyl_xy):
        m = face == 6 + c
        ang = 2 * math.pi * u[m]
        ca, sa = np.cos(ang), np.sin(ang)
        pts[m] = np.stack([cx + scene.cyl_r * ca, cy + scene.cyl_r * sa, z[m]], 1)
        nrm[m] = np.stack([ca, sa, np.zeros(m.sum())], 1)
    return np.ascontiguousarray(pts), np.ascontiguousarray(nrm)


def _ray_dirs(n_rings: int, n_az: int, fov_deg: float) -> np.ndarray:
    elev = np.radians(np.linspace(-fov_deg, fov_deg, n_rings))
    az = np.arange(n_az) * (2 * math.pi / n_az)
    ce, se = np.cos(elev), np.sin(elev)
    d = np.empty((n_rings, n_az, 3))
    d[..., 0] = ce[:, None] * np.cos(az)[None, :]
    d[..., 1] = ce[:, None] * np.sin(az)[None, :]
    d[..., 2] = se[:, None]
    return d.reshape(-1, 3)


def cast_scan(scene: Scene, pose: np.ndarray, n_rings: int, n_az: int, fov_deg: float,
              noise_sigma: float = 0.01, seed: int = SEED_SCAN) -> np.ndarray:
    """Ray-cast a spinning LiDAR at `pose` (map <- sensor); returns points in the
    SENSOR frame (n_rings*n_az, 3) with Gaussian range noise."""
    d_s = _ray_dirs(n_rings, n_az, fov_deg)
    R, o = pose[:3, :3], pose[:3, 3]
    d = d_s @ R.T
    n = d.shape[0]
    tbest = np.full(n, np.inf)
    h = scene.half
    with np.errstate(divide="ignore", invalid="ignore"):
        for axis, val in ((0, -h), (0, h), (1, -h), (1, h), (2, scene.zmin), (2, scene.zmax)):
            t = (val - o[axis]) / d[:, axis]
            t = np.where((t > 1e-9) & np.isfinite(t), t, np.inf)
            # closed convex room seen from inside: the nearest positive plane hit is the hit
            tbest = np.minimum(tbest, t)
        for cx, cy in scene.cyl_xy:
            ox, oy = o[0] - cx, o[1] - cy
            A = d[:, 0] ** 2 + d[:, 1] ** 2
            B = 2 * (ox * d[:, 0] + oy * d[:, 1])
            C = ox * ox + oy * oy - scene.cyl_r ** 2
            disc = B * B - 4 * A * C
            ok = (disc >= 0) & (A > 1e-12)
            sq = np.sqrt(np.where(ok, disc, 0.0))
            t = np.where(ok, (-B - sq) / (2 * A), np.inf)
            zhit = o[2] + t * d[:, 2]
            t = np.where((t > 1e-9) & (zhit >= scene.zmin) & (zhit <= scene.zmax), t, np.inf)
            tbest = np.minimum(tbest, t)
    rng = np.random.default_rng(seed)
    rng_noise = rng.normal(0.0, noise_sigma, size=n) if noise_sigma > 0 else 0.0
    return np.ascontiguousarray(d_s * (tbest + rng_noise)[:, None])


def vlp16_scan(scene: Scene, pose: np.ndarray, frame: int = 0, noise_sigma: float = 0.01,
               n_az: int = 4096) -> np.ndarray:
    """16 rings, elevation -15..+15 deg step 2 deg, 4096 azimuth steps = 65 536 pts."""
    return cast_scan(scene, pose, 16, n_az, 15.0, noise_sigma, SEED_SCAN + frame)


def os128_scan(scene: Scene, pose: np.ndarray, frame: int = 0, noise_sigma: float = 0.01,
               n_az: int = 1024) -> np.ndarray:
    """128 rings, +-22.5 deg, 1024 azimuth steps = 131 072 pts."""
    return cast_scan(scene, pose, 128, n_az, 22.5, noise_sigma, SEED_SCAN + frame)


def figure_eight_poses(n_frames: int = 200, step: float = 0.1) -> np.ndarray:
    """C3 trajectory: figure-eight (Gerono lemniscate), ~step m/frame, yaw-following."""
    total = n_frames * step
    a = total / 6.1  # lemniscate of Gerono length ~ 6.1 a
    s = np.linspace(0.0, 2 * math.pi, n_frames, endpoint=False)
    x = a * np.sin(s)
    y = a * np.sin(s) * np.cos(s)
    dx = a * np.cos(s)
    dy = a * (np.cos(s) ** 2 - np.sin(s) ** 2)
    yaw = np.arctan2(dy, dx)
    poses = np.tile(np.eye(4), (n_frames, 1, 1))
    for i in range(n_frames):
        poses[i, :3, :3] = rpy_to_R(0.0, 0.0, float(yaw[i]))
        poses[i, :3, 3] = [x[i], y[i], 0.0]
    return poses


def config1_inputs(n_az: int = 4096):
    """C1 (BASELINE.json configs[0]): two VLP-16 scans 0.3 m apart."""
    scene = make_scene()
    a = vlp16_scan(scene, np.eye(4), frame=0, n_az=n_az)
    b = vlp16_scan(scene, make_pose([0.3, 0.0, 0.0], [0, 0, 0]), frame=1, n_az=n_az)
    return a, b


def config2_inputs(n_map: int = 1_000_000, n_az: int = 4096):
    """C2 (BASELINE.json configs[1]): 64k-pt scan (sensor frame, captured at
    T_gt) vs N-pt map with normals; init = identity, truth = T_gt."""
    scene = make_scene()
    tgt, nrm = sample_map(scene, n_map)
    src = vlp16_scan(scene, ground_truth_pose(), frame=0, n_az=n_az)
    return src, tgt, nrm, ground_truth_pose()


def se3_error(Ta: np.ndarray, Tb: np.ndarray):
    """(|dt| in m, rotation angle of dR in rad) between two 4x4 poses."""
    dT = np.linalg.inv(Ta) @ Tb
    dt = float(np.linalg.norm(dT[:3, 3]))
    R = dT[:3, :3]
    # sin(theta) from the skew part resolves small angles (acos of the trace is quantised at ~2e-8 rad near identity: one ulp of
    # the trace); acos only for large angles, where the skew part loses the branch
    s = float(np.linalg.norm([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])) / 2.0
    c = (np.trace(R) - 1.0) / 2.0
    if c > 0.5:
        return dt, float(math.asin(min(1.0, s)))
    return dt, float(math.acos(max(-1.0, min(1.0, c))))
