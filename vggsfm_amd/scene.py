"""Deterministic synthetic SfM scenes used by tests and bench.py.

The reference ships no datasets for the triangulation / BA path, so every workload in
BASELINE.json's ``configs`` is synthesised here (numpy only, seeded, identical on every
machine).  Conventions follow the reference pipeline exactly (OpenCV camera frame,
``x_cam = R X + t``; pixel coordinates of a 1024x1024 padded image, principal point at the
centre; ``intrinsics = [[f,0,cx],[0,f,cy],[0,0,1]]``; tracks/vis/score float32,
cameras/points float64 -- see ``vggsfm/models/triangulator.py:82-114`` and
``vggsfm/utils/tensor_to_pycolmap.py:184,206-211`` in the reference).
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

IMAGE_SIZE = 1024


@dataclass
class Scene:
    """One synthetic scene in the reference's dense (frames x tracks) layout."""

    extrinsics: np.ndarray          # (S,3,4) f64 ground truth [R|t]
    intrinsics: np.ndarray          # (S,3,3) f64 ground truth K
    extra_params: Optional[np.ndarray]  # (S,1) f64 radial k, or None (SIMPLE_PINHOLE)
    points3D: np.ndarray            # (N,3) f64 ground truth
    tracks: np.ndarray              # (S,N,2) f32 pixel observations (noise + outliers)
    vis: np.ndarray                 # (S,N)  f32 in {0,1}
    score: np.ndarray               # (S,N)  f32 (all ones)
    mask: np.ndarray                # (S,N)  bool  visible
    outlier: np.ndarray             # (S,N)  bool  visible and replaced by an outlier
    camera_type: str = "SIMPLE_PINHOLE"
    shared_camera: bool = False
    image_size: int = IMAGE_SIZE
    meta: dict = field(default_factory=dict)

    @property
    def S(self):
        return self.tracks.shape[0]

    @property
    def N(self):
        return self.tracks.shape[1]


def _look_at(center, target):
    z = target - center
    z = z / np.linalg.norm(z)
    x = np.cross(np.array([0.0, 1.0, 0.0]), z)
    x = x / np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z], axis=0)
    t = -R @ center
    return R, t


def make_cameras(S, camera_type="SIMPLE_PINHOLE", shared_camera=False, seed=0, arc_deg=60.0,
                 radius=4.0, focal=1000.0, radial_k=0.05):
    """Cameras on an arc of `radius` around (0,0,4); camera 0 is the identity pose."""
    rng = np.random.Generator(np.random.PCG64(seed))
    target = np.array([0.0, 0.0, radius])
    ext = np.zeros((S, 3, 4))
    for i in range(S):
        th = np.deg2rad(arc_deg) * (i / max(S - 1, 1))
        c = target + radius * np.array([np.sin(th), 0.3 * np.sin(2.0 * th) * np.sin(th), -np.cos(th)])
        R, t = _look_at(c, target)
        ext[i, :, :3] = R
        ext[i, :, 3] = t
    if shared_camera:
        f = np.full(S, focal)
    else:
        f = focal * (1.0 + 0.05 * rng.uniform(-1.0, 1.0, size=S))
    K = np.zeros((S, 3, 3))
    K[:, 0, 0] = f
    K[:, 1, 1] = f
    K[:, 0, 2] = IMAGE_SIZE / 2.0
    K[:, 1, 2] = IMAGE_SIZE / 2.0
    K[:, 2, 2] = 1.0
    extra = None
    if camera_type == "SIMPLE_RADIAL":
        extra = np.full((S, 1), radial_k)
    elif camera_type != "SIMPLE_PINHOLE":
        raise ValueError(f"Camera type {camera_type} is not supported yet")
    return ext, K, extra


def project(points3D, ext, K, extra):
    """Exact COLMAP projection (S,N,2) f64 and depth (S,N)."""
    Xc = np.einsum("sij,nj->sni", ext[:, :, :3], points3D) + ext[:, None, :, 3]
    z = Xc[..., 2]
    u = Xc[..., 0] / z
    v = Xc[..., 1] / z
    if extra is not None:
        r2 = u * u + v * v
        d = 1.0 + extra[:, 0][:, None] * r2
        u = u * d
        v = v * d
    x = K[:, 0, 0][:, None] * u + K[:, 0, 2][:, None]
    y = K[:, 1, 1][:, None] * v + K[:, 1, 2][:, None]
    return np.stack([x, y], axis=-1), z


def make_scene(S, N, camera_type="SIMPLE_PINHOLE", shared_camera=False, seed=0, noise_px=0.5,
               outlier_frac=0.05, full_visibility=False, track_seed=None):
    """Scene of SURVEY §8d.  `track_seed` lets several ranks share cameras (seed) but draw
    disjoint track shards."""
    ext, K, extra = make_cameras(S, camera_type, shared_camera, seed)
    rng = np.random.Generator(np.random.PCG64(seed + 1 if track_seed is None else track_seed))
    pts = np.array([0.0, 0.0, 4.0]) + rng.uniform(-1.0, 1.0, size=(N, 3))
    lo = max(3, int(np.ceil(0.1 * S)))
    hi = max(lo, int(np.ceil(0.4 * S)))
    if full_visibility or S <= 3:
        start = np.zeros(N, dtype=np.int64)
        length = np.full(N, S, dtype=np.int64)
    else:
        length = rng.integers(lo, hi + 1, size=N)
        start = (rng.uniform(0.0, 1.0, size=N) * (S - length + 1)).astype(np.int64)
    frames = np.arange(S)[:, None]
    mask = (frames >= start[None]) & (frames < (start + length)[None])
    uv, _ = project(pts, ext, K, extra)
    uv = uv + rng.normal(0.0, noise_px, size=uv.shape)
    outlier = (rng.uniform(0.0, 1.0, size=mask.shape) < outlier_frac) & mask
    off = rng.uniform(-50.0, 50.0, size=uv.shape)
    uv = np.where(outlier[..., None], uv + off, uv)
    tracks = uv.astype(np.float32)
    vis = mask.astype(np.float32)
    score = np.ones_like(vis)
    return Scene(ext, K, extra, pts, tracks, vis, score, mask, outlier, camera_type, shared_camera,
                 meta=dict(seed=seed, noise_px=noise_px, outlier_frac=outlier_frac))


def _rodrigues(w):
    th = np.linalg.norm(w)
    if th < 1e-16:
        return np.eye(3)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def perturb_for_ba(scene, seed=0, rot_deg=0.5, trans=0.02, focal_rel=0.02, point=0.02):
    """Initial state for BA-only runs: perturbed GT cameras (camera 0 untouched) and points."""
    rng = np.random.Generator(np.random.PCG64(seed + 1000))
    S = scene.S
    ext = scene.extrinsics.copy()
    K = scene.intrinsics.copy()
    for i in range(1, S):
        dR = _rodrigues(np.deg2rad(rot_deg) * rng.normal(size=3))
        ext[i, :, :3] = dR @ ext[i, :, :3]
        ext[i, :, 3] = ext[i, :, 3] + trans * rng.normal(size=3)
    if scene.shared_camera:
        fs = 1.0 + focal_rel * rng.normal()
        K[:, 0, 0] *= fs
        K[:, 1, 1] *= fs
    else:
        fs = 1.0 + focal_rel * rng.normal(size=S)
        K[:, 0, 0] *= fs
        K[:, 1, 1] *= fs
    extra = None if scene.extra_params is None else scene.extra_params.copy()
    pts = scene.points3D + point * rng.normal(size=scene.points3D.shape)
    return ext, K, extra, pts


def make_scene_device(S, N, camera_type="SIMPLE_PINHOLE", shared_camera=False, seed=0, noise_px=0.5, outlier_frac=0.05,
                      track_seed=None, device="cuda", point_noise=0.02):
    """The same scene model drawn with torch on the device (the dense (S,N) random fields of ``make_scene`` cost ~25 s of
    numpy per 100k tracks at 400 frames): cameras from ``make_cameras`` (identical), points / visibility windows /
    pixel noise / outliers from a torch generator -- the same distributions, a DIFFERENT random stream, so this is
    for throughput workloads (bench.py's 400-frame configurations), not for parity fixtures.
    Returns a namespace: extrinsics / intrinsics / extra_params (numpy, ground truth), points3D (N,3) f64, tracks (S,N,2)
    f32, mask (S,N) bool, points3D_init (N,3) f64 = ground truth + N(0, point_noise) -- tensors on `device`."""
    import types

    import torch
    ext, K, extra = make_cameras(S, camera_type, shared_camera, seed)
    g = torch.Generator(device=device)
    g.manual_seed(int(seed + 1 if track_seed is None else track_seed))
    dev = torch.device(device)
    U = lambda *shape: torch.rand(*shape, generator=g, device=dev, dtype=torch.float64)
    pts = torch.tensor([0.0, 0.0, 4.0], device=dev, dtype=torch.float64) + (2.0 * U(N, 3) - 1.0)
    lo = max(3, int(np.ceil(0.1 * S)))
    hi = max(lo, int(np.ceil(0.4 * S)))
    length = torch.randint(lo, hi + 1, (N,), generator=g, device=dev)
    start = (U(N) * (S - length + 1).double()).long()
    frames = torch.arange(S, device=dev)[:, None]
    mask = (frames >= start[None]) & (frames < (start + length)[None])
    E, Kt = torch.from_numpy(ext).to(dev), torch.from_numpy(K).to(dev)
    Xc = torch.einsum("sij,nj->sni", E[:, :, :3], pts) + E[:, None, :, 3]
    u, v = Xc[..., 0] / Xc[..., 2], Xc[..., 1] / Xc[..., 2]
    if extra is not None:
        d = 1.0 + torch.from_numpy(extra).to(dev)[:, 0][:, None] * (u * u + v * v)
        u, v = u * d, v * d
    uv = torch.stack([Kt[:, 0, 0][:, None] * u + Kt[:, 0, 2][:, None], Kt[:, 1, 1][:, None] * v + Kt[:, 1, 2][:, None]], -1)
    del Xc, u, v
    uv += noise_px * torch.randn(uv.shape, generator=g, device=dev, dtype=torch.float32)
    outlier = (torch.rand(mask.shape, generator=g, device=dev) < outlier_frac) & mask
    uv += outlier[..., None] * (100.0 * torch.rand(uv.shape, generator=g, device=dev, dtype=torch.float32) - 50.0)
    init = pts + point_noise * torch.randn(pts.shape, generator=g, device=dev, dtype=torch.float64)
    return types.SimpleNamespace(extrinsics=ext, intrinsics=K, extra_params=extra, points3D=pts, points3D_init=init,
                                 tracks=uv.to(torch.float32), mask=mask, camera_type=camera_type, shared_camera=shared_camera,
                                 S=S, N=N)
