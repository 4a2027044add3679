"""ORACLE (test infrastructure only): Python driver of oracle/ba_oracle.c.  PARITY UNPINNED.

Restates, around the C solver, what the reference's BA call sites do on the CPU:

* ``batch_matrix_to_pycolmap`` problem construction (vggsfm/utils/tensor_to_pycolmap.py:16-160):
  a 3D point per track with >= 2 masked observations, no observations for a point with a
  coordinate >= 3000, one camera per frame or a single shared camera;
* ``pycolmap.bundle_adjustment`` = COLMAP 3.10 ``BundleAdjustmentController`` [published
  source, not in /root/reference]: delete negative-depth observations (a track of length
  <= 2 is deleted entirely), fix the pose of image 0 and the x translation of image 1, run
  Ceres with TRIVIAL loss, refine focal + extra params, keep the principal point fixed;
* ``Reconstruction.normalize(5.0, 0.1, 0.9, True)`` (vggsfm/utils/triangulation.py:1212-1218);
* ``pycolmap.pose_refinement`` = ``RefineAbsolutePose``: CauchyLoss(1), points constant,
  gradient_tolerance 1.0, 100 iterations (vggsfm/utils/triangulation.py:387,590).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Problem(ctypes.Structure):
    _fields_ = [("num_cams", ctypes.c_int32), ("num_pts", ctypes.c_int32), ("num_obs", ctypes.c_int32),
                ("num_intr", ctypes.c_int32), ("camera_model", ctypes.c_int32), ("refine_focal", ctypes.c_int32),
                ("refine_extra", ctypes.c_int32), ("loss", ctypes.c_int32), ("loss_scale", ctypes.c_double),
                ("cam_intr", ctypes.c_void_p), ("row_ptr", ctypes.c_void_p), ("obs_cam", ctypes.c_void_p),
                ("obs_uv", ctypes.c_void_p), ("cam_const", ctypes.c_void_p), ("intr_const", ctypes.c_void_p),
                ("pt_const", ctypes.c_void_p)]


class Options(ctypes.Structure):
    _fields_ = [("max_num_iterations", ctypes.c_int32), ("max_num_consecutive_invalid_steps", ctypes.c_int32),
                ("jacobi_scaling", ctypes.c_int32), ("function_tolerance", ctypes.c_double),
                ("gradient_tolerance", ctypes.c_double), ("parameter_tolerance", ctypes.c_double),
                ("initial_trust_region_radius", ctypes.c_double), ("max_trust_region_radius", ctypes.c_double),
                ("min_trust_region_radius", ctypes.c_double), ("min_lm_diagonal", ctypes.c_double),
                ("max_lm_diagonal", ctypes.c_double), ("min_relative_decrease", ctypes.c_double)]


class Iter(ctypes.Structure):
    _fields_ = [("iteration", ctypes.c_int32), ("cost", ctypes.c_double), ("cost_change", ctypes.c_double),
                ("gradient_max_norm", ctypes.c_double), ("step_norm", ctypes.c_double),
                ("relative_decrease", ctypes.c_double), ("radius", ctypes.c_double), ("successful", ctypes.c_int32)]


class Summary(ctypes.Structure):
    _fields_ = [("initial_cost", ctypes.c_double), ("final_cost", ctypes.c_double),
                ("num_iterations", ctypes.c_int32), ("num_successful_steps", ctypes.c_int32),
                ("num_unsuccessful_steps", ctypes.c_int32), ("termination", ctypes.c_int32),
                ("n_reduced", ctypes.c_int32), ("num_log", ctypes.c_int32)]


def build(force=False):
    so = os.path.join(_HERE, "_build", "libba_oracle.so")
    src = os.path.join(_HERE, "ba_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.bao_solve.restype = ctypes.c_int
        _LIB.bao_num_threads.restype = ctypes.c_int
    return _LIB


def ceres_options(max_num_iterations=100, function_tolerance=0.0, gradient_tolerance=1e-4,
                  parameter_tolerance=0.0):
    """COLMAP 3.10 BundleAdjustmentOptions.solver_options defaults + Ceres 2.x defaults."""
    return Options(max_num_iterations, 10, 1, function_tolerance, gradient_tolerance, parameter_tolerance,
                   1e4, 1e16, 1e-32, 1e-6, 1e32, 1e-3)


def prepare_ba_options():
    """vggsfm/utils/triangulation_helpers.py:626-635: tolerances x10, 50 iterations."""
    return ceres_options(50, 0.0 * 10, 1e-4 * 10, 0.0 * 10)


# ------------------------------------------------------------------ rotations
def rotmat_to_quat(R):
    """Eigen Quaternion(Matrix3) (x,y,z,w), then normalised (COLMAP normalises before BA)."""
    R = np.asarray(R, dtype=np.float64)
    out = np.zeros(R.shape[:-2] + (4,))
    for idx in np.ndindex(R.shape[:-2]):
        m = R[idx]
        t = m[0, 0] + m[1, 1] + m[2, 2]
        q = np.zeros(4)
        if t > 0:
            t = np.sqrt(t + 1.0)
            q[3] = 0.5 * t
            t = 0.5 / t
            q[0] = (m[2, 1] - m[1, 2]) * t
            q[1] = (m[0, 2] - m[2, 0]) * t
            q[2] = (m[1, 0] - m[0, 1]) * t
        else:
            i = 0
            if m[1, 1] > m[0, 0]:
                i = 1
            if m[2, 2] > m[i, i]:
                i = 2
            j, k = (i + 1) % 3, (i + 2) % 3
            t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
            q[i] = 0.5 * t
            t = 0.5 / t
            q[3] = (m[k, j] - m[j, k]) * t
            q[j] = (m[j, i] + m[i, j]) * t
            q[k] = (m[k, i] + m[i, k]) * t
        out[idx] = q / np.linalg.norm(q)
    return out


def quat_to_rotmat(q):
    """Eigen toRotationMatrix of (x,y,z,w)."""
    q = np.asarray(q, dtype=np.float64)
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    R = np.stack([1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx,
                  txz - twy, tyz + twx, 1 - (txx + tyy)], -1)
    return R.reshape(q.shape[:-1] + (3, 3))


# ------------------------------------------------------------------ bundle adjustment
MODEL = {"SIMPLE_PINHOLE": 0, "SIMPLE_RADIAL": 1}


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def solve_csr(cam_q, cam_t, intr, pts, cam_intr, row_ptr, obs_cam, obs_uv, camera_model, options,
              refine_focal=True, refine_extra=True, loss=0, loss_scale=1.0, cam_const=None, intr_const=None,
              pt_const=None, log_cap=256):
    """Raw entry: arrays are modified in place (float64 / int32 / uint8, C-contiguous)."""
    pb = Problem(len(cam_t), len(pts), len(obs_cam), len(intr), camera_model, int(refine_focal), int(refine_extra),
                 loss, loss_scale, _ptr(cam_intr), _ptr(row_ptr), _ptr(obs_cam), _ptr(obs_uv), _ptr(cam_const),
                 _ptr(intr_const), _ptr(pt_const))
    summ = Summary()
    log = (Iter * log_cap)()
    lib().bao_solve(ctypes.byref(pb), ctypes.byref(options), _ptr(cam_q), _ptr(cam_t), _ptr(intr), _ptr(pts),
                    ctypes.byref(summ), log, log_cap)
    its = [dict(iteration=l.iteration, cost=l.cost, cost_change=l.cost_change, gradient_max_norm=l.gradient_max_norm,
                step_norm=l.step_norm, relative_decrease=l.relative_decrease, radius=l.radius,
                successful=bool(l.successful)) for l in log[:min(summ.num_log, log_cap)]]
    return dict(initial_cost=summ.initial_cost, final_cost=summ.final_cost, num_iterations=summ.num_iterations,
                num_successful_steps=summ.num_successful_steps, num_unsuccessful_steps=summ.num_unsuccessful_steps,
                termination=summ.termination, n_reduced=summ.n_reduced, iterations=its)


def build_observations(points3d, extrinsics, tracks, masks, max_points3D_val=3000, filter_negative_depth=True):
    """Problem construction of batch_matrix_to_pycolmap + the controller's negative-depth filter.
    Returns (point_index (P',) into the input tracks, row_ptr, obs_cam, obs_uv, deleted (P_valid,) bool,
    valid_idx)."""
    masks = np.asarray(masks, bool)
    valid_idx = np.nonzero(masks.sum(0) >= 2)[0]
    pts = np.asarray(points3d, np.float64)[valid_idx]
    m = masks[:, valid_idx].copy()
    m[:, ~(pts < max_points3D_val).all(-1)] = False
    # FilterObservationsWithNegativeDepth, frame-major order as COLMAP iterates images
    z = np.einsum("sj,pj->sp", extrinsics[:, 2, :3], pts) + extrinsics[:, 2, 3][:, None]
    deleted = np.zeros(len(valid_idx), bool)
    length = m.sum(0)
    eps = np.finfo(np.float64).eps
    for s in range(m.shape[0] if filter_negative_depth else 0):
        bad = np.nonzero(m[s] & ~(z[s] >= eps) & ~deleted)[0]
        for p in bad:
            if length[p] <= 2:
                deleted[p] = True
                m[:, p] = False
                length[p] = 0
            else:
                m[s, p] = False
                length[p] -= 1
    counts = m.sum(0)
    row_ptr = np.zeros(len(valid_idx) + 1, np.int32)
    row_ptr[1:] = np.cumsum(counts)
    pp, ss = np.nonzero(m.T)                  # point-major, frames ascending
    obs_cam = ss.astype(np.int32)
    uv = np.asarray(tracks)[ss, valid_idx[pp]].astype(np.float64)
    return valid_idx, row_ptr, obs_cam, np.ascontiguousarray(uv), deleted


def bundle_adjustment(points3d, extrinsics, intrinsics, tracks, masks, extra_params=None, shared_camera=False,
                      camera_type="SIMPLE_PINHOLE", options=None, normalize=False, constant_points=None,
                      constant_pose_frames=None, filter_negative_depth=True, refine_focal=True, refine_extra=True,
                      loss=0, loss_scale=1.0):
    """What `batch_matrix_to_pycolmap -> pycolmap.bundle_adjustment [-> normalize] ->
    pycolmap_to_batch_matrix` computes.  Returns (points3D_opt (P_valid,3) with zero rows for deleted
    points, extrinsics (S,3,4), intrinsics (S,3,3), extra_params (S,1)|None, summary)."""
    if camera_type not in MODEL:
        raise ValueError(f"Camera type {camera_type} is not supported yet")
    options = options or ceres_options()
    S = len(extrinsics)
    extrinsics = np.asarray(extrinsics, np.float64)
    intrinsics = np.asarray(intrinsics, np.float64)
    valid_idx, row_ptr, obs_cam, obs_uv, deleted = build_observations(points3d, extrinsics, tracks, masks,
                                                                      filter_negative_depth=filter_negative_depth)
    pts = np.ascontiguousarray(np.asarray(points3d, np.float64)[valid_idx])
    cam_q = np.ascontiguousarray(rotmat_to_quat(extrinsics[:, :, :3]))
    cam_t = np.ascontiguousarray(extrinsics[:, :, 3])
    n_intr = 1 if shared_camera else S
    intr = np.zeros((n_intr, 4))
    src = slice(0, 1) if shared_camera else slice(None)
    intr[:, 0] = intrinsics[src, 0, 0]
    intr[:, 1] = intrinsics[src, 0, 2]
    intr[:, 2] = intrinsics[src, 1, 2]
    if camera_type == "SIMPLE_RADIAL":
        intr[:, 3] = np.asarray(extra_params, np.float64)[src, 0]
    cam_intr = (np.zeros(S, np.int32) if shared_camera else np.arange(S, dtype=np.int32))
    cam_const = np.zeros(S, np.uint8)
    if constant_pose_frames is None:
        cam_const[0] = 1            # SetConstantCamPose(image 0)
        if S > 1:
            cam_const[1] = 2        # SetConstantCamPositions(image 1, {0})
    else:                           # explicit BundleAdjustmentConfig (video_runner.py:813-829)
        cam_const[list(constant_pose_frames)] = 1
    pt_const = None if constant_points is None else np.ascontiguousarray(
        np.asarray(constant_points, bool)[valid_idx].astype(np.uint8))
    summary = solve_csr(cam_q, cam_t, intr, pts, cam_intr, row_ptr, obs_cam, obs_uv, MODEL[camera_type], options,
                        refine_focal=refine_focal, refine_extra=refine_extra, loss=loss, loss_scale=loss_scale,
                        cam_const=cam_const, pt_const=pt_const)
    ext = np.concatenate([quat_to_rotmat(cam_q), cam_t[:, :, None]], -1)
    pts[deleted] = 0.0
    if normalize:
        ext, pts = normalize_reconstruction(ext, pts, ~deleted)
    K = np.zeros((S, 3, 3))
    K[:, 0, 0] = K[:, 1, 1] = intr[cam_intr, 0]
    K[:, 0, 2] = intr[cam_intr, 1]
    K[:, 1, 2] = intr[cam_intr, 2]
    K[:, 2, 2] = 1.0
    extra = intr[cam_intr, 3][:, None].copy() if camera_type == "SIMPLE_RADIAL" else None
    summary["valid_idx"] = valid_idx
    summary["deleted"] = deleted
    return pts, ext, K, extra, summary


def normalize_reconstruction(ext, pts, pts_alive=None, extent=5.0, p0=0.1, p1=0.9):
    """COLMAP 3.10 Reconstruction::Normalize(extent, p0, p1, use_images=True) [published source]:
    robust bbox of the camera centres (per-axis sorted, float32 coordinates), scale to `extent`,
    translate the centroid of the kept range to the origin."""
    S = len(ext)
    if S < 2:
        return ext, pts
    centers = -np.einsum("sji,sj->si", ext[:, :, :3], ext[:, :, 3])
    c32 = np.sort(centers.astype(np.float32), axis=0)
    P0 = int(p0 * (S - 1)) if S > 3 else 0
    P1 = int(p1 * (S - 1)) if S > 3 else S - 1
    bmin = c32[P0].astype(np.float64)
    bmax = c32[P1].astype(np.float64)
    mean = c32[P0:P1 + 1].astype(np.float64).sum(0) / (P1 - P0 + 1)
    old_extent = np.linalg.norm(bmax - bmin)
    scale = 1.0 if old_extent < np.finfo(np.float64).eps else extent / old_extent
    ext = ext.copy()
    # Sim3(scale, I, -scale*mean): X' = s (X - mean); cam_from_world' : R' = R, t' = s (t + R mean)
    ext[:, :, 3] = scale * (ext[:, :, 3] + np.einsum("sij,j->si", ext[:, :, :3], mean))
    pts = pts.copy()
    alive = np.ones(len(pts), bool) if pts_alive is None else pts_alive
    pts[alive] = scale * (pts[alive] - mean)
    return ext, pts


def pose_refinement(extrinsic, points2D, points3D, inlier_mask, intr_params, camera_type="SIMPLE_PINHOLE",
                    refine_focal_length=True, refine_extra_params=True):
    """pycolmap.pose_refinement (RefineAbsolutePose): one camera, constant points, CauchyLoss(1).
    Returns (extrinsic (3,4), intr_params (4,), summary).  COLMAP weights each residual by
    loss only; covariance weighting is off."""
    sel = np.nonzero(np.asarray(inlier_mask, bool))[0]
    n = len(sel)
    cam_q = np.ascontiguousarray(rotmat_to_quat(extrinsic[None, :, :3]))
    cam_t = np.ascontiguousarray(extrinsic[None, :, 3].astype(np.float64))
    intr = np.zeros((1, 4))
    intr[0, :len(intr_params)] = intr_params
    pts = np.ascontiguousarray(np.asarray(points3D, np.float64)[sel])
    row_ptr = np.arange(n + 1, dtype=np.int32)
    obs_cam = np.zeros(n, np.int32)
    obs_uv = np.ascontiguousarray(np.asarray(points2D, np.float64)[sel])
    opt = ceres_options(100, 1e-6, 1.0, 1e-8)      # Ceres defaults + COLMAP gradient_tolerance=1.0
    summary = solve_csr(cam_q, cam_t, intr, pts, np.zeros(1, np.int32), row_ptr, obs_cam, obs_uv, MODEL[camera_type],
                        opt, refine_focal=refine_focal_length, refine_extra=refine_extra_params, loss=1,
                        loss_scale=1.0, pt_const=np.ones(n, np.uint8))
    ext = np.concatenate([quat_to_rotmat(cam_q)[0], cam_t[0][:, None]], -1)
    return ext, intr[0].copy(), summary
