"""TEST INFRASTRUCTURE: run ``vggsfm_amd.pycolmap_compat`` without a GPU by swapping the device solvers for the CPU
oracle (``oracle/ba.py``, ``oracle/geometry.py``) through pytest's monkeypatch.

The build container has the reference tree but no GPU; the GPU box has a GPU but no reference tree.  What needs the
reference's UNMODIFIED code on top of the compat object model (its runner tail, its ``Triangulator.forward``, its
``tensor_to_pycolmap``) is therefore exercised here with the oracle as the solver -- the object model, the problem
construction and the read-back are the product's; the arithmetic is the checker's -- while ``-m gpu`` tests run the same
compat entry points on the device solvers and compare them with the oracle on identical problems
(tests/test_gpu_pycolmap_compat.py).  Nothing under ``vggsfm_amd/`` imports this module or ``oracle/``.
"""
import numpy as np
import torch

from oracle import ba as OB
from oracle import geometry as OG


def _n(t):
    return None if t is None else (t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t))


def _t(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a))


LOSS = {"TRIVIAL": 0, "CAUCHY": 1, "HUBER": 2, "SOFT_L1": 3}


def bundle_adjustment(points3d, extrinsics, intrinsics, tracks, masks, image_size=None, extra_params=None,
                      shared_camera=False, camera_type="SIMPLE_PINHOLE", options=None, normalize=False,
                      constant_points=None, constant_pose_frames=None, filter_negative_depth=True):
    """Signature of ``vggsfm_amd.ba.bundle_adjustment``; arithmetic by oracle/ba_oracle.c."""
    from vggsfm_amd.ba_options import BundleAdjustmentOptions
    options = options or BundleAdjustmentOptions()
    so = options.solver_options
    o = OB.ceres_options(so.max_num_iterations, so.function_tolerance, so.gradient_tolerance, so.parameter_tolerance)
    pts, ext, K, extra, summ = OB.bundle_adjustment(
        _n(points3d), _n(extrinsics), _n(intrinsics), _n(tracks), _n(masks).astype(bool), _n(extra_params), shared_camera,
        camera_type, options=o, normalize=normalize, constant_points=_n(constant_points),
        constant_pose_frames=constant_pose_frames, filter_negative_depth=filter_negative_depth,
        refine_focal=options.refine_focal_length, refine_extra=options.refine_extra_params,
        loss=LOSS[options.loss_function_type], loss_scale=options.loss_function_scale)
    summ["valid_idx"], summ["deleted"] = _t(summ["valid_idx"]), _t(summ["deleted"])
    return _t(pts), _t(ext), _t(K), _t(extra), summ


def pose_refinement_batch(extrinsics, intr_params, points2D, points3D, inlier_mask, frame_ids, camera_type,
                          refine_flags, refopts=None):
    """Signature of ``vggsfm_amd.pose.pose_refinement_batch``; one oracle solve per listed frame."""
    ext, intr = _n(extrinsics).astype(np.float64).copy(), _n(intr_params).astype(np.float64).copy()
    p2, p3, mk, rf = _n(points2D), _n(points3D), _n(inlier_mask).astype(bool), _n(refine_flags)
    sums = []
    for f in [int(i) for i in (frame_ids.tolist() if torch.is_tensor(frame_ids) else frame_ids)]:
        k = 4 if camera_type == "SIMPLE_RADIAL" else 3
        e, p, s = OB.pose_refinement(ext[f], p2[f], p3, mk[f], intr[f, :k], camera_type,
                                     refine_focal_length=bool(rf[f] & 1), refine_extra_params=bool(rf[f] & 2))
        ext[f], intr[f, :k] = e, p[:k]
        sums.append(dict(frame=f, **{key: s[key] for key in ("initial_cost", "final_cost", "num_iterations", "termination")}))
    return _t(ext), _t(intr), sums


def absolute_pose_estimation_batch(extrinsics, intr_params, points2D, points3D, candidate_mask, frame_ids, *a, **k):
    """Every estimate "fails" (pycolmap returns None), as in the harness that produced the golden vectors
    (oracle/pycolmap_shim.py): the estimator draws from COLMAP's RNG, there is nothing to pin."""
    S, P = candidate_mask.shape
    z = torch.zeros(S, dtype=torch.bool)
    return extrinsics, intr_params, z, z.to(torch.int32), torch.zeros((S, P), dtype=torch.bool)


def observation_filter(points3D, extrinsics, intrinsics, extra_params, tracks, masks, max_reproj_error, min_tri_angle):
    """Signature of ``vggsfm_amd.video.observation_filter``; the two passes by oracle/geometry.py."""
    pts, ext, K, xp = _n(points3D), _n(extrinsics), _n(intrinsics), _n(extra_params)
    tr, mk = _n(tracks).astype(np.float64), _n(masks).astype(bool)
    _, detail = OG.filter_all_points3D(pts, tr, ext, K, xp, max_reproj_error=max_reproj_error, check_triangle=False,
                                       return_detail=True, hard_max=-1)
    inl = mk & detail
    keep, _ = OG.filter_all_points3D(pts, np.where(inl[..., None], tr, 1e9), ext, K, xp, max_reproj_error=max_reproj_error,
                                     min_tri_angle=min_tri_angle, check_triangle=True, hard_max=-1)
    keep = keep & (inl.sum(0) >= 2)
    return _t(inl & keep[None]), _t(keep)


def patch(monkeypatch):
    """compat entries -> CPU tensors + oracle arithmetic, for the duration of one test."""
    import vggsfm_amd.ba as BA
    import vggsfm_amd.pose as POSE
    import vggsfm_amd.pycolmap_compat as C
    import vggsfm_amd.video as V
    monkeypatch.setattr(C, "DEVICE", "cpu")
    monkeypatch.setattr(BA, "bundle_adjustment", bundle_adjustment)
    monkeypatch.setattr(POSE, "pose_refinement_batch", pose_refinement_batch)
    monkeypatch.setattr(POSE, "absolute_pose_estimation_batch", absolute_pose_estimation_batch)
    monkeypatch.setattr(V, "observation_filter", observation_filter)


# ---------------------------------------------------------------------------------------------------------------------
# The geometry kernels behind ``vggsfm_amd.video.VideoGeometry`` (geometry.hip, triangulate.hip), for the CPU rehearsal
# of the video golden (tests/test_video_golden_cpu.py): the DRIVER -- window bounds, shrink / step-back, table updates --
# is the product's, the arithmetic is oracle/geometry.py.  The -m gpu twin (tests/test_gpu_video_golden.py) runs the
# same driver on the device kernels.
def filter_all_points3D(points3D, points2D, extrinsics, intrinsics, extra_params=None, max_reproj_error=4,
                        min_tri_angle=1.5, check_triangle=True, return_detail=False, hard_max=100, max_points_num=1000000):
    out = OG.filter_all_points3D(_n(points3D).astype(np.float64), _n(points2D).astype(np.float64),
                                 _n(extrinsics).astype(np.float64), _n(intrinsics).astype(np.float64),
                                 None if extra_params is None else _n(extra_params).astype(np.float64),
                                 max_reproj_error=max_reproj_error, min_tri_angle=min_tri_angle,
                                 check_triangle=check_triangle, return_detail=return_detail, hard_max=hard_max)
    return tuple(_t(o) for o in out) if isinstance(out, tuple) else _t(out)


def cam_from_img(pred_tracks, intrinsics, extra_params=None):
    return _t(OG.cam_from_img(_n(pred_tracks).astype(np.float64), _n(intrinsics).astype(np.float64),
                              None if extra_params is None else _n(extra_params).astype(np.float64)))


def triangulate_tracks(extrinsics, tracks_normalized, max_ransac_iters=256, lo_num=50, max_angular_error=2,
                       min_tri_angle=1.5, track_vis=None, track_score=None, max_tri_points_num=819200, **kw):
    S, N = tracks_normalized.shape[:2]
    pairs = OG.generate_combinations(S)
    assert len(pairs) <= max_ransac_iters and S * N <= max_tri_points_num, "one chunk, all pairs (no RNG draw) only"
    pts, num, mask = OG.triangulate_tracks_chunk(_n(extrinsics).astype(np.float64), _n(tracks_normalized).astype(np.float64),
                                                 pairs, lo_num, max_angular_error, min_tri_angle,
                                                 None if track_vis is None else _n(track_vis),
                                                 None if track_score is None else _n(track_score))
    return _t(pts), _t(num.astype(np.int64)), _t(mask)


def patch_video_geometry(monkeypatch):
    """`patch` + the geometry entry points ``vggsfm_amd.video`` holds by name."""
    import vggsfm_amd._lib as LIB
    import vggsfm_amd.video as V
    patch(monkeypatch)
    monkeypatch.setattr(LIB, "require_gpu", lambda *t: None)
    monkeypatch.setattr(V, "filter_all_points3D", filter_all_points3D)
    monkeypatch.setattr(V, "cam_from_img", cam_from_img)
    monkeypatch.setattr(V, "triangulate_tracks", triangulate_tracks)
    monkeypatch.setattr(V, "pose_refinement_batch", pose_refinement_batch)
    monkeypatch.setattr(V, "absolute_pose_estimation_batch", absolute_pose_estimation_batch)
