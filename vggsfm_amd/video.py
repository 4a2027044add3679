"""Geometry of the sliding-window video path (``vggsfm/runners/video_runner.py``, SURVEY.md section 3.2 / 8a
"Video" row), as tensor functions on the device.  The learned parts (camera predictor, tracker) and the
``point_dict`` / ``frame_dict`` bookkeeping stay with the caller; what is here is every geometric step of
``VideoRunner.move_window`` and ``joint_BA``:

    align_camera_extrinsics / apply_transformation   utils/align.py:145-252   similarity alignment of the predicted
                                                                              cameras of the next window
    align_next_window                                video_runner.py:938-1017 pose-only refinement of frames 1..S-1
                                                                              on the carried-over 3D points
    filter_points_and_compute_masks                  video_runner.py:905-936  reprojection filter + track length
    triangulate_window_tracks                        video_runner.py:1189-1262 (after predict_tracks)
    window_bundle_adjustment                         video_runner.py:800-838  (vggsfm_amd.ba)
    joint_bundle_adjustment                          video_runner.py:494-541

All kernels underneath are the ones of the batch path (pose.hip, geometry.hip, triangulate.hip, ba.hip).
"""
import torch

from . import _lib
from . import ba as _ba
from .ba import window_bundle_adjustment  # noqa: F401  (re-export: the local BA of a window)
from .ba_options import AbsolutePoseEstimationOptions, AbsolutePoseRefinementOptions, BundleAdjustmentOptions
from .pose import absolute_pose_estimation_batch, pose_refinement_batch
from .utils.triangulation import _from_intr_params, _intr_params, triangulate_tracks
from .utils.triangulation_helpers import cam_from_img, filter_all_points3D


def align_camera_extrinsics(cameras_src, cameras_tgt, estimate_scale=True, eps=1e-9):
    """utils/align.py:145-205.  (B,3,4) x (B,3,4) -> (align_R (1,3,3), align_T (1,3), align_s): the similarity
    that maps the source cameras onto the target ones (OpenCV convention, x_cam = R X + t).  B is a window
    (<= 33 frames): plain torch ops, no kernel."""
    R_src, R_tgt = cameras_src[:, :, :3], cameras_tgt[:, :, :3]
    RRcov = torch.bmm(R_tgt.transpose(2, 1), R_src).mean(0)
    U, _, Vh = torch.linalg.svd(RRcov)
    align_R = Vh.transpose(0, 1) @ U.t()
    T_src, T_tgt = cameras_src[:, :, 3], cameras_tgt[:, :, 3]
    A = torch.bmm(T_src[:, None], R_src)[:, 0]
    B = torch.bmm(T_tgt[:, None], R_src)[:, 0]
    Amu, Bmu = A.mean(0, keepdim=True), B.mean(0, keepdim=True)
    if estimate_scale and A.shape[0] > 1:
        Ac, Bc = A - Amu, B - Bmu
        align_s = (Ac * Bc).mean() / (Ac ** 2).mean().clamp(eps)
    else:
        align_s = 1.0
    align_T = Bmu - align_s * Amu
    return align_R[None], align_T, align_s


def apply_transformation(cameras_src, align_R, align_T, align_s, return_extri=True):
    """utils/align.py:208-252."""
    R_src, T_src = cameras_src[:, :, :3], cameras_src[:, :, 3]
    aligned_R = torch.bmm(R_src, align_R.expand(R_src.shape[0], 3, 3))
    aligned_T = torch.bmm(R_src, align_T[..., None].repeat(R_src.shape[0], 1, 1))[..., 0] + T_src * align_s
    if return_extri:
        return torch.cat([aligned_R, aligned_T.unsqueeze(-1)], dim=-1)
    return aligned_R, aligned_T


def filter_points_and_compute_masks(points, tracks, extrinsics, intrinsics, extra_params=None, min_valid_track_length=3,
                                    max_reproj_error=4):
    """video_runner.py:905-936.  intrinsics (1,3,3) / extra_params (1,k) of the single video camera (or per frame).
    Returns (filtered_points, filtered_tracks, filtered_inlier_masks, valid_tracks_mask)."""
    S = extrinsics.shape[0]
    K = intrinsics.expand(S, -1, -1) if intrinsics.shape[0] == 1 else intrinsics
    ep = None if extra_params is None else (extra_params.expand(S, -1) if extra_params.shape[0] == 1 else extra_params)
    _, inlier_mask = filter_all_points3D(points, tracks, extrinsics, K, extra_params=ep, max_reproj_error=max_reproj_error,
                                         return_detail=True, hard_max=-1)
    valid_tracks_mask = inlier_mask.sum(dim=0) >= min_valid_track_length
    return points[valid_tracks_mask], tracks[:, valid_tracks_mask], inlier_mask[:, valid_tracks_mask], valid_tracks_mask


def align_next_window(extrinsics, tracks, inlier, points3D, intrinsics, extra_params=None, camera_type="SIMPLE_RADIAL",
                      min_vis_num=50, use_pnp=False, generator=None):
    """video_runner.py:938-1017: frame 0 keeps its pose; every other frame is refined on the fixed 3D points with
    focal length and distortion constant (CauchyLoss(1), gradient tolerance 1.0); a frame with <= min_vis_num inliers
    uses ALL points (`inlier_mask[:] = 1`).  use_pnp: the poses first come from absolute_pose_estimation on the inlier
    matches (P3P RANSAC, max_error 12 px; device restatement, random => parity unpinned).  All frames run concurrently."""
    _lib.require_gpu(extrinsics, tracks, inlier, points3D)
    S = extrinsics.shape[0]
    dev = tracks.device
    inl = inlier.bool().clone()
    few = inl.sum(dim=1) <= min_vis_num
    if bool(few[1:].any()):
        print("Too small inliers")
    inl[few] = True
    K = intrinsics.expand(S, -1, -1) if intrinsics.shape[0] == 1 else intrinsics
    ep = None if extra_params is None else (extra_params.expand(S, -1) if extra_params.shape[0] == 1 else extra_params)
    params = _intr_params(K.to(torch.float64), ep)
    flags = torch.zeros(S, dtype=torch.uint8, device=dev)          # refine_focal_length = refine_extra_params = False
    ids = torch.arange(1, S, device=dev)
    ext = extrinsics.to(torch.float64)
    if use_pnp:
        estopt = AbsolutePoseEstimationOptions()
        estopt.ransac.max_error = 12
        ext, _, _, _, _ = absolute_pose_estimation_batch(ext, params, tracks, points3D, inl, ids, camera_type, flags, estopt,
                                                         AbsolutePoseRefinementOptions(), generator=generator)
    ext, _, _ = pose_refinement_batch(ext, params, tracks, points3D, inl, ids, camera_type, flags,
                                      AbsolutePoseRefinementOptions())
    return ext


def triangulate_window_tracks(pred_track, pred_vis, pred_score, extrinsics, intrinsics, extra_params=None,
                              max_reproj_error=4, min_valid_track_length=3):
    """The geometry of video_runner.py:1189-1262 after `predict_tracks`: undistort, LO-RANSAC triangulation over
    the window, reprojection filter.  Returns (filtered_points, filtered_tracks, filtered_inlier_masks, filtered_vis)."""
    S = extrinsics.shape[0]
    K = intrinsics.expand(S, -1, -1) if intrinsics.shape[0] == 1 else intrinsics
    ep = None if extra_params is None else (extra_params.expand(S, -1) if extra_params.shape[0] == 1 else extra_params)
    tn = cam_from_img(pred_track, K, ep)
    pts, _, _ = triangulate_tracks(extrinsics, tn, track_vis=pred_vis, track_score=pred_score)
    fp, ft, fm, valid = filter_points_and_compute_masks(pts, pred_track, extrinsics, intrinsics, extra_params,
                                                        min_valid_track_length, max_reproj_error)
    return fp, ft, fm, pred_vis[:, valid]


def joint_bundle_adjustment(points3d, extrinsics, intrinsics, tracks, masks, extra_params=None,
                            camera_type="SIMPLE_RADIAL", reproj_error=2.0, tri_angle=1.5, normalize=True, options=None):
    """video_runner.py:494-541 on tensors: [normalize] -> pycolmap.bundle_adjustment (default options, shared camera)
    -> ObservationManager.filter_all_points3D(reproj_error, tri_angle) + filter_observations_with_negative_depth
    -> [normalize].  Returns (points3D (P',3), extrinsics, intrinsics (1,3,3), extra_params (1,1)|None,
    inlier_masks (S,P') bool, keep (P',) bool, summary) -- P' = tracks with >= 2 observations, `keep` marks the
    points that survive the two filters.  The ObservationManager rules restated here [COLMAP 3.10,
    observation_manager.cc]: an observation is dropped when its squared reprojection error exceeds reproj_error^2
    or its depth is not positive; a point is dropped when fewer than 2 observations remain or when no pair of its
    remaining views subtends at least tri_angle degrees."""
    S = extrinsics.shape[0]
    ext, pts = extrinsics.to(torch.float64), points3d.to(torch.float64)
    if normalize:
        ext, pts = _ba.normalize_reconstruction(ext, pts)
    K = intrinsics.expand(S, -1, -1) if intrinsics.shape[0] == 1 else intrinsics
    ep = None if extra_params is None else (extra_params.expand(S, -1) if extra_params.shape[0] == 1 else extra_params)
    p_opt, e_opt, K_opt, x_opt, summ = _ba.bundle_adjustment(pts, ext, K, tracks, masks, None, ep, True, camera_type,
                                                             options or BundleAdjustmentOptions())
    vi, deleted = summ["valid_idx"], summ["deleted"]
    tr, mk = tracks[:, vi], masks[:, vi].bool()
    # per-observation reprojection / depth test (`detail`), then the triangulation-angle test over the survivors
    _, detail = filter_all_points3D(p_opt, tr, e_opt, K_opt, x_opt, max_reproj_error=reproj_error, check_triangle=False,
                                    return_detail=True, hard_max=-1)
    inl = mk & detail
    keep, _ = filter_all_points3D(p_opt, torch.where(inl[..., None], tr, torch.full_like(tr, 1e9)), e_opt, K_opt, x_opt,
                                  max_reproj_error=reproj_error, min_tri_angle=tri_angle, check_triangle=True,
                                  hard_max=-1)
    keep = keep & (inl.sum(0) >= 2) & ~deleted
    inl = inl & keep[None]
    if normalize:
        e_opt, p_opt = _ba.normalize_reconstruction(e_opt, p_opt, keep)
    return p_opt, e_opt, K_opt[0:1], (None if x_opt is None else x_opt[0:1]), inl, keep, summ
