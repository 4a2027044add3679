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
import random

import torch

from . import _lib
from . import ba as _ba
from .ba import window_bundle_adjustment  # noqa: F401  (re-export: the local BA of a window)
from .ba_options import AbsolutePoseEstimationOptions, AbsolutePoseRefinementOptions, BundleAdjustmentOptions
from .pose import absolute_pose_estimation_batch, pose_refinement_batch
from .utils.triangulation import _intr_params, triangulate_tracks
from .utils.triangulation_helpers import cam_from_img, filter_all_points3D


def align_camera_extrinsics(cameras_src, cameras_tgt, estimate_scale=True, eps=1e-9):
    """Same contract as the reference's utils/align.py:145-205: (B,3,4) source and target cameras (x_cam = R X + t) ->
    (align_R (1,3,3), align_T (1,3), align_s) such that `apply_transformation` carries the source set onto the target.

    Derivation used here.  Rotation: the orthogonal Q maximising sum_i tr(Q^T R_tgt,i^T R_src,i) is the polar factor of
    M = mean_i R_tgt,i^T R_src,i; with M = U S V^T that is Q = V U^T.  Translation / scale: a camera's pose in ITS OWN
    source frame, p_i = R_src,i^T t_src,i (minus its centre), and the target translation seen from the same frame,
    q_i = R_src,i^T t_tgt,i, are related by q_i ~ s p_i + T; s is the one-dimensional least-squares slope over all 3 B
    coordinates of the centred sets, T the offset of the means.  B is a window (<= 33 frames): torch ops, no kernel."""
    rot_s, rot_t = cameras_src[..., :3], cameras_tgt[..., :3]
    m = torch.einsum("bji,bjk->ik", rot_t, rot_s) / rot_s.shape[0]
    u, _, vh = torch.linalg.svd(m)
    q = vh.t() @ u.t()
    p_own = torch.einsum("bji,bj->bi", rot_s, cameras_src[..., 3])
    q_own = torch.einsum("bji,bj->bi", rot_s, cameras_tgt[..., 3])
    p_mean, q_mean = p_own.mean(0, keepdim=True), q_own.mean(0, keepdim=True)
    slope = 1.0
    if estimate_scale and p_own.shape[0] > 1:
        dp, dq = p_own - p_mean, q_own - q_mean
        slope = (dp * dq).mean() / dp.square().mean().clamp(eps)
    return q[None], q_mean - slope * p_mean, slope


def apply_transformation(cameras_src, align_R, align_T, align_s, return_extri=True):
    """utils/align.py:208-252: R_i <- R_i Q, t_i <- R_i T + s t_i for the (Q, T, s) of `align_camera_extrinsics`."""
    rot = cameras_src[..., :3]
    new_rot = rot @ align_R.reshape(3, 3)
    new_t = torch.einsum("bij,j->bi", rot, align_T.reshape(3)) + align_s * cameras_src[..., 3]
    if return_extri:
        return torch.cat([new_rot, new_t[..., None]], dim=-1)
    return new_rot, new_t


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


def observation_filter(points3D, extrinsics, intrinsics, extra_params, tracks, masks, max_reproj_error, min_tri_angle):
    """``ObservationManager.filter_all_points3D(max_reproj_error, min_tri_angle)`` +
    ``filter_observations_with_negative_depth`` on dense tensors [COLMAP 3.10, observation_manager.cc]: an observation
    is dropped when its squared reprojection error exceeds max^2 or its depth is not positive; a point is dropped when
    fewer than 2 observations remain or when no pair of its remaining views subtends at least `min_tri_angle`
    degrees.  points3D (P,3), cameras of the S frames, tracks (S,P,2), masks (S,P) -> (inlier (S,P) bool, keep (P,))."""
    mk = masks.bool()
    # per-observation reprojection / depth test (`detail`), then the triangulation-angle test over the survivors
    _, detail = filter_all_points3D(points3D, tracks, extrinsics, intrinsics, extra_params,
                                    max_reproj_error=max_reproj_error, check_triangle=False, return_detail=True,
                                    hard_max=-1)
    inl = mk & detail
    keep, _ = filter_all_points3D(points3D, torch.where(inl[..., None], tracks, torch.full_like(tracks, 1e9)), extrinsics,
                                  intrinsics, extra_params, max_reproj_error=max_reproj_error,
                                  min_tri_angle=min_tri_angle, check_triangle=True, hard_max=-1)
    keep = keep & (inl.sum(0) >= 2)
    return inl & keep[None], keep


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
    inl, keep = observation_filter(p_opt, e_opt, K_opt, x_opt, tracks[:, vi], masks[:, vi], reproj_error, tri_angle)
    keep = keep & ~deleted
    inl = inl & keep[None]
    if normalize:
        e_opt, p_opt = _ba.normalize_reconstruction(e_opt, p_opt, keep)
    return p_opt, e_opt, K_opt[0:1], (None if x_opt is None else x_opt[0:1]), inl, keep, summ


class VideoGeometry:
    """The geometric half of ``VideoRunner`` (vggsfm/runners/video_runner.py:64-247 ``run``, 640-905 ``move_window``,
    1051-1187 ``prepare_window_data``, 494-541 ``joint_BA``) with the learned parts injected as callables and the
    ``point_dict`` / ``frame_dict`` state held in a device-resident :class:`TrackTable`:

        camera_prior(frame_from, frame_to) -> (frame_to - frame_from, 3, 4) predicted extrinsics (OpenCV), any gauge
            -- the reference's ``average_camera_prediction`` over (last window, next window), video_runner.py:662-667
        track_existing(frame_from, frame_to, query_uv (P,2)) -> (tracks (S,P,2), vis (S,P))
            -- ``predict_tracks`` with the carried-over points as queries in frame_from, video_runner.py:1133-1146
        track_new(frame_from, frame_to) -> (tracks (S,N,2), vis (S,N), score (S,N) | None)
            -- ``predict_tracks`` with fresh query points, video_runner.py:1204-1216

    Everything else -- similarity alignment of the predicted cameras, the window-shrinking rule, pose alignment on
    the carried-over points, the reprojection filters, LO-RANSAC triangulation of the new tracks, the local BA with
    constant carried-over points and constant first pose, the table updates, the joint BA with filters and
    normalisation -- runs here, on the device, with the kernels of the batch path."""

    def __init__(self, intrinsics, extra_params=None, camera_type="SIMPLE_RADIAL", max_query_pts=1024, device="cuda",
                 generator=None, camera_predictor=None, images=None):
        """`camera_predictor` + `images` (1,S,3,H,W): the default ``camera_prior`` = the reference's own recipe,
        ``average_camera_prediction`` over (last window, next window) with the query frames (first, middle, last)
        (video_runner.py:655-667; ``vggsfm_amd.utils.utils.camera_prior_from_predictor``)."""
        from .track_table import TrackTable
        from .utils.utils import camera_prior_from_predictor
        self.camera_prior = None if camera_predictor is None else camera_prior_from_predictor(camera_predictor, images)
        if generator is not None and torch.device(generator.device).type != torch.device(device).type:
            raise ValueError(f"generator lives on {generator.device}, the tables on {device}: pass a generator of the same device")
        self.device = torch.device(device)
        self.intrinsics = intrinsics[0:1].to(self.device).clone()                       # 1x3x3 (video_runner.py:149)
        if extra_params is None and camera_type == "SIMPLE_RADIAL":
            extra_params = torch.zeros(1, 1, device=self.device, dtype=self.intrinsics.dtype)   # :146-147
        self.extra_params = None if extra_params is None else extra_params[0:1].to(self.device).clone()
        self.camera_type = camera_type
        self.max_query_pts = int(max_query_pts)
        self.generator = generator
        self._vis_order = {}
        self.table = TrackTable(self.device)

    # ------------------------------------------------------------------ state
    def add_initial_window(self, pred, start_idx, end_idx):
        """``convert_pred_to_point_frame_dict(init_pred, ...)`` (video_runner.py:152)."""
        self.table.add_window_prediction(pred, start_idx, end_idx)

    def select_existing_points(self, frame_idx, max_ratio=1):
        """video_runner.py:1062-1084: the points visible in `frame_idx`, at most max_query_pts * max_ratio of them, with their
        3D position and their pixel in that frame -- IN THE ORDER OF THE REFERENCE'S LIST ``frame_dict[f]["visible_points"]``,
        which is what its draw acts on: over the cap it takes ``sorted(random.sample(list, cap))`` from PYTHON's global
        ``random`` (:1068-1071; demo.py seeds it with cfg.seed) -- ``random.sample`` picks POSITIONS --, under the cap it keeps
        the list as it stands, and the order of this window's carried-over points is the order in which they are appended to the
        lists of the window's frames.  A frame's list: the points that were new in the window that registered it (``move_window``
        adds them first, :866), ascending, then that window's carried-over points in THEIR order (:893-903); every joint BA
        rebuilds all lists in plain id order (``reconstruction_to_dicts``).  `_vis_order[frame]` holds that order for the
        frames registered since the last joint BA (absent: id order).  The same global ``random`` state then gives the same
        subset (tests/golden/video_pinhole_t160.npz: the reference's default 1024-point cap is hit in every window).  Not
        reproduced: after a step-back the reference's list of a RE-registered frame holds its carried-over points twice
        (:454) and a draw from that frame could return a point twice; the table keeps a point once.  With a ``generator``
        (constructor) the subset comes from ``torch.randperm`` on it instead, in id order."""
        t = self.table
        sel = torch.nonzero(t.obs_frame == frame_idx).squeeze(1)                 # (observations are sorted by point id)
        cap = self.max_query_pts * max_ratio
        n = int(sel.numel())
        order = self._vis_order.get(int(frame_idx)) if self.generator is None else None
        if order is not None:
            ids_sorted = t.obs_point[sel]
            where = torch.searchsorted(ids_sorted, order)
            if order.numel() == n and bool((ids_sorted[where.clamp(max=n - 1)] == order).all()):
                sel = sel[where]                                                  # the list's order
            # (else: the list and the table disagree -- a re-registered frame; id order)
        if n > cap:
            if self.generator is None:
                pick = torch.as_tensor(random.sample(range(n), cap), dtype=torch.long, device=sel.device)
                sel = sel[pick]
                sel = sel[torch.argsort(t.obs_point[sel])]                        # sorted(...) of the drawn ids
            else:
                sel = sel[torch.sort(torch.randperm(n, device=self.device, generator=self.generator)[:cap]).values]
        ids = t.obs_point[sel]
        return ids, t.xyz[ids], t.obs_uv[sel]

    # ------------------------------------------------------------------ one window
    def move_window(self, start_idx, end_idx, window_size, camera_prior, track_existing, track_new,
                    min_valid_track_length=3, track_vis_thres=0.05, use_pnp=False):
        """video_runner.py:640-905.  (start_idx, end_idx) is the LAST window.  Returns (start_idx, end_idx, success)."""
        last_window_size = end_idx - start_idx
        assert last_window_size > 0, "last_window_size should be positive"
        t = self.table
        last_start_idx, start_idx = start_idx, end_idx
        end_idx = start_idx + window_size
        print(f"Processing window from {start_idx} to {end_idx}")
        camera_prior = camera_prior or self.camera_prior
        if camera_prior is None:
            raise ValueError("no camera_prior: pass one, or construct VideoGeometry with camera_predictor and images")
        # predicted cameras of (last window, next window), aligned to the last window's reconstruction
        pred_extri = camera_prior(last_start_idx, end_idx).to(self.device, torch.float64)
        last_extri = t.extri[last_start_idx:start_idx].to(torch.float64)
        rel_r, rel_t, rel_s = align_camera_extrinsics(pred_extri[:last_window_size], last_extri)
        aligned_next = apply_transformation(pred_extri[last_window_size:], rel_r, rel_t, rel_s)
        # prepare_window_data: carried-over points tracked through (start_idx - 1 .. end_idx)
        ids, xyz, uv = self.select_existing_points(start_idx - 1)
        tracks_e, vis_e = track_existing(start_idx - 1, end_idx, uv)
        tracks_e, vis_e = tracks_e.to(self.device), vis_e.to(self.device)
        inl_e = vis_e > track_vis_thres
        extri_plus_one = torch.cat([t.extri[start_idx - 1:start_idx].to(torch.float64), aligned_next], dim=0)
        # frames that see fewer than 50 carried-over points end the window early (video_runner.py:709-749)
        few = inl_e.sum(dim=1) < 50
        if bool(few.any()):
            first_invalid = int(torch.nonzero(few)[0, 0])
            if first_invalid > 2:
                window_size = first_invalid - 1
                print(f"Shrink the window from {start_idx}-{end_idx} to {start_idx}-{start_idx + window_size}")
                end_idx = start_idx + window_size
                tracks_e, vis_e, inl_e = tracks_e[:window_size + 1], vis_e[:window_size + 1], inl_e[:window_size + 1]
                extri_plus_one = extri_plus_one[:window_size + 1]
            else:
                print("No valid frame, step back")
                return last_start_idx - 1, start_idx - 1, False
        K, ep = self.intrinsics, self.extra_params
        align_ext = align_next_window(extri_plus_one, tracks_e, inl_e, xyz, K, ep, self.camera_type, use_pnp=use_pnp,
                                      generator=self.generator)
        pts_e, tr_e, m_e, keep_e = filter_points_and_compute_masks(xyz, tracks_e, align_ext, K, ep, min_valid_track_length)
        ids_e = ids[keep_e]
        # new tracks over the (possibly shrunk) window, triangulated with the aligned cameras
        tracks_n, vis_n, score_n = track_new(start_idx - 1, end_idx)
        tracks_n, vis_n = tracks_n.to(self.device), vis_n.to(self.device)
        score_n = None if score_n is None else score_n.to(self.device)
        pts_n, tr_n, m_n, vis_nf = triangulate_window_tracks(tracks_n, vis_n, score_n, align_ext, K, ep,
                                                            min_valid_track_length=min_valid_track_length)
        ne = int(pts_e.shape[0])
        pts_all = torch.cat([pts_e.to(torch.float64), pts_n.to(torch.float64)], dim=0)
        tr_all = torch.cat([tr_e, tr_n.to(tr_e.dtype)], dim=1)
        m_all = torch.cat([m_e, m_n], dim=1)
        # local BA: first pose and carried-over points constant, intrinsics fixed (video_runner.py:800-838)
        S1 = align_ext.shape[0]
        Kw = K.expand(S1, -1, -1)
        epw = None if ep is None else ep.expand(S1, -1)
        p_opt, ext_opt, _, _, summ = window_bundle_adjustment(pts_all, align_ext, Kw, tr_all, m_all, ne, epw, True,
                                                              self.camera_type)
        if summ["termination"] == 5:                                 # FAILURE (log_ba_summary -> RuntimeError)
            raise RuntimeError("Bundle adjustment failed")
        pts_opt = torch.zeros_like(pts_all)                          # pycolmap_to_batch_matrix: one row per track
        pts_opt[summ["valid_idx"]] = p_opt
        # the new points that survive the optimised cameras enter the table
        npts, ntr, nm, nkeep = filter_points_and_compute_masks(pts_opt[ne:], tr_all[:, ne:], ext_opt, K, ep,
                                                               min_valid_track_length)
        pred = {"extrinsics_opencv": ext_opt[1:], "pred_track": ntr[1:], "pred_vis": vis_nf[:, nkeep][1:],
                "valid_2D_mask": nm[1:], "valid_tracks": torch.ones(int(nkeep.sum()), dtype=torch.bool, device=self.device),
                "points3D": npts, "points3D_rgb": None}
        first_new = int(t.num_points)
        t.add_window_prediction(pred, start_idx, end_idx)
        # the carried-over points keep xyz / id and gain the observations of this window (video_runner.py:868-903)
        _, etr, em, ekeep = filter_points_and_compute_masks(pts_opt[:ne], tr_all[:, :ne], ext_opt, K, ep, min_valid_track_length)
        evis = vis_e[:, keep_e][:, ekeep]
        eids = ids_e[ekeep]
        if eids.numel():
            mapping = torch.zeros(t.num_points, dtype=torch.long, device=self.device)
            mapping[eids] = torch.arange(eids.numel(), device=self.device)
            t.update_points(start_idx, end_idx, em[1:], etr[1:], evis[1:], eids, mapping)
        # the reference's per-frame lists of this window's frames (select_existing_points): new points, then carried-over ones
        new_ids = first_new + torch.arange(int(nkeep.sum()), device=self.device)
        for fi, f in enumerate(range(start_idx, end_idx)):
            self._vis_order[f] = torch.cat([new_ids[nm[1 + fi].bool()], eids[em[1 + fi].bool()]])
        return start_idx, end_idx, True

    # ------------------------------------------------------------------ joint BA
    def joint_BA(self, start_idx, end_idx, reproj_error=2.0, tri_angle=1.5, normalize=True):
        """video_runner.py:494-541: BA over every frame so far with the shared camera refined, the ObservationManager
        filters, normalisation; the table is rebuilt from the result (``reconstruction_to_dicts``)."""
        t = self.table
        xyz, ext, tracks, masks, _ = t.window_tensors(start_idx, end_idx)
        pts, e, K, x, inl, keep, summ = joint_bundle_adjustment(xyz, ext, self.intrinsics, tracks, masks, self.extra_params,
                                                                self.camera_type, reproj_error, tri_angle, normalize)
        self.intrinsics = K.to(torch.float32)[0:1].clone()           # (.float() in the reference)
        if self.camera_type == "SIMPLE_RADIAL" and x is not None:
            self.extra_params = x.to(torch.float32)[0:1].clone()
        vi = summ["valid_idx"]
        rgb = t.rgb[vi] if t.rgb.shape[0] == xyz.shape[0] else None
        t.reset_from_tensors(pts, e, tracks[:, vi], inl, keep, rgb, start_idx)
        self._vis_order.clear()                                      # (every list rebuilt in id order)
        return summ

    # ------------------------------------------------------------------ the loop of VideoRunner.run
    def run(self, num_frames, init_end_idx, window_size, camera_prior, track_existing, track_new, joint_BA_interval=6,
            use_pnp=False):
        """video_runner.py:156-189 after the initial window has been added: slide until the last frame, joint BA every
        `joint_BA_interval` windows and once more at the end."""
        start_idx, end_idx, T = 0, init_end_idx, num_frames
        window_counter = 0
        while end_idx < T:
            ws = T - end_idx if (T - end_idx) <= int(1.25 * window_size) else window_size
            start_idx, end_idx, ok = self.move_window(start_idx, end_idx, ws, camera_prior, track_existing, track_new,
                                                      use_pnp=use_pnp)
            if not ok:
                print("Moving window failed, trying again. (This should not happen in most cases)")
                self.max_query_pts *= 2
                start_idx, end_idx, ok = self.move_window(start_idx, end_idx, window_size, camera_prior, track_existing,
                                                          track_new, use_pnp=use_pnp)
                self.max_query_pts //= 2
                if not ok:
                    raise RuntimeError("moving the window failed twice")
            if window_counter % joint_BA_interval == 0:
                print("Running joint BA:")
                self.joint_BA(0, end_idx, normalize=True)
            window_counter += 1
        print("Running joint BA for the entire sequence:")
        self.joint_BA(0, T, normalize=True)
        return self.table
