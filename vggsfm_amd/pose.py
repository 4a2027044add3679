"""Batched absolute-pose refinement on the device (host side of ``vgg_pose_refine``).

Replaces the per-frame ``pycolmap.pose_refinement`` calls of the reference
(vggsfm/utils/triangulation.py:387,590; vggsfm/runners/video_runner.py:1001): all frames are refined
concurrently, one workgroup each, straight from the dense (S,P) track tensor.
"""
import ctypes

import torch

from . import _lib
from .ba import MODEL_ID, quat_to_rotmat, rotmat_to_quat
from .ba_options import LOSS_ID, AbsolutePoseEstimationOptions, AbsolutePoseRefinementOptions


def _options(refopts: AbsolutePoseRefinementOptions):
    # RefineAbsolutePose: Ceres defaults + COLMAP's gradient_tolerance / max_num_iterations
    return _lib.BAOptions(refopts.max_num_iterations, 10, 1, 1e-6, refopts.gradient_tolerance, 1e-8,
                          1e4, 1e16, 1e-32, 1e-6, 1e32, 1e-3, 0)


def pose_refinement_batch(extrinsics, intr_params, points2D, points3D, inlier_mask, frame_ids, camera_type,
                          refine_flags, refopts=None):
    """Refine the frames listed in `frame_ids`.

    extrinsics (S,3,4) f64, intr_params (S,4) f64 = (f,cx,cy,k) per frame, points2D (S,P,2), points3D (P,3),
    inlier_mask (S,P) bool, frame_ids 1-D long tensor/list, refine_flags (S,) uint8 (bit0 focal, bit1 extra).
    Returns (extrinsics (S,3,4), intr_params (S,4), summaries list of dicts) -- untouched for other frames."""
    _lib.require_gpu(extrinsics, points2D, points3D, inlier_mask)
    L = _lib.lib()
    refopts = refopts or AbsolutePoseRefinementOptions()
    dev = points3D.device
    S, P = inlier_mask.shape
    ext = extrinsics.to(torch.float64)
    cam_q = rotmat_to_quat(ext[:, :, :3]).contiguous()
    cam_t = ext[:, :, 3].contiguous()
    intr = intr_params.to(torch.float64).contiguous().clone()
    fid = torch.as_tensor(frame_ids, dtype=torch.int32, device=dev).contiguous()
    F = int(fid.numel())
    if F == 0:
        return ext.clone(), intr, []
    pts = points3D.to(torch.float64).contiguous()
    if points2D.dtype == torch.float64:
        tr, is64 = points2D.contiguous(), 1
    else:
        tr, is64 = points2D.to(torch.float32).contiguous(), 0
    mk = inlier_mask.to(torch.uint8).contiguous()
    rf = refine_flags.to(device=dev, dtype=torch.uint8).contiguous()
    summ = torch.zeros(F * ctypes.sizeof(_lib.BASummary), dtype=torch.uint8, device=dev)
    co = _options(refopts)
    _lib.check(L.vgg_pose_refine(_lib.ptr(pts), _lib.ptr(tr), is64, _lib.ptr(mk), S, P, _lib.ptr(fid), F,
                                 _lib.ptr(cam_q), _lib.ptr(cam_t), _lib.ptr(intr), MODEL_ID[camera_type], _lib.ptr(rf),
                                 ctypes.byref(co), LOSS_ID["CAUCHY"], ctypes.c_double(refopts.loss_function_scale),
                                 _lib.ptr(summ), _lib.stream_ptr()), "vgg_pose_refine")
    out_ext = torch.cat([quat_to_rotmat(cam_q), cam_t[:, :, None]], -1)
    # frames that were not refined keep their exact input matrices
    keep = torch.ones(S, dtype=torch.bool, device=dev)
    keep[fid.long()] = False
    out_ext[keep] = ext[keep]
    raw = summ.cpu().numpy().tobytes()
    sums = []
    for i in range(F):
        s = _lib.BASummary.from_buffer_copy(raw[i * ctypes.sizeof(_lib.BASummary):(i + 1) * ctypes.sizeof(_lib.BASummary)])
        sums.append(dict(frame=int(fid[i]), initial_cost=s.initial_cost, final_cost=s.final_cost,
                         num_iterations=s.num_iterations, termination=s.termination))
    return out_ext, intr, sums


def focal_length_factors(estoptions):
    """COLMAP EstimateAbsolutePose: quadratically spaced focal length factors (1 factor = 1.0 when not estimating)."""
    if not estoptions.estimate_focal_length:
        return [1.0]
    n = estoptions.num_focal_length_samples
    lo, hi = estoptions.min_focal_length_ratio, estoptions.max_focal_length_ratio
    return [lo + (hi - lo) * (i * (1.0 / n)) * (i * (1.0 / n)) for i in range(n)]


def draw_minimal_samples(candidate_mask, num_hypotheses, generator=None):
    """(F,P) bool -> (F,H,3) int32: three DISTINCT candidate indices per hypothesis, uniform (host-seeded torch RNG on the
    device).  Frames with fewer than 3 candidates get zeros (the caller marks them failed)."""
    F = candidate_mask.shape[0]
    r = torch.rand((F, num_hypotheses, 3), dtype=torch.float64, device=candidate_mask.device, generator=generator)
    return samples_from_uniforms(candidate_mask, r)


def samples_from_uniforms(candidate_mask, r):
    """The sampling arithmetic of `draw_minimal_samples` on given uniform numbers r (F,H,3) in [0,1): position among the
    frame's candidates (in index order) -> point index.  (Separate so that a test can replay recorded numbers.)"""
    F, P = candidate_mask.shape
    num_hypotheses = r.shape[1]
    cnt = candidate_mask.sum(1)
    table = torch.argsort((~candidate_mask).to(torch.uint8), dim=1, stable=True)        # candidates first, in index order
    n = cnt.clamp(min=3).to(torch.float64)[:, None]
    nl = n.long()
    r0 = torch.minimum((r[..., 0] * n).long(), nl - 1)
    r1 = torch.minimum((r[..., 1] * (n - 1)).long(), nl - 2)
    r1 = r1 + (r1 >= r0).long()
    r2 = torch.minimum((r[..., 2] * (n - 2)).long(), nl - 3)
    lo, hi = torch.minimum(r0, r1), torch.maximum(r0, r1)
    r2 = r2 + (r2 >= lo).long()
    r2 = r2 + (r2 >= hi).long()
    pos = torch.stack([r0, r1, r2], -1).clamp(max=P - 1)
    smp = table.gather(1, pos.reshape(F, -1)).reshape(F, num_hypotheses, 3)
    smp[cnt < 3] = 0
    return smp.to(torch.int32).contiguous()


def p3p_ransac(points2D_normalized, points3D, candidate_mask, samples, max_error_sq, frames_per_sample_set=1):
    """Host side of ``vgg_p3p_ransac``.  points2D_normalized (F,P,2) f64, points3D (P,3) f64, candidate_mask (F,P) bool or
    None, samples (F / frames_per_sample_set, H, 3) int32, max_error_sq (F,) f64.
    Returns (pose (F,3,4), num_inliers (F,) int32, residual_sum (F,), best (F,) int32, inlier_mask (F,P) bool)."""
    _lib.require_gpu(points2D_normalized, points3D, samples, max_error_sq)
    L = _lib.lib()
    dev = points3D.device
    F, P = points2D_normalized.shape[0], points2D_normalized.shape[1]
    H = samples.shape[1]
    x = points2D_normalized.to(torch.float64).contiguous()
    X = points3D.to(torch.float64).contiguous()
    mk = None if candidate_mask is None else candidate_mask.to(torch.uint8).contiguous()
    smp = samples.to(torch.int32).contiguous()
    thr = max_error_sq.to(torch.float64).contiguous()
    pose = torch.zeros((F, 3, 4), dtype=torch.float64, device=dev)
    num = torch.zeros(F, dtype=torch.int32, device=dev)
    rsum = torch.zeros(F, dtype=torch.float64, device=dev)
    best = torch.zeros(F, dtype=torch.int32, device=dev)
    inl = torch.zeros((F, P), dtype=torch.uint8, device=dev)
    if F == 0:
        return pose, num, rsum, best, inl.bool()
    ws = torch.empty(int(L.vgg_p3p_ransac_workspace_bytes(F, H)), dtype=torch.uint8, device=dev)
    _lib.check(L.vgg_p3p_ransac(_lib.ptr(x), _lib.ptr(X), _lib.ptr(mk), _lib.ptr(smp), F, frames_per_sample_set, P, H,
                                _lib.ptr(thr), _lib.ptr(pose), _lib.ptr(num), _lib.ptr(rsum), _lib.ptr(best), _lib.ptr(inl),
                                _lib.ptr(ws), _lib.stream_ptr()), "vgg_p3p_ransac")
    return pose, num, rsum, best, inl.bool()


def absolute_pose_estimation_batch(extrinsics, intr_params, points2D, points3D, candidate_mask, frame_ids, camera_type,
                                   refine_flags, estoptions=None, refopts=None, generator=None, max_virtual_points=1 << 25):
    """``pycolmap.absolute_pose_estimation`` for the frames in `frame_ids`, all at once: P3P RANSAC over the
    candidate matches (optionally over COLMAP's 30 focal length factors), then the non-linear pose refinement on the
    RANSAC inliers (reference call sites: vggsfm/utils/triangulation.py:413-430, video_runner.py:991-998).

    extrinsics (S,3,4), intr_params (S,4) = (f,cx,cy,k), points2D (S,P,2) pixels, points3D (P,3), candidate_mask (S,P)
    bool (the reference passes ``points2D[mask], points3D[mask]``), refine_flags (S,) uint8 for the refinement.
    Returns (extrinsics (S,3,4), intr_params (S,4), success (S,) bool, num_inliers (S,) int32, inliers (S,P) bool);
    frames that fail (fewer than 3 candidates / no hypothesis with >= 3 inliers: pycolmap returns None) and frames not
    listed keep their inputs."""
    from .utils.triangulation_helpers import cam_from_img      # (imports pose-free helpers only)

    _lib.require_gpu(extrinsics, points2D, points3D, candidate_mask)
    estoptions = estoptions or AbsolutePoseEstimationOptions()
    refopts = refopts or AbsolutePoseRefinementOptions()
    dev = points3D.device
    S, P = candidate_mask.shape
    ext = extrinsics.to(torch.float64).clone()
    intr = intr_params.to(torch.float64).clone()
    fid = torch.as_tensor(frame_ids, dtype=torch.long, device=dev)
    success = torch.zeros(S, dtype=torch.bool, device=dev)
    num_inl = torch.zeros(S, dtype=torch.int32, device=dev)
    inliers = torch.zeros((S, P), dtype=torch.bool, device=dev)
    if fid.numel() == 0 or P < 3:
        return ext, intr, success, num_inl, inliers
    factors = torch.tensor(focal_length_factors(estoptions), dtype=torch.float64, device=dev)
    G = int(factors.numel())
    H = int(estoptions.ransac.num_hypotheses)
    has_k = camera_type == "SIMPLE_RADIAL"
    per_call = max(1, int(max_virtual_points // max(1, G * P)))           # frames per launch (bounds the (F G, P, 2) array)
    for c0 in range(0, int(fid.numel()), per_call):
        f = fid[c0:c0 + per_call]
        Fc = int(f.numel())
        cand = candidate_mask[f].bool()
        samples = draw_minimal_samples(cand, H, generator)
        # virtual frames (frame-major, factor-minor): points normalised with the scaled camera, as COLMAP does
        foc = (intr[f, 0][:, None] * factors[None]).reshape(-1)          # (Fc G,)
        K = torch.zeros((Fc * G, 3, 3), dtype=torch.float64, device=dev)
        K[:, 0, 0] = foc
        K[:, 1, 1] = foc
        K[:, 2, 2] = 1.0
        K[:, 0, 2] = intr[f, 1].repeat_interleave(G)
        K[:, 1, 2] = intr[f, 2].repeat_interleave(G)
        extra = intr[f, 3].repeat_interleave(G)[:, None].contiguous() if has_k else None
        pts2 = points2D[f].repeat_interleave(G, dim=0)
        xn = cam_from_img(pts2, K, extra)
        thr = (float(estoptions.ransac.max_error) / foc) ** 2             # CamFromImgThreshold: max_error / focal
        pose, num, rsum, _, inl = p3p_ransac(xn, points3D, cand.repeat_interleave(G, dim=0), samples, thr, G)
        # best factor per frame: most inliers, then the smaller residual sum, then the first factor
        num2, rs2 = num.reshape(Fc, G).long(), rsum.reshape(Fc, G)
        top = num2.max(1).values
        key = torch.where(num2 == top[:, None], rs2, torch.full_like(rs2, float("inf")))
        j = key.argmin(1)                                                 # first minimum
        sel = torch.arange(Fc, device=dev) * G + j
        ok = (top >= 3) & (cand.sum(1) >= 3)
        fo = f[ok]
        ext[fo] = pose[sel][ok]
        intr[fo, 0] = foc[sel][ok]
        success[fo] = True
        num_inl[fo] = top[ok].to(torch.int32)
        inliers[fo] = inl[sel][ok]
    done = torch.nonzero(success).squeeze(1)
    if done.numel():
        # RefineAbsolutePose on the RANSAC inliers (pycolmap.absolute_pose_estimation does this before returning)
        ext, intr, _ = pose_refinement_batch(ext, intr, points2D, points3D, inliers, done, camera_type, refine_flags, refopts)
    return ext, intr, success, num_inl, inliers
