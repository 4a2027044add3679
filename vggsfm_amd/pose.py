"""Batched absolute-pose refinement on the device (host side of ``vgg_pose_refine``).

Replaces the per-frame ``pycolmap.pose_refinement`` calls of the reference
(vggsfm/utils/triangulation.py:387,590; vggsfm/runners/video_runner.py:1001): all frames are refined
concurrently, one workgroup each, straight from the dense (S,P) track tensor.
"""
import ctypes

import torch

from . import _lib
from .ba import MODEL_ID, quat_to_rotmat, rotmat_to_quat
from .ba_options import LOSS_ID, AbsolutePoseRefinementOptions


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
