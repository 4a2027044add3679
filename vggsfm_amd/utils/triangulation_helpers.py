"""Host-side mirror of ``vggsfm/utils/triangulation_helpers.py`` (reference), backed by HIP kernels.

Same function names, argument meaning, return values and error behaviour as the reference, so
``vggsfm.runners`` code can import these instead.  Tensors must live on the MI355X; results are
fresh tensors on the same device.  Chunking arguments (``max_points_num``) are accepted for
signature compatibility and ignored: the kernels stream the whole problem without the reference's
(S*S,P) temporaries, and the result does not depend on chunking.
"""
import ctypes

import numpy as np
import torch

from .. import _lib
from ..ba_options import BundleAdjustmentOptions


def _f64c(t):
    return t.to(torch.float64).contiguous()


def _tracks_arg(tracks):
    if tracks.dtype == torch.float64:
        return tracks.contiguous(), 1
    return tracks.to(torch.float32).contiguous(), 0


def _extra(extra_params):
    if extra_params is None:
        return None, 0
    if extra_params.dim() != 2:
        raise ValueError("extra_params must be BxN")
    k = extra_params.shape[1]
    if k not in (1, 2, 4):
        raise ValueError("Unsupported number of distortion parameters")
    return _f64c(extra_params), k


def project_3D_points(points3D, extrinsics, intrinsics=None, extra_params=None, return_points_cam=False,
                      default=0, only_points_cam=False):
    """Reference: triangulation_helpers.py:311-355.  points3D (P,3), extrinsics (S,3,4), intrinsics (S,3,3)
    -> (S,P,2) [, (S,3,P)]."""
    if default != 0:
        raise NotImplementedError("only default=0 is used by the reference")
    _lib.require_gpu(points3D, extrinsics)
    L = _lib.lib()
    pts = _f64c(points3D)
    ext = _f64c(extrinsics)
    S, P = ext.shape[0], pts.shape[0]
    dev = pts.device
    want_cam = return_points_cam or only_points_cam
    out_cam = torch.empty((S, 3, P), dtype=torch.float64, device=dev) if want_cam else None
    if only_points_cam:
        _lib.check(L.vgg_project_points(_lib.ptr(pts), P, _lib.ptr(ext), None, None, 0, S, None, _lib.ptr(out_cam),
                                        _lib.stream_ptr()), "vgg_project_points")
        return out_cam
    K = _f64c(intrinsics)
    ep, k = _extra(extra_params)
    out_uv = torch.empty((S, P, 2), dtype=torch.float64, device=dev)
    _lib.check(L.vgg_project_points(_lib.ptr(pts), P, _lib.ptr(ext), _lib.ptr(K), _lib.ptr(ep), k, S,
                                    _lib.ptr(out_uv), _lib.ptr(out_cam), _lib.stream_ptr()), "vgg_project_points")
    if return_points_cam:
        return out_uv, out_cam
    return out_uv


def filter_all_points3D(points3D, points2D, extrinsics, intrinsics, extra_params=None, max_reproj_error=4,
                        min_tri_angle=1.5, check_triangle=True, return_detail=False, hard_max=300,
                        max_points_num=819200, behind_value=1e6):
    """Reference: triangulation_helpers.py:133-307.  Returns (mask (P) bool, detail (S,P) bool | None)."""
    _lib.require_gpu(points3D, points2D, extrinsics, intrinsics)
    L = _lib.lib()
    pts = _f64c(points3D)
    ext = _f64c(extrinsics)
    K = _f64c(intrinsics)
    ep, k = _extra(extra_params)
    tr, is64 = _tracks_arg(points2D)
    S, P = ext.shape[0], pts.shape[0]
    dev = pts.device
    mask = torch.empty(P, dtype=torch.uint8, device=dev)
    detail = torch.empty((S, P), dtype=torch.uint8, device=dev) if return_detail else None
    ws = torch.empty(max(int(L.vgg_filter_points_workspace_bytes(S)), 8), dtype=torch.uint8, device=dev)
    _lib.check(L.vgg_filter_points(_lib.ptr(pts), P, _lib.ptr(tr), is64, _lib.ptr(ext), _lib.ptr(K), _lib.ptr(ep), k, S,
                                   ctypes.c_double(max_reproj_error), ctypes.c_double(min_tri_angle),
                                   int(bool(check_triangle)), ctypes.c_double(hard_max), ctypes.c_double(behind_value),
                                   _lib.ptr(mask), _lib.ptr(detail), _lib.ptr(ws), _lib.stream_ptr()),
               "vgg_filter_points")
    return mask.bool(), (detail.bool() if return_detail else None)


def cam_from_img(pred_tracks, intrinsics, extra_params=None):
    """Reference: triangulation_helpers.py:398-428 (+ distortion.py:27-99 when extra_params is given)."""
    _lib.require_gpu(pred_tracks, intrinsics)
    L = _lib.lib()
    tr, is64 = _tracks_arg(pred_tracks)
    K = _f64c(intrinsics)
    ep, k = _extra(extra_params)
    S, P = tr.shape[0], tr.shape[1]
    dev = tr.device
    out = torch.empty((S, P, 2), dtype=torch.float64, device=dev)
    max_it = 100
    ws = None
    if k:
        ws = torch.empty(int(L.vgg_cam_from_img_workspace_bytes(S, P, max_it)), dtype=torch.uint8, device=dev)
    iters = ctypes.c_int(0)
    _lib.check(L.vgg_cam_from_img(_lib.ptr(tr), is64, _lib.ptr(K), _lib.ptr(ep), k, S, P, _lib.ptr(out), max_it,
                                  ctypes.c_double(1e-10), ctypes.c_double(1e-6),
                                  ctypes.c_double(float(torch.finfo(torch.float64).eps)), _lib.ptr(ws),
                                  ctypes.byref(iters), _lib.stream_ptr()), "vgg_cam_from_img")
    return out


def create_intri_matrix(focal_length, principal_point):
    """Reference: triangulation_helpers.py:590-623."""
    shape = focal_length.shape[:-1]
    K = torch.zeros(*shape, 3, 3, dtype=focal_length.dtype, device=focal_length.device)
    K[..., 0, 0] = focal_length[..., 0]
    K[..., 1, 1] = focal_length[..., 1]
    K[..., 2, 2] = 1.0
    K[..., 0, 2] = principal_point[..., 0]
    K[..., 1, 2] = principal_point[..., 1]
    return K


def prepare_ba_options():
    """Reference: triangulation_helpers.py:626-635 (tolerances x10, 50 iterations)."""
    o = BundleAdjustmentOptions()
    o.solver_options.function_tolerance *= 10
    o.solver_options.gradient_tolerance *= 10
    o.solver_options.parameter_tolerance *= 10
    o.solver_options.max_num_iterations = 50
    o.solver_options.max_linear_solver_iterations = 200
    o.print_summary = False
    return o


def generate_combinations(N):
    """Reference: triangulation_helpers.py:638-645 (`itertools.combinations(np.arange(N), 2)` materialised on the
    host: 60 ms at N = 200, and the reference repeats it for every chunk).  Same pairs in the same lexicographic
    order from `np.triu_indices`."""
    i, j = np.triu_indices(N, 1)
    return np.stack([i, j], 1).astype(np.int64).reshape(-1, 2)
