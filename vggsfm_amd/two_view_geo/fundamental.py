"""``estimate_fundamental`` on the device (vggsfm/two_view_geo/fundamental.py:43-183): 7-point RANSAC, two rounds of
8-point local optimisation, winner by (inlier count, mean inlier residual) -- every image pair of the batch at once.
Host side of ``vgg_fmat_seven_point`` / ``vgg_fmat_score`` / ``vgg_fmat_eight_point`` / ``vgg_fmat_residuals``; the
only host-side step between the launches is the stable sort that selects the lo_num best hypotheses."""
import ctypes

import torch

from .. import _lib
from .utils import generate_samples

BIG = 1e6


def _score(L, p1, p2, vm, F, fvalid, thr):
    B, N = p1.shape[0], p1.shape[1]
    K = F.shape[1]
    cnt = torch.empty((B, K), dtype=torch.int32, device=p1.device)
    rs = torch.empty((B, K), dtype=torch.float64, device=p1.device)
    _lib.check(L.vgg_fmat_score(_lib.ptr(p1), _lib.ptr(p2), _lib.ptr(vm), _lib.ptr(F), _lib.ptr(fvalid), B, N, K,
                                ctypes.c_double(thr), _lib.ptr(cnt), _lib.ptr(rs), _lib.stream_ptr()), "vgg_fmat_score")
    return cnt, rs


def _eight_point(L, p1, p2, vm, Fsrc, cnt_src, lo, thr):
    B, N = p1.shape[0], p1.shape[1]
    K = Fsrc.shape[1]
    lo = min(int(lo), K)
    order = torch.sort(cnt_src, dim=1, descending=True, stable=True).indices[:, :lo].to(torch.int32).contiguous()
    F = torch.empty((B, lo, 9), dtype=torch.float64, device=p1.device)
    ok = torch.empty((B, lo), dtype=torch.uint8, device=p1.device)
    _lib.check(L.vgg_fmat_eight_point(_lib.ptr(p1), _lib.ptr(p2), _lib.ptr(vm), _lib.ptr(Fsrc), _lib.ptr(cnt_src), _lib.ptr(order),
                                      B, N, K, lo, ctypes.c_double(thr), _lib.ptr(F), _lib.ptr(ok), _lib.stream_ptr()),
               "vgg_fmat_eight_point")
    return F, ok


def estimate_fundamental(points1, points2, max_ransac_iters=4096, max_error=1, lo_num=300, valid_mask=None, squared=True,
                         second_refine=True, loopresidual=False, return_residuals=False, samples=None):
    """points1, points2 (B,N,2) pixels; valid_mask (B,N) bool.  Returns (best_fmat (B,3,3) scaled to F[2,2] = 1 when
    |F[2,2]| > 1e-8 (normalize_transformation), best_inlier_num (B,), best_inlier_mask (B,N) [, best_residuals (B,N)]).
    `samples` (H,7) int overrides the host draw (tests); `loopresidual` is accepted and ignored (the kernels never
    materialise the (B,K,N) residual tensor it was there to avoid)."""
    if not squared:
        raise NotImplementedError("only squared Sampson thresholds (the reference's default) are implemented")
    _lib.require_gpu(points1, points2, valid_mask)
    L = _lib.lib()
    dev = points1.device
    B, N, _ = points1.shape
    if N < 8:
        raise ValueError(f"need at least 8 matches, got {N}")
    p1 = points1.to(torch.float64).contiguous()
    p2 = points2.to(torch.float64).contiguous()
    vm = None if valid_mask is None else valid_mask.to(torch.uint8).contiguous()
    thr = float(max_error) ** 2
    if samples is None:
        samples = generate_samples(N, max_ransac_iters, 7)
    smp = torch.as_tensor(samples, dtype=torch.int32, device=dev).contiguous()
    H = int(smp.shape[0])
    F7 = torch.empty((B, H, 3, 9), dtype=torch.float64, device=dev)
    v7 = torch.empty((B, H, 3), dtype=torch.uint8, device=dev)
    _lib.check(L.vgg_fmat_seven_point(_lib.ptr(p1), _lib.ptr(p2), _lib.ptr(smp), B, N, H, _lib.ptr(F7), _lib.ptr(v7),
                                      _lib.stream_ptr()), "vgg_fmat_seven_point")
    Fa, va = F7.reshape(B, 3 * H, 9), v7.reshape(B, 3 * H)
    cnt, rs = _score(L, p1, p2, vm, Fa, va, thr)
    F8, v8 = _eight_point(L, p1, p2, vm, Fa, cnt, lo_num, thr)
    c8, r8 = _score(L, p1, p2, vm, F8, v8, thr)
    allF, allc, allr = [Fa, F8], [cnt, c8], [rs, r8]
    if second_refine:
        # the reference ranks and refits the second round on residuals_lo BEFORE its valid mask is applied
        # (fundamental.py:126-152): invalid matches take part there, and only there
        if vm is not None and not bool(vm.all()):
            ones = torch.ones_like(vm)
            c8u, _ = _score(L, p1, p2, ones, F8, v8, thr)
            F9, v9 = _eight_point(L, p1, p2, ones, F8, c8u, lo_num // 2, thr)
        else:
            F9, v9 = _eight_point(L, p1, p2, vm, F8, c8, lo_num // 2, thr)
        c9, r9 = _score(L, p1, p2, vm, F9, v9, thr)
        allF.append(F9); allc.append(c9); allr.append(r9)                       # noqa: E702
    Fall, call, rall = torch.cat(allF, 1), torch.cat(allc, 1).long(), torch.cat(allr, 1)
    # most inliers, then the smallest mean inlier residual, then the lowest index (two_view_geo/utils.py:63-87)
    mean = torch.where(call > 0, rall / call.clamp(min=1).double(), torch.full_like(rall, BIG))
    top = call.max(dim=1, keepdim=True).values
    best = torch.where(call == top, mean, torch.full_like(mean, float("inf"))).argmin(dim=1)
    ar = torch.arange(B, device=dev)
    Fb = Fall[ar, best].contiguous()
    res = torch.empty((B, N), dtype=torch.float64, device=dev)
    _lib.check(L.vgg_fmat_residuals(_lib.ptr(p1), _lib.ptr(p2), _lib.ptr(vm), _lib.ptr(Fb), B, N, _lib.ptr(res),
                                    _lib.stream_ptr()), "vgg_fmat_residuals")
    found = call[ar, best] >= 0
    mask = (res <= thr) & found[:, None]
    num = torch.where(found, call[ar, best], torch.zeros_like(top[:, 0]))
    F33 = Fb[:, 8:9]
    Fn = torch.where(F33.abs() > 1e-8, Fb / torch.where(F33.abs() > 1e-8, F33, torch.ones_like(F33)), Fb).reshape(B, 3, 3)
    if return_residuals:
        return Fn, num, mask, res
    return Fn, num, mask
