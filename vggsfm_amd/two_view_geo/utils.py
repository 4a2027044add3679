"""Helpers of the two-view stage under the reference's names (vggsfm/two_view_geo/utils.py): the sample generator, the
Sampson distance and the inlier test on the device (``vgg_fmat_residuals``), the hypothesis score, and the re-exports
of the helpers `estimate_preliminary_cameras` uses."""
import numpy as np
import torch

from .. import _lib


def generate_samples(N, target_num, sample_num, expand_ratio=2):
    """vggsfm/two_view_geo/utils.py:39-60: `target_num` index tuples of `sample_num` distinct match indices in [0, N),
    drawn like the reference does (numpy's global RNG: target_num * expand_ratio tuples, the ones with a repeated
    index dropped, the first target_num kept -- fewer when too many are dropped)."""
    draw = np.random.randint(0, N, size=(target_num * expand_ratio, sample_num))
    srt = np.sort(draw, axis=1)
    distinct = (srt[:, 1:] != srt[:, :-1]).all(axis=1)
    return draw[distinct][:target_num]


def calculate_residual_indicator(residuals, max_residual, debug=False, check=False, nanvalue=1e6):
    """utils.py:63-87 (`debug` and `check` are accepted and unused, as in the reference -- its call sites pass them:
    utils/triangulation.py:937-941, two_view_geo/fundamental.py:168): score of every hypothesis = its inlier count + a term in [0, 1) that prefers the smaller mean inlier
    residual (normalised by the largest mean of the whole tensor, so it never changes the order by count).
    residuals (B,S,N) -> (indicator (B,S) f64, inlier_num (B,S) i64, inlier_mask (B,S,N) bool).  Plain tensor ops: the
    RANSAC kernels compute the same ranking on the fly (vgg_fmat_score + the selection in `estimate_fundamental`)."""
    inlier_mask = residuals <= max_residual
    inlier_num = inlier_mask.sum(dim=-1)
    ind = (inlier_mask.float() * residuals).sum(dim=-1) / inlier_num
    ind = torch.nan_to_num(ind, nan=nanvalue, posinf=nanvalue, neginf=nanvalue)
    thres = ind.max() + 1e-6
    ind = (thres - ind) / thres
    return ind.double() + inlier_num.double(), inlier_num, inlier_mask


def sampson_epipolar_distance_batched(pts1, pts2, Fm, squared=True, eps=1e-8, debug=False, evaluation=False):
    """utils.py:90-173: Sampson distance of the matches pts1 (B,N,(2|3)) <-> pts2 (B,N,(2|3)) under the fundamental
    matrices Fm (B,K,3,3) -> (B,K,N); squared by default.  One launch of `vgg_fmat_residuals` over the B K (pair,
    hypothesis) combinations, in float64.  `evaluation` only empties the allocator cache in the reference (:147-153): accepted,
    nothing to do here.  `debug=True` returns the reference's five intermediates (numerator, denominator, distance, epipolar
    lines of both sides; :167-168) -- a diagnostic path, plain tensor operations on the device."""
    if not isinstance(Fm, torch.Tensor):
        raise TypeError(f"Fm type is not a torch.Tensor. Got {type(Fm)}")
    if Fm.shape[-2:] != (3, 3):
        raise ValueError(f"Fm must be a (B, K, 3, 3) tensor. Got {Fm.shape}")
    _lib.require_gpu(pts1, pts2, Fm)
    if debug:
        h = lambda p: p if p.shape[-1] == 3 else torch.cat((p, torch.ones_like(p[..., :1])), dim=-1)
        K = Fm.shape[1]
        F64 = Fm.to(torch.float64)
        q1 = h(pts1).to(torch.float64)[:, None].expand(-1, K, -1, -1)
        q2 = h(pts2).to(torch.float64)[:, None].expand(-1, K, -1, -1)
        line1_in_2 = torch.einsum("bkij,bkjn->bkin", q1, F64.transpose(-2, -1))
        line2_in_1 = torch.einsum("bkij,bkjn->bkin", q2, F64)
        numerator = (q2 * line1_in_2).sum(dim=-1).pow(2)
        denominator = line1_in_2[..., :2].norm(2, dim=-1).pow(2) + line2_in_1[..., :2].norm(2, dim=-1).pow(2)
        return numerator, denominator, (numerator / denominator).to(pts1.dtype), line1_in_2, line2_in_1
    B, N = pts1.shape[0], pts1.shape[1]
    K = Fm.shape[1]
    p1 = pts1[..., :2].to(torch.float64)[:, None].expand(-1, K, -1, -1).reshape(B * K, N, 2).contiguous()
    p2 = pts2[..., :2].to(torch.float64)[:, None].expand(-1, K, -1, -1).reshape(B * K, N, 2).contiguous()
    F = Fm.to(torch.float64).reshape(B * K, 9).contiguous()
    vm = torch.ones((B * K, N), dtype=torch.uint8, device=pts1.device)
    res = torch.empty((B * K, N), dtype=torch.float64, device=pts1.device)
    _lib.check(_lib.lib().vgg_fmat_residuals(_lib.ptr(p1), _lib.ptr(p2), _lib.ptr(vm), _lib.ptr(F), B * K, N, _lib.ptr(res),
                                             _lib.stream_ptr()), "vgg_fmat_residuals")
    res = res.reshape(B, K, N)
    return res if squared else (res + eps).sqrt()


def inlier_by_fundamental(fmat, tracks, max_error=0.5):
    """utils.py:300-322: fmat (B,S-1,3,3) of the pairs (frame 0, frame s), tracks (B,S,N,2) -> inlier mask (B,S-1,N):
    squared Sampson distance <= max_error^2."""
    B, S, N, _ = tracks.shape
    left = tracks[:, 0:1].expand(-1, S - 1, -1, -1).reshape(B * (S - 1), N, 2)
    right = tracks[:, 1:].reshape(B * (S - 1), N, 2)
    res = sampson_epipolar_distance_batched(left, right, fmat.reshape(B * (S - 1), 1, 3, 3), squared=True)[:, 0]
    return (res <= max_error ** 2).reshape(B, S - 1, N)


def get_default_intri(width, height, device, dtype, ratio=1.0):
    from .estimate_preliminary import get_default_intri as f
    return f(width, height, device, dtype, ratio)


def remove_cheirality(R, t, points1, points2, focal_length=None, principal_point=None):
    from .estimate_preliminary import remove_cheirality as f
    return f(R, t, points1, points2, focal_length, principal_point)


def calculate_depth_batch(proj_matrices, points3D):
    """utils.py:398-412: proj_matrices (B,3,4), points3D (B,N,3) -> depth (B,N) = third row of P [X; 1] (the operations of
    the reference, in its order)."""
    B, N, _ = points3D.shape
    homo = torch.cat((points3D, torch.ones(B, N, 1, dtype=points3D.dtype, device=points3D.device)), dim=-1)
    return torch.einsum("bij,bkj->bki", proj_matrices, homo)[..., 2]


def triangulate_point_batch(cam1_from_world, cam2_from_world, points1, points2):
    """utils.py:366-395: two-view DLT of the matches points1 / points2 (B,N,2, normalised image coordinates) under the
    (B,3,4) camera pairs -> (B,N,3).  The 4 x 4 systems go through the pair-triangulation kernel
    (``vgg_triangulate_by_pair``: smallest eigenvector of A^T A by Jacobi sweeps) instead of a batched SVD: the same
    point up to rounding, one launch per pair."""
    from ..utils.triangulation import triangulate_by_pair
    out = []
    for b in range(points1.shape[0]):
        ext = torch.stack([cam1_from_world[b], cam2_from_world[b]]).to(torch.float64)[None]
        trk = torch.stack([points1[b], points2[b]]).to(torch.float64)[None]
        out.append(triangulate_by_pair(ext, trk)[0][0])
    return torch.stack(out)


def check_cheirality_batch(R, t, points1, points2):
    """utils.py:415-448: for the B relative poses (R (B,3,3), t (B,3)) of the second camera, how many of the N matches
    triangulate to a depth in (eps, 1000 x baseline) in both cameras -> (valid_nums (B,), points3D (B,N,3))."""
    B = points1.shape[0]
    R64, t64 = R.to(torch.float64), t.to(torch.float64)
    P1 = torch.eye(3, 4, dtype=torch.float64, device=R.device).expand(B, -1, -1)
    P2 = torch.cat([R64, t64[:, :, None]], dim=-1)
    k_min = torch.finfo(torch.float64).eps
    max_depth = 1000.0 * torch.linalg.norm(R64.transpose(-2, -1) @ t64[:, :, None], dim=1)
    X = triangulate_point_batch(P1, P2, points1, points2)
    d1, d2 = calculate_depth_batch(P1, X), calculate_depth_batch(P2, X)
    valid = (d1 > k_min) & (d1 < max_depth) & (d2 > k_min) & (d2 < max_depth)
    return valid.sum(dim=-1), X
