"""Host-side mirror of ``vggsfm/utils/triangulation.py`` (reference) on the HIP kernels.

Same names, arguments and return values as the reference functions used by ``Triangulator`` and the
runners.  Random draws: like the reference, one ``torch.randperm(C(S,2))`` on the HOST (global CPU
RNG) per chunk of ``max_tri_points_num`` slots, consumed in the same order (reference
triangulation.py:712-758, 804-813), so the hypothesis view pairs are identical for the same seed.
"""
import ctypes
import math

import numpy as np
import torch

from .. import _lib
from .triangulation_helpers import generate_combinations


def _launch_chunk(ext, tn_chunk, ivc_chunk, pairs, lo_num, max_angular_error, min_tri_angle):
    """tn_chunk (S,n,2), ivc_chunk (S,n) bool, pairs (H,2) long -> points (n,3), num (n), mask (n,S)."""
    L = _lib.lib()
    dev = tn_chunk.device
    S, n = tn_chunk.shape[0], tn_chunk.shape[1]
    H = pairs.shape[0]
    tn_t = tn_chunk.to(torch.float64).permute(1, 0, 2).contiguous()          # track-major
    ivc_t = ivc_chunk.t().contiguous().to(torch.uint8)
    pairs_d = pairs.to(device=dev, dtype=torch.int32).contiguous()
    pts = torch.empty((n, 3), dtype=torch.float64, device=dev)
    num = torch.empty(n, dtype=torch.int64, device=dev)
    mask = torch.empty((n, S), dtype=torch.uint8, device=dev)
    ws = torch.zeros(int(L.vgg_triangulate_workspace_bytes(S, n, H, lo_num)), dtype=torch.uint8, device=dev)
    thr = ctypes.c_double(2.0 * math.pi + 1e-6)
    for _ in range(2):
        used = thr.value
        _lib.check(L.vgg_triangulate_tracks(_lib.ptr(ext), _lib.ptr(tn_t), _lib.ptr(ivc_t), _lib.ptr(pairs_d), S, n, H,
                                            lo_num, ctypes.c_double(max_angular_error), ctypes.c_double(min_tri_angle),
                                            _lib.ptr(pts), _lib.ptr(num), _lib.ptr(mask), ctypes.byref(thr), _lib.ptr(ws),
                                            _lib.stream_ptr()), "vgg_triangulate_tracks")
        if thr.value == used:
            break
    return pts, num, mask.bool()


def triangulate_tracks_single_chunk(extrinsics, tracks_normalized, max_ransac_iters=256, lo_num=50, max_angular_error=2,
                                    min_tri_angle=1.5, track_vis=None, track_score=None):
    """Reference: triangulation.py:776-956."""
    _lib.require_gpu(extrinsics, tracks_normalized)
    S = tracks_normalized.shape[0]
    ransac_idx = torch.from_numpy(generate_combinations(S))
    if max_ransac_iters > len(ransac_idx):
        max_ransac_iters = len(ransac_idx)
    else:
        ransac_idx = ransac_idx[torch.randperm(len(ransac_idx))[:max_ransac_iters]]      # host RNG, as the reference
    lo_num = lo_num if max_ransac_iters >= lo_num else max_ransac_iters
    if track_score is not None:
        ivc = torch.logical_or(track_vis <= 0.05, track_score <= 0.5)
    else:
        ivc = track_vis <= 0.05
    ext = extrinsics.to(torch.float64).contiguous()
    return _launch_chunk(ext, tracks_normalized, ivc, ransac_idx, lo_num, max_angular_error, min_tri_angle)


def triangulate_tracks(extrinsics, tracks_normalized, max_ransac_iters=256, lo_num=50, max_angular_error=2,
                       min_tri_angle=1.5, track_vis=None, track_score=None, max_tri_points_num=819200):
    """Reference: triangulation.py:677-773.  extrinsics (S,3,4), tracks_normalized (S,N,2), vis/score (S,N)
    -> points (N,3) f64, inlier_num (N) int64, inlier_mask (N,S) bool.
    The chunking only exists to consume the RNG and the chunk-global indicator threshold exactly like the
    reference; the kernel itself has no memory reason to chunk."""
    all_tri_points_num = extrinsics.shape[0] * tracks_normalized.shape[1]
    if all_tri_points_num > max_tri_points_num:
        num_splits = (all_tri_points_num + max_tri_points_num - 1) // max_tri_points_num
        split_tn = torch.chunk(tracks_normalized, num_splits, dim=1)
        split_vis = torch.chunk(track_vis, num_splits, dim=1) if track_vis is not None else [None] * num_splits
        split_score = torch.chunk(track_score, num_splits, dim=1) if track_score is not None else [None] * num_splits
        pts, nums, masks = [], [], []
        for i in range(len(split_tn)):
            p, n, m = triangulate_tracks_single_chunk(extrinsics, split_tn[i], max_ransac_iters, lo_num,
                                                      max_angular_error, min_tri_angle, split_vis[i], split_score[i])
            pts.append(p), nums.append(n), masks.append(m)
        return torch.cat(pts, 0), torch.cat(nums, 0), torch.cat(masks, 0)
    return triangulate_tracks_single_chunk(extrinsics, tracks_normalized, max_ransac_iters, lo_num, max_angular_error,
                                           min_tri_angle, track_vis, track_score)


def triangulate_by_pair(extrinsics, tracks_normalized, eps=1e-12):
    """Reference: triangulation.py:45-135.  extrinsics (1,S,3,4), tracks_normalized (1,S,N,2) ->
    points (S-1,N,3), cheirality (S-1,N) bool, triangulation angle deg (S-1,N).
    Each (query, reference) pair is a two-view DLT: run the LO-RANSAC kernel per pair with the single
    hypothesis (0, s); its point is the two-view DLT point.  Cheirality / angle are cheap tensor ops."""
    assert extrinsics.shape[0] == 1
    ext = extrinsics[0].to(torch.float64)
    tn = tracks_normalized[0].to(torch.float64)
    S, N = tn.shape[0], tn.shape[1]
    dev = tn.device
    pts = torch.empty((S - 1, N, 3), dtype=torch.float64, device=dev)
    no_ivc = torch.zeros((2, N), dtype=torch.bool, device=dev)
    pair01 = torch.tensor([[0, 1]])
    for s in range(1, S):
        e2 = torch.stack([ext[0], ext[s]]).contiguous()
        t2 = torch.stack([tn[0], tn[s]])
        # huge angular tolerance / zero angle threshold: the winner is the unique two-view hypothesis
        p, _, _ = _launch_chunk(e2, t2, no_ivc, pair01, 1, 180.0, -1.0)
        pts[s - 1] = p
    R, t = ext[:, :, :3], ext[:, :, 3]
    centers = -torch.einsum("sji,sj->si", R, t)
    z0 = torch.einsum("j,snj->sn", R[0, 2], pts) + t[0, 2]
    zs = torch.einsum("sj,snj->sn", R[1:, 2], pts) + t[1:, 2][:, None]
    cheirality = ~((z0 <= 0) | (zs <= 0))
    bsq = (centers[1:] - centers[0]).norm(dim=-1) ** 2
    r1 = (pts - centers[0]).norm(dim=-1) ** 2
    r2 = (pts - centers[1:, None]).norm(dim=-1) ** 2
    den = 2.0 * torch.sqrt(r1 * r2)
    nom = r1 + r2 - bsq[:, None]
    bad = den <= eps
    nom = torch.where(bad, torch.ones_like(nom), nom)
    den = torch.where(bad, torch.ones_like(den), den)
    ang = torch.abs(torch.acos(torch.clamp(nom / den, -1.0, 1.0)))
    ang = torch.min(ang, torch.pi - ang) * (180.0 / torch.pi)
    return pts, cheirality, ang
