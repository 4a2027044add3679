"""Host-side mirror of ``vggsfm/utils/triangulation.py`` (reference) on the HIP kernels.

Same names, arguments and return values as the reference functions used by ``Triangulator`` and the
runners.  Random draws: like the reference, one ``torch.randperm(C(S,2))`` on the HOST (global CPU
RNG) per chunk of ``max_tri_points_num`` slots, consumed in the same order (reference
triangulation.py:712-758, 804-813), so the hypothesis view pairs are identical for the same seed.
"""
import ctypes
import math

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
    ransac_idx, lo_num = _draw_pairs(S, max_ransac_iters, lo_num)
    if track_score is not None:
        ivc = torch.logical_or(track_vis <= 0.05, track_score <= 0.5)
    else:
        ivc = track_vis <= 0.05
    ext = extrinsics.to(torch.float64).contiguous()
    return _launch_chunk(ext, tracks_normalized, ivc, ransac_idx, lo_num, max_angular_error, min_tri_angle)


_PAIRS = {}


def _all_pairs(S):
    """All C(S,2) view pairs in the reference's order, built once per view count (the reference rebuilds the list for
    every chunk: 25 times per call at configs[2])."""
    if S not in _PAIRS:
        _PAIRS.clear()
        _PAIRS[S] = torch.from_numpy(generate_combinations(S))
    return _PAIRS[S]


def _draw_pairs(S, max_ransac_iters, lo_num):
    """The hypothesis pairs of ONE reference chunk (triangulation.py:799-816): all C(S,2) pairs, or a
    torch.randperm draw from the global CPU RNG when there are more than max_ransac_iters."""
    ransac_idx = _all_pairs(S)
    if max_ransac_iters > len(ransac_idx):
        max_ransac_iters = len(ransac_idx)
    else:
        ransac_idx = ransac_idx[torch.randperm(len(ransac_idx))[:max_ransac_iters]]      # host RNG, as the reference
    lo_num = lo_num if max_ransac_iters >= lo_num else max_ransac_iters
    return ransac_idx, lo_num


def reference_chunks(S, N, max_tri_points_num=819200):
    """(chunk_size, num_chunks) of the reference's torch.chunk split of the track axis (triangulation.py:712-723)."""
    num_splits = 1
    if S * N > max_tri_points_num:
        num_splits = (S * N + max_tri_points_num - 1) // max_tri_points_num
    chunk_size = -(-N // num_splits) if N > 0 else 1                  # torch.chunk: ceil(N / chunks) per chunk
    return chunk_size, max(1, -(-N // chunk_size))


def triangulate_tracks(extrinsics, tracks_normalized, max_ransac_iters=256, lo_num=50, max_angular_error=2,
                       min_tri_angle=1.5, track_vis=None, track_score=None, max_tri_points_num=819200, chunk_range=None,
                       check_finite=True):
    """Reference: triangulation.py:677-773.  extrinsics (S,3,4), tracks_normalized (S,N,2), vis/score (S,N)
    -> points (N,3) f64, inlier_num (N) int64, inlier_mask (N,S) bool.
    The reference splits the track axis into ceil(S*N/max_tri_points_num) chunks (torch.chunk), each with its own
    randperm draw and its own chunk-global indicator threshold.  Both are reproduced -- the draws are made up
    front, in chunk order, from the same global CPU RNG -- but all chunks run in ONE launch
    (`vgg_triangulate_tracks_chunks`): the kernel has no memory reason to chunk.
    Kernel limits (include/vggsfm_amd.h): max_ransac_iters <= 256 hypotheses per track, lo_num <= 64 (the reference's
    call sites use 256 / 128 and 50).
    chunk_range = (c0, c1): only the reference chunks c0 .. c1-1 are triangulated and the results cover just their tracks
    [c0 chunk_size, min(N, c1 chunk_size)) -- the RNG draws of ALL chunks are still consumed, in order, so that ranks which
    share the work by whole chunks (vggsfm_amd.dist.triangulate_tracks_sharded) reproduce the single-rank result bit for
    bit.
    Error behaviour: non-finite normalised tracks raise ``torch.linalg.LinAlgError`` as in the reference, whose batched
    ``eigh`` fails on the DLT matrix of a view pair with such a ray whether the view is visible or not
    (triangulation_helpers.py:87; checked live against the reference in the CPU suite).  The test is one reduction ENQUEUED in
    front of the launches and read together with the call's one synchronisation at the end (no host round trip before the
    asynchronous chunk groups) -- the error is raised after the kernel has run, nothing is returned.
    `check_finite=False` skips it: the kernel then treats such a track as the reference's NaN mean would -- all its RANSAC
    hypotheses void."""
    if max_ransac_iters > 256 or lo_num > 64 or max_ransac_iters < 1 or lo_num < 1:
        raise ValueError(f"triangulate_tracks: max_ransac_iters={max_ransac_iters} (1..256) / lo_num={lo_num} (1..64) are outside "
                         "what vgg_triangulate_tracks_chunks supports (the reference calls it with 256 or 128, and 50)")
    _lib.require_gpu(extrinsics, tracks_normalized)
    L = _lib.lib()
    S, N = tracks_normalized.shape[0], tracks_normalized.shape[1]
    dev = tracks_normalized.device
    finite = torch.isfinite(tracks_normalized).all().to(torch.int64).reshape(1) if check_finite else None   # (device, not read yet)
    rng_at_entry = torch.get_rng_state() if check_finite else None         # (5 KB host copy; for the error path below)
    chunk_size, total_chunks = reference_chunks(S, N, max_tri_points_num)
    c0, c1 = (0, total_chunks) if chunk_range is None else (max(0, int(chunk_range[0])), min(total_chunks, int(chunk_range[1])))
    c1 = max(c0, c1)
    t0, t1 = min(N, c0 * chunk_size), min(N, c1 * chunk_size)
    n_loc, nc = t1 - t0, c1 - c0                      # tracks / chunks this call triangulates
    # device-side preparation first (asynchronous): it runs under the host-side draws below
    sl = slice(t0, t1)
    if track_score is not None:
        ivc = torch.logical_or(track_vis[:, sl] <= 0.05, track_score[:, sl] <= 0.5)
    elif track_vis is not None:
        ivc = track_vis[:, sl] <= 0.05
    else:
        ivc = torch.zeros((S, n_loc), dtype=torch.bool, device=dev)
    ext = extrinsics.to(torch.float64).contiguous()
    tn_t = tracks_normalized[:, sl].to(torch.float64).permute(1, 0, 2).contiguous()          # track-major
    ivc_t = ivc.t().contiguous().to(torch.uint8)
    pts = torch.empty((n_loc, 3), dtype=torch.float64, device=dev)
    num = torch.empty(n_loc, dtype=torch.int64, device=dev)
    mask = torch.empty((n_loc, S), dtype=torch.uint8, device=dev)
    first_thr = 2.0 * math.pi + 1e-6
    thr_dev = torch.full((max(nc, 1),), first_thr, dtype=torch.float64, device=dev)
    gmax_dev = torch.zeros(max(nc, 1), dtype=torch.int64, device=dev)
    centers = torch.empty(3 * S, dtype=torch.float64, device=dev)
    # The hypothesis pairs are drawn on the host, chunk by chunk, from the global CPU RNG (one torch.randperm(C(S,2)) per chunk,
    # as the reference does: ~0.2 ms each at 200 views, 25 chunks at configs[2]) -- and every GROUP of chunks is enqueued
    # (vgg_triangulate_tracks_chunks_enqueue, no synchronisation) as soon as its pairs are on the device, so the draws of the
    # next group run while the GPU works on this one.  The pairs go up on a side stream: a blocking copy on the launch stream
    # would wait for the kernel in front of it.
    groups = min(nc, ASYNC_GROUPS) if nc >= 4 else 1
    per_group = -(-nc // groups) if nc else 1
    pairs_dev, lo, H = None, None, None
    main = torch.cuda.current_stream(dev)
    side = _side_stream(dev)
    pending, g_first = [], c0
    for c in range(total_chunks):
        idx, lo_c = _draw_pairs(S, max_ransac_iters, lo_num)                   # RNG consumed in chunk order, for ALL chunks
        lo = lo_c if lo is None else lo
        if not (c0 <= c < c1) or n_loc == 0:
            continue
        pending.append(idx)
        if len(pending) < per_group and c != c1 - 1:
            continue
        host = torch.stack(pending).to(torch.int32)                            # (G,H,2)
        if pairs_dev is None:
            H = host.shape[1]
            pairs_dev = torch.empty((nc, H, 2), dtype=torch.int32, device=dev)
            side.wait_stream(main)
        g0, g1 = g_first - c0, c - c0 + 1
        with torch.cuda.stream(side):
            pairs_dev[g0:g1].copy_(host)
        main.wait_stream(side)
        ta, tb = min(n_loc, g0 * chunk_size), min(n_loc, g1 * chunk_size)
        if tb > ta:
            _lib.check(L.vgg_triangulate_tracks_chunks_enqueue(
                _lib.ptr(ext), _lib.ptr(tn_t[ta:tb]), _lib.ptr(ivc_t[ta:tb]), _lib.ptr(pairs_dev[g0:g1]), S, tb - ta, H, g1 - g0,
                chunk_size, lo, ctypes.c_double(max_angular_error), ctypes.c_double(min_tri_angle), _lib.ptr(pts[ta:tb]),
                _lib.ptr(num[ta:tb]), _lib.ptr(mask[ta:tb]), _lib.ptr(thr_dev[g0:g1]), _lib.ptr(gmax_dev[g0:g1]), _lib.ptr(centers),
                _lib.stream_ptr()), "vgg_triangulate_tracks_chunks_enqueue")
        pending, g_first = [], c + 1
    def raise_if_nonfinite(flag):
        if flag is not None and not int(flag):
            # The reference fails inside its FIRST chunk, after that chunk's randperm draw and before any other
            # (triangulation.py:812 -> triangulation_helpers.py:87).  This call has drawn for every chunk by now: put the
            # global CPU RNG where a caller that catches the error would find it there.
            torch.set_rng_state(rng_at_entry)
            if N > 0:
                _draw_pairs(S, max_ransac_iters, lo_num)
            raise torch.linalg.LinAlgError("triangulate_tracks: non-finite normalised track coordinates (the reference's "
                                           "linalg.eigh fails on the DLT matrices of such views)")
    if n_loc == 0 or nc == 0:
        raise_if_nonfinite(None if finite is None else finite.cpu()[0])
        return pts, num, mask.bool()
    # the one synchronisation of the call: the largest mean inlier error of every chunk (the indicator's chunk-global threshold)
    # and, behind it in the same copy, the finiteness flag of the input
    back = (gmax_dev if finite is None else torch.cat([gmax_dev, finite])).cpu().numpy()
    raise_if_nonfinite(None if finite is None else back[-1])
    measured = back[:max(nc, 1)].copy().view("float64") + 1e-6
    if not (measured == first_thr).all():
        # (pathological: every hypothesis of a chunk has an inlier -- the threshold the launch assumed was not the chunk's
        #  maximum; run again with the measured ones, synchronously, as rounds 1-3 did)
        ws = torch.zeros(int(L.vgg_triangulate_chunks_workspace_bytes(S, nc)), dtype=torch.uint8, device=dev)
        thr = (ctypes.c_double * nc)(*[float(x) for x in measured])
        for _ in range(2):
            used = list(thr)
            _lib.check(L.vgg_triangulate_tracks_chunks(_lib.ptr(ext), _lib.ptr(tn_t), _lib.ptr(ivc_t), _lib.ptr(pairs_dev), S, n_loc, H,
                                                       nc, chunk_size, lo, ctypes.c_double(max_angular_error),
                                                       ctypes.c_double(min_tri_angle), _lib.ptr(pts), _lib.ptr(num),
                                                       _lib.ptr(mask), thr, _lib.ptr(ws), _lib.stream_ptr()),
                       "vgg_triangulate_tracks_chunks")
            if list(thr) == used:
                break
    return pts, num, mask.bool()


ASYNC_GROUPS = 5          # chunk groups a multi-chunk triangulate_tracks call is enqueued in (host draws overlap the GPU)
_SIDE = {}


def _side_stream(dev):
    key = (dev.type, dev.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


def triangulate_by_pair(extrinsics, tracks_normalized, eps=1e-12):
    """Reference: triangulation.py:45-135.  extrinsics (1,S,3,4), tracks_normalized (1,S,N,2) ->
    points (S-1,N,3), cheirality (S-1,N) bool, triangulation angle deg (S-1,N).
    One thread per (frame, track) two-view DLT (`vgg_triangulate_by_pair`, bit-identical to the two-view
    hypothesis (0, s) of the LO-RANSAC kernel); cheirality / angle are cheap tensor ops."""
    assert extrinsics.shape[0] == 1
    ext = extrinsics[0].to(torch.float64)
    tn = tracks_normalized[0].to(torch.float64)
    S, N = tn.shape[0], tn.shape[1]
    dev = tn.device
    _lib.require_gpu(ext, tn)
    ext, tn = ext.contiguous(), tn.contiguous()
    pts = torch.empty((S - 1, N, 3), dtype=torch.float64, device=dev)
    _lib.check(_lib.lib().vgg_triangulate_by_pair(_lib.ptr(ext), _lib.ptr(tn), S, N, _lib.ptr(pts), _lib.stream_ptr()),
               "vgg_triangulate_by_pair")
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


# ==========================================================================================
# BA / pose-refinement drivers (reference: triangulation.py:138-647, 1020-1242).  The reference marshals
# tensors into pycolmap objects with O(S*P) Python loops and back; here everything stays on the device.
# ==========================================================================================
from .. import ba as _ba                                    # noqa: E402
from ..ba_options import AbsolutePoseEstimationOptions, AbsolutePoseRefinementOptions, BundleAdjustmentOptions  # noqa: E402
from ..pose import absolute_pose_estimation_batch, pose_refinement_batch  # noqa: E402
from .triangulation_helpers import cam_from_img, filter_all_points3D, prepare_ba_options  # noqa: E402


def get_valid_frame_mask(intrinsics, extrinsics, extra_params, scale):
    """Reference: triangulation.py:1222-1242."""
    valid = torch.logical_and(intrinsics[:, 0, 0] >= 0.1 * scale, intrinsics[:, 0, 0] <= 30 * scale)
    if extra_params is not None:
        if extra_params.dim() == 1:
            extra_params = extra_params[:, None]
        valid = torch.logical_and(valid, (extra_params.abs() <= 1.0).all(dim=-1))
    return torch.logical_and(valid, (extrinsics[:, :, 3].abs() <= 30).all(-1))


def _intr_params(intrinsics, extra_params):
    """(S,3,3) K [+ (S,1) k] -> (S,4) COLMAP parameter rows f,cx,cy,k."""
    S = intrinsics.shape[0]
    p = torch.zeros((S, 4), dtype=torch.float64, device=intrinsics.device)
    p[:, 0] = intrinsics[:, 0, 0]
    p[:, 1] = intrinsics[:, 0, 2]
    p[:, 2] = intrinsics[:, 1, 2]
    if extra_params is not None:
        p[:, 3] = extra_params.to(torch.float64)[:, 0]
    return p


def _from_intr_params(p, camera_type):
    S = p.shape[0]
    K = torch.zeros((S, 3, 3), dtype=torch.float64, device=p.device)
    K[:, 0, 0] = K[:, 1, 1] = p[:, 0]
    K[:, 0, 2], K[:, 1, 2], K[:, 2, 2] = p[:, 1], p[:, 2], 1.0
    extra = p[:, 3:4].clone() if camera_type == "SIMPLE_RADIAL" else None
    return K, extra


def _check_camera_type(camera_type):
    if camera_type not in ("SIMPLE_PINHOLE", "SIMPLE_RADIAL"):
        raise ValueError(f"Camera type {camera_type} is not supported yet")


def _refine_frames(extrinsics, intrinsics, extra_params, tracks2D, points3D, inlier, refine_ids, shared_camera,
                   camera_type, min_inliers, first_frame_refines_intrinsics):
    """Common body of init_refine_pose / refine_pose: what the reference's per-frame loop computes.
    Returns (extrinsics, intrinsics, extra_params, refined (S,) bool)."""
    S = extrinsics.shape[0]
    dev = tracks2D.device
    counts = inlier.sum(dim=1)
    enough = counts > min_inliers
    want = torch.zeros(S, dtype=torch.bool, device=dev)
    want[torch.as_tensor(refine_ids, dtype=torch.long, device=dev)] = True
    todo = want & enough
    params = _intr_params(intrinsics, extra_params)
    ext = extrinsics.to(torch.float64).clone()
    flags = torch.full((S,), 3, dtype=torch.uint8, device=dev)
    if shared_camera:
        # one pycolmap.Camera object serves every frame: frame 0 may refine it, later frames keep it fixed
        # (reference :373-375) and see the refined values
        flags[1:] = 0
        if not first_frame_refines_intrinsics:
            flags[0] = 0
        if bool(todo[0]):
            ext, params, _ = pose_refinement_batch(ext, params, tracks2D, points3D, inlier, [0], camera_type, flags)
        params = params[0:1].expand(S, -1).contiguous()
        rest = torch.nonzero(todo[1:]).squeeze(1) + 1
        ext, params, _ = pose_refinement_batch(ext, params, tracks2D, points3D, inlier, rest, camera_type, flags)
        params = params[0:1].expand(S, -1).contiguous()
    else:
        ids = torch.nonzero(todo).squeeze(1)
        ext, params, _ = pose_refinement_batch(ext, params, tracks2D, points3D, inlier, ids, camera_type, flags)
    K, extra = _from_intr_params(params, camera_type)
    return ext, K, extra, todo


def init_refine_pose(extrinsics, intrinsics, extra_params, inlier, points3D, tracks, valid_track_mask_init, image_size,
                     init_idx, max_reproj_error=12, shared_camera=False, camera_type="SIMPLE_PINHOLE"):
    """Reference: triangulation.py:482-647.  Refines every frame except the initial pair on the fixed
    initial point cloud (frames with <= 50 inliers are left alone)."""
    _check_camera_type(camera_type)
    S = extrinsics.shape[0]
    P = tracks.shape[1]
    assert len(intrinsics) == S and inlier.shape[0] == S - 1 and inlier.shape[1] == P and len(valid_track_mask_init) == P
    inl = torch.cat([torch.ones_like(inlier[0:1]), inlier], dim=0)[:, valid_track_mask_init]
    tracks2D = tracks[:, valid_track_mask_init]
    refine_ids = [r for r in range(S) if r != 0 and r != init_idx + 1]
    few = torch.nonzero((inl.sum(1) <= 50)[torch.as_tensor(refine_ids, dtype=torch.long, device=inl.device)]).squeeze(1) \
        if refine_ids else []
    for k in (few.tolist() if len(refine_ids) else []):
        print("This frame only has inliers:", int(inl[refine_ids[k]].sum()))
    ext, K, extra, _ = _refine_frames(extrinsics, intrinsics, extra_params, tracks2D, points3D, inl, refine_ids, shared_camera,
                                      camera_type, 50, first_frame_refines_intrinsics=False)
    scale = image_size.max()
    valid = get_valid_frame_mask(K, ext, extra, scale)
    if (~valid).sum() > 0:
        print("some frames are invalid after BA refinement")
        ext[~valid] = extrinsics[~valid].to(ext.dtype)
        K[~valid] = intrinsics[~valid].to(ext.dtype)
        if extra_params is not None:
            extra[~valid] = extra_params[~valid].to(ext.dtype)
    return ext, K, extra, valid


def refine_pose(extrinsics, intrinsics, extra_params, inlier, points3D, tracks, valid_track_mask, image_size,
                shared_camera=False, max_reproj_error=12, camera_type="SIMPLE_PINHOLE", force_estimate=False):
    """Reference: triangulation.py:260-479.  Inliers = given mask AND reprojection error <= 12 px with
    positive depth; frames with > 100 such inliers are refined.  With `force_estimate` the other frames (and frames
    refined to a focal length outside [0.1, 30] x image size) go through ``absolute_pose_estimation_batch`` -- the
    device restatement of pycolmap.absolute_pose_estimation (P3P RANSAC + refinement; random, parity unpinned).
    `shared_camera` semantics of that fallback: the focal length estimated for a frame stays with that frame (row) of
    the returned intrinsics; only frame 0 refines intrinsics afterwards and the next shared-camera BA reads row 0 alone
    (tensor_to_pycolmap.py:76-107), exactly as in the reference.  (The reference mutates ONE pycolmap camera object
    frame after frame, so there a frame's estimate also becomes the starting focal of the frames processed after it;
    here all frames are estimated concurrently from the same starting intrinsics.)"""
    _check_camera_type(camera_type)
    S = extrinsics.shape[0]
    P = tracks.shape[1]
    assert len(intrinsics) == S and inlier.shape[0] == S and inlier.shape[1] == P and len(valid_track_mask) == P
    empty = points3D.abs().sum(-1) <= 0
    if empty.sum() > 0:
        tmp = valid_track_mask.clone()
        tmp[valid_track_mask] = ~empty
        valid_track_mask = tmp
        points3D = points3D[~empty]
    tracks2D = tracks[:, valid_track_mask]
    _, reproj_inlier = filter_all_points3D(points3D, tracks2D, extrinsics, intrinsics, extra_params,
                                           max_reproj_error=max_reproj_error, check_triangle=False, return_detail=True,
                                           hard_max=-1, behind_value=1e9)
    inl = torch.logical_and(inlier[:, valid_track_mask], reproj_inlier)
    counts = inl.sum(1)
    for r in torch.nonzero(counts <= 100).squeeze(1).tolist():
        print(f"Frame {r} only has {int(counts[r])} geo_vis inliers")
    ext, K, extra, refined = _refine_frames(extrinsics, intrinsics, extra_params, tracks2D, points3D, inl, list(range(S)),
                                            shared_camera, camera_type, 100, first_frame_refines_intrinsics=True)
    if force_estimate:
        # reference :397-432: frames with <= 100 inliers, or refined to a wild focal length, are re-estimated from the
        # visibility-only matches by absolute_pose_estimation (P3P RANSAC over 30 focal length factors + refinement);
        # with <= 50 such matches -- or when the estimate fails -- all points are used
        scale_px = float(image_size.max())
        focal = K[:, 0, 0]
        est = (counts <= 100) | (refined & ((focal < 0.1 * scale_px) | (focal > 30 * scale_px)))
        if bool(est.any()):
            dev = tracks.device
            cand = inlier[:, valid_track_mask].bool().clone()
            few = cand.sum(1) <= 50
            for r in torch.nonzero(est).squeeze(1).tolist():
                print(f"Estimating absolute poses by visible matches for frame {r}" if not bool(few[r]) else
                      f"Warning! Estimating absolute poses by non visible matches for frame {r}")
            cand[few] = True
            flags = torch.full((S,), 3, dtype=torch.uint8, device=dev)
            if shared_camera:
                flags[1:] = 0
            estopt = AbsolutePoseEstimationOptions(estimate_focal_length=True)
            estopt.ransac.max_error = float(max_reproj_error)
            refopt = AbsolutePoseRefinementOptions(refine_focal_length=True, refine_extra_params=True)
            params = _intr_params(K, extra)
            ids = torch.nonzero(est).squeeze(1)
            ext, params, ok, _, _ = absolute_pose_estimation_batch(ext, params, tracks2D, points3D, cand, ids, camera_type, flags,
                                                                   estopt, refopt)
            retry = ids[~ok[ids] & ~few[ids]]
            if retry.numel():                                   # estanswer is None -> once more with every point
                ext, params, _, _, _ = absolute_pose_estimation_batch(ext, params, tracks2D, points3D, torch.ones_like(cand), retry,
                                                                      camera_type, flags, estopt, refopt)
            K, extra = _from_intr_params(params, camera_type)
    scale = image_size.max()
    valid = get_valid_frame_mask(K, ext, extra, scale)
    if (~valid).sum() > 0:
        print("some frames are invalid after BA refinement")
        ext[~valid] = extrinsics[~valid].to(ext.dtype)
        K[~valid] = intrinsics[~valid].to(ext.dtype)
        if extra_params is not None:
            extra[~valid] = extra_params[~valid].to(ext.dtype)
    return ext, K, extra, valid


def init_BA(extrinsics, intrinsics, extra_params, tracks, points_3d_pair, inlier, image_size, shared_camera=False,
            init_max_reproj_error=0.5, camera_type="SIMPLE_PINHOLE"):
    """Reference: triangulation.py:138-257: two-view BA on the best initial pair, then reprojection filter."""
    _check_camera_type(camera_type)
    init_idx = torch.argmax(inlier.sum(dim=-1)).item()
    init_indices = [0, init_idx + 1]
    toBA_ext, toBA_K = extrinsics[init_indices], intrinsics[init_indices]
    toBA_extra = extra_params[init_indices] if extra_params is not None else None
    toBA_tracks = tracks[init_indices]
    toBA_points3D = points_3d_pair[init_idx]
    toBA_masks = inlier[init_idx].unsqueeze(0)
    toBA_masks = torch.cat([torch.ones_like(toBA_masks), toBA_masks], dim=0)
    toBA_valid = toBA_masks.sum(dim=0) >= 2
    toBA_masks = toBA_masks[:, toBA_valid]
    toBA_points3D = toBA_points3D[toBA_valid]
    toBA_tracks = toBA_tracks[:, toBA_valid]
    pts_opt, ext_opt, K_opt, extra_opt, summary = _ba.bundle_adjustment(
        toBA_points3D, toBA_ext, toBA_K, toBA_tracks, toBA_masks, image_size, toBA_extra, shared_camera, camera_type,
        prepare_ba_options())
    valid3D, _ = filter_all_points3D(pts_opt, toBA_tracks, ext_opt, K_opt, extra_opt, check_triangle=False,
                                     max_reproj_error=init_max_reproj_error)
    pts_opt = pts_opt[valid3D]
    filtered = toBA_valid.clone()
    filtered[toBA_valid] = valid3D
    extrinsics[init_indices] = ext_opt.to(extrinsics.dtype)          # in place, like the reference (:243-246)
    intrinsics[init_indices] = K_opt.to(intrinsics.dtype)
    if extra_params is not None:
        extra_params[init_indices] = extra_opt.to(extra_params.dtype)
    return pts_opt, extrinsics, intrinsics, extra_params, filtered, summary, init_idx


def _revert_negative_focal(ext, K, extra, ext_old, K_old, extra_old):
    bad = K[:, 0, 0] < 0
    if bad.any():
        ext[bad], K[bad] = ext_old[bad].to(ext.dtype), K_old[bad].to(K.dtype)
        if extra is not None:
            extra[bad] = extra_old[bad].to(extra.dtype)
    return ext, K, extra


def global_BA(triangulated_points, valid_tracks, pred_tracks, inlier_mask, extrinsics, intrinsics, extra_params, image_size,
              shared_camera=False, camera_type="SIMPLE_PINHOLE"):
    """Reference: triangulation.py:1020-1073 (prepare_ba_options, then normalize(5, 0.1, 0.9, True))."""
    BA_points = triangulated_points[valid_tracks]
    BA_tracks = pred_tracks[:, valid_tracks]
    BA_masks = inlier_mask[valid_tracks].transpose(0, 1)
    pts, ext, K, extra, summary = _ba.bundle_adjustment(BA_points, extrinsics, intrinsics, BA_tracks, BA_masks, image_size,
                                                        extra_params, shared_camera, camera_type, prepare_ba_options(),
                                                        normalize=True)
    ext, K, extra = _revert_negative_focal(ext, K, extra, extrinsics, intrinsics, extra_params)
    return pts, ext, K, extra, summary


def iterative_global_BA(pred_tracks, intrinsics, extrinsics, pred_vis, pred_score, valid_tracks, points3D_opt, image_size,
                        shared_camera=False, min_valid_track_length=2, max_reproj_error=1, ba_options=None, lastBA=False,
                        camera_type="SIMPLE_PINHOLE", extra_params=None):
    """Reference: triangulation.py:1076-1209: retriangulate (128 hypotheses) -> filter -> BA -> filter."""
    tn = cam_from_img(pred_tracks, intrinsics, extra_params)
    best_pts, _, _ = triangulate_tracks(extrinsics, tn, track_vis=pred_vis, track_score=pred_score, max_ransac_iters=128)
    best_pts[valid_tracks] = points3D_opt
    _, filt = filter_all_points3D(best_pts, pred_tracks, extrinsics, intrinsics, extra_params=extra_params,
                                  max_reproj_error=max_reproj_error, return_detail=True)
    valid_tracks = filt.sum(dim=0) >= min_valid_track_length
    BA_points = best_pts[valid_tracks]
    BA_tracks = pred_tracks[:, valid_tracks]
    BA_masks = filt[:, valid_tracks]
    if ba_options is None:
        ba_options = BundleAdjustmentOptions()
    pts, ext, K, extra, summary = _ba.bundle_adjustment(BA_points, extrinsics, intrinsics, BA_tracks, BA_masks, image_size,
                                                        extra_params, shared_camera, camera_type, ba_options, normalize=True)
    ext, K, extra = _revert_negative_focal(ext, K, extra, extrinsics, intrinsics, extra_params)
    _, filt2 = filter_all_points3D(pts, pred_tracks[:, valid_tracks], ext, K, extra_params=extra,
                                   max_reproj_error=max_reproj_error, return_detail=True)
    after = filt2.sum(dim=0) >= min_valid_track_length
    vt = valid_tracks.clone()
    vt[valid_tracks] = after
    pts = pts[after]
    BA_inlier_masks = filt2[:, after]
    return pts, ext, K, extra, vt, BA_inlier_masks, summary


def triangulate_extra_points(extra_track, extra_vis, extra_score, extrinsics, intrinsics, extra_params=None,
                             max_reproj_error=4):
    """The geometry of the dense-point pass of ``VGGSfMRunner.triangulate_extra_points`` (vggsfm/runners/runner.py:696-736)
    for one neighbourhood of frames: undistort, LO-RANSAC triangulate with the refined cameras, keep points with more
    than 3 inlier views that also pass the reprojection / triangulation-angle filter.
    extra_track (S,N,2) px, extra_vis / extra_score (S,N), cameras of the S neighbour frames.
    Returns (points3D (N,3), valid (N,) bool)."""
    tn = cam_from_img(extra_track, intrinsics, extra_params)
    pts, inlier_num, _ = triangulate_tracks(extrinsics, tn, track_vis=extra_vis, track_score=extra_score)
    valid, _ = filter_all_points3D(pts, extra_track, extrinsics, intrinsics, extra_params=extra_params,
                                   max_reproj_error=max_reproj_error)
    return pts, (inlier_num > 3) & valid
