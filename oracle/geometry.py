"""ORACLE (test infrastructure, never shipped, never on the product path).

CPU restatement, in plain numpy float64, of the *torch half* of VGGSfM's geometry hot
path.  Every function cites the reference lines it restates (paths under
``/root/reference``).  Pinning: ``tests/test_oracle_golden.py`` checks each function
against ``tests/golden/*.npz``, which ``oracle/gen_golden.py`` produced by importing and
running the reference's own torch functions on the CPU in the build container.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.

Tie-break contract (SURVEY §7 hard part 2): the reference sorts RANSAC hypotheses with
``torch.sort(descending=True)`` whose tie order is implementation-defined; this oracle, the
golden generator and the HIP kernel all use *stable* descending order (ties keep the lower
hypothesis index first).
"""
import itertools

import numpy as np

PI = float(np.pi)


# --------------------------------------------------------------------------------------
# distortion / normalisation            vggsfm/utils/distortion.py, triangulation_helpers.py
# --------------------------------------------------------------------------------------
def apply_distortion(extra_params, u, v):
    """vggsfm/utils/distortion.py:102-159 (1-, 2-, 4-parameter models). extra (B,k); u,v (B,N)."""
    k = extra_params.shape[1]
    u2, v2 = u * u, v * v
    r2 = u2 + v2
    if k == 1:
        radial = extra_params[:, 0][:, None] * r2
        du, dv = u * radial, v * radial
    elif k == 2:
        radial = extra_params[:, 0][:, None] * r2 + extra_params[:, 1][:, None] * r2 * r2
        du, dv = u * radial, v * radial
    elif k == 4:
        k1, k2, p1, p2 = (extra_params[:, i][:, None] for i in range(4))
        uv = u * v
        radial = k1 * r2 + k2 * r2 * r2
        du = u * radial + 2 * p1 * uv + p2 * (r2 + 2 * u2)
        dv = v * radial + 2 * p2 * uv + p1 * (r2 + 2 * v2)
    else:
        raise ValueError("Unsupported number of distortion parameters")
    return u + du, v + dv


def iterative_undistortion(extra_params, tracks_normalized, max_iterations=100, max_step_norm=1e-10,
                           rel_step_size=1e-6):
    """vggsfm/utils/distortion.py:27-99: Newton with central-difference Jacobian; the stop
    test is the max squared step over the WHOLE tensor, so every element runs the same
    number of iterations.  Returns (undistorted (B,N,2), iterations_run)."""
    u = tracks_normalized[..., 0].copy()
    v = tracks_normalized[..., 1].copy()
    ou, ov = u.copy(), v.copy()
    eps = np.finfo(u.dtype).eps
    iters = 0
    for _ in range(max_iterations):
        iters += 1
        ud, vd = apply_distortion(extra_params, u, v)
        dx, dy = ou - ud, ov - vd
        su = np.maximum(np.abs(u) * rel_step_size, eps)
        sv = np.maximum(np.abs(v) * rel_step_size, eps)
        pu = apply_distortion(extra_params, u + su, v)
        mu = apply_distortion(extra_params, u - su, v)
        pv = apply_distortion(extra_params, u, v + sv)
        mv = apply_distortion(extra_params, u, v - sv)
        j00 = (pu[0] - mu[0]) / (2 * su) + 1
        j01 = (pv[0] - mv[0]) / (2 * sv)
        j10 = (pu[1] - mu[1]) / (2 * su)
        j11 = (pv[1] - mv[1]) / (2 * sv) + 1
        # 2x2 solve with partial pivoting (LAPACK gesv semantics of torch.linalg.solve)
        swap = np.abs(j10) > np.abs(j00)
        a00 = np.where(swap, j10, j00)
        a01 = np.where(swap, j11, j01)
        a10 = np.where(swap, j00, j10)
        a11 = np.where(swap, j01, j11)
        b0 = np.where(swap, dy, dx)
        b1 = np.where(swap, dx, dy)
        l10 = a10 / a00
        u11 = a11 - l10 * a01
        y1 = b1 - l10 * b0
        d1 = y1 / u11
        d0 = (b0 - a01 * d1) / a00
        u = u + d0
        v = v + d1
        if np.max(d0 * d0 + d1 * d1) < max_step_norm:
            break
    return np.stack([u, v], axis=-1), iters


def cam_from_img(pred_tracks, intrinsics, extra_params=None):
    """vggsfm/utils/triangulation_helpers.py:398-428.  tracks (S,N,2), K (S,3,3)."""
    pp = np.stack([intrinsics[:, 0, 2], intrinsics[:, 1, 2]], -1)[:, None, :]
    fl = np.stack([intrinsics[:, 0, 0], intrinsics[:, 1, 1]], -1)[:, None, :]
    tn = (pred_tracks - pp) / fl
    if extra_params is not None:
        tn, _ = iterative_undistortion(extra_params, tn)
    return tn


def project_3D_points(points3D, extrinsics, intrinsics=None, extra_params=None, return_points_cam=False,
                      only_points_cam=False):
    """vggsfm/utils/triangulation_helpers.py:311-395 (project_3D_points + img_from_cam).
    Returns (S,P,2) [, (S,3,P)]; NaNs in the pixel output replaced by 0."""
    R = extrinsics[:, :, :3]
    t = extrinsics[:, :, 3]
    pc = np.einsum("sij,pj->sip", R, points3D) + t[:, :, None]      # (S,3,P)
    if only_points_cam:
        return pc
    with np.errstate(divide="ignore", invalid="ignore"):
        u = pc[:, 0] / pc[:, 2]
        v = pc[:, 1] / pc[:, 2]
        if extra_params is not None:
            u, v = apply_distortion(extra_params, u, v)
        x = intrinsics[:, 0, 0][:, None] * u + intrinsics[:, 0, 1][:, None] * v + intrinsics[:, 0, 2][:, None]
        y = intrinsics[:, 1, 0][:, None] * u + intrinsics[:, 1, 1][:, None] * v + intrinsics[:, 1, 2][:, None]
    # torch.nan_to_num(nan=0): +-inf become +-finfo.max, exactly numpy's default
    p2 = np.nan_to_num(np.stack([x, y], -1), nan=0.0)
    if return_points_cam:
        return p2, pc
    return p2


# --------------------------------------------------------------------------------------
# angles                                             vggsfm/utils/triangulation_helpers.py
# --------------------------------------------------------------------------------------
def proj_centers(extrinsics):
    """-R^T t, triangulation_helpers.py:485,531."""
    return -np.einsum("sji,sj->si", extrinsics[:, :, :3], extrinsics[:, :, 3])


def _tri_angle_deg(r1sq, r2sq, bsq, eps=1e-12):
    """law-of-cosines triangulation angle, min(theta, pi-theta), degrees
    (triangulation_helpers.py:503-519 / 568-586; triangulation.py:111-133)."""
    den = 2.0 * np.sqrt(r1sq * r2sq)
    nom = r1sq + r2sq - bsq
    bad = den <= eps
    nom = np.where(bad, 1.0, nom)
    den = np.where(bad, 1.0, den)
    c = np.clip(nom / den, -1.0, 1.0)
    th = np.abs(np.arccos(c))
    th = np.minimum(th, PI - th)
    return th * (180.0 / PI)


def tri_angles_all_pairs(extrinsics, points3D):
    """calculate_triangulation_angle_exhaustive (triangulation_helpers.py:524-587) -> (S*S,P) deg."""
    c = proj_centers(extrinsics)
    S = len(c)
    c1 = np.repeat(c, S, axis=0)
    c2 = np.tile(c, (S, 1))
    bsq = np.linalg.norm(c1 - c2, axis=-1) ** 2
    r1 = np.linalg.norm(points3D[None] - c1[:, None], axis=-1) ** 2
    r2 = np.linalg.norm(points3D[None] - c2[:, None], axis=-1) ** 2
    return _tri_angle_deg(r1, r2, bsq[:, None])


def angular_error(point2D, point3D, cam_from_world):
    """calculate_normalized_angular_error_batched (triangulation_helpers.py:431-472).
    point2D (S,N,2) normalised rays; point3D (H,N,3); cams (S,3,4) -> (H,S,N) radians."""
    ray1 = np.concatenate([point2D, np.ones_like(point2D[..., :1])], -1)
    ray1 = ray1 / np.maximum(np.linalg.norm(ray1, axis=-1, keepdims=True), 1e-12)
    ray2 = np.einsum("sij,hnj->hsni", cam_from_world[:, :, :3], point3D) + cam_from_world[None, :, None, :, 3]
    ray2 = ray2 / np.maximum(np.linalg.norm(ray2, axis=-1, keepdims=True), 1e-12)
    c = np.clip((ray1[None] * ray2).sum(-1), -1.0, 1.0)
    return np.arccos(c)


# --------------------------------------------------------------------------------------
# DLT                                                    triangulation_helpers.py:27-131
# --------------------------------------------------------------------------------------
def dlt_batched(cams_from_world, points, mask=None, compute_tri_angle=False, check_cheirality=False):
    """triangulate_multi_view_point_batched.  cams (B,n,3,4), points (B,n,2), mask (B,n).
    Returns X (B,3) [, tri-angle 'any pair >= thr' helper angles (B,n*n)][, invalid_cheirality (B)]."""
    B, n, _ = points.shape
    ph = np.concatenate([points, np.ones((B, n, 1))], -1)
    pn = ph / np.linalg.norm(ph, axis=-1, keepdims=True)
    outer = np.einsum("bni,bnj->bnij", pn, pn)
    terms = cams_from_world - np.einsum("bnij,bnik->bnjk", outer, cams_from_world)
    if mask is not None:
        terms = terms * mask[:, :, None, None]
    A = np.einsum("bnij,bnik->bjk", terms, terms)
    _, vec = np.linalg.eigh(A)
    v0 = vec[:, :, 0]
    with np.errstate(divide="ignore", invalid="ignore"):
        X = (v0 / v0[:, -1:])[:, :3]
    out = [X]
    if compute_tri_angle:
        R = cams_from_world[:, :, :, :3]
        t = cams_from_world[:, :, :, 3]
        c = -np.einsum("bnji,bnj->bni", R, t)
        c1 = np.repeat(c, n, axis=1)
        c2 = np.tile(c, (1, n, 1))
        bsq = np.linalg.norm(c1 - c2, axis=-1) ** 2
        with np.errstate(invalid="ignore"):
            r1 = np.linalg.norm(X[:, None] - c1, axis=-1) ** 2
            r2 = np.linalg.norm(X[:, None] - c2, axis=-1) ** 2
            out.append(_tri_angle_deg(r1, r2, bsq))
    if check_cheirality:
        with np.errstate(invalid="ignore"):
            z = np.einsum("bnj,bj->bn", cams_from_world[:, :, 2, :3], X) + cams_from_world[:, :, 2, 3]
            out.append((z <= 0).any(axis=1))
    return out[0] if len(out) == 1 else tuple(out)


def triangulate_by_pair(extrinsics, tracks_normalized):
    """vggsfm/utils/triangulation.py:45-135 for B=1.  ext (S,3,4), tracks (S,N,2) ->
    points (S-1,N,3), cheirality (S-1,N) bool, tri-angle deg (S-1,N)."""
    S, N, _ = tracks_normalized.shape
    pts = np.zeros((S - 1, N, 3))
    che = np.zeros((S - 1, N), bool)
    ang = np.zeros((S - 1, N))
    c = proj_centers(extrinsics)
    for s in range(1, S):
        cams = np.broadcast_to(np.stack([extrinsics[0], extrinsics[s]])[None], (N, 2, 3, 4))
        p2 = np.stack([tracks_normalized[0], tracks_normalized[s]], axis=1)
        X, inv = dlt_batched(cams, p2, check_cheirality=True)
        pts[s - 1] = X
        che[s - 1] = ~inv
        bsq = np.linalg.norm(c[s] - c[0]) ** 2
        with np.errstate(invalid="ignore"):
            r1 = np.linalg.norm(X - c[0], axis=-1) ** 2
            r2 = np.linalg.norm(X - c[s], axis=-1) ** 2
            ang[s - 1] = _tri_angle_deg(r1, r2, bsq)
    return pts, che, ang


# --------------------------------------------------------------------------------------
# filter                                                triangulation_helpers.py:133-307
# --------------------------------------------------------------------------------------
def filter_all_points3D(points3D, points2D, extrinsics, intrinsics, extra_params=None, max_reproj_error=4,
                        min_tri_angle=1.5, check_triangle=True, return_detail=False, hard_max=300):
    """filter_all_points3D_single_chunk semantics (chunking does not change the result)."""
    p2, pc = project_3D_points(points3D, extrinsics, intrinsics, extra_params, return_points_cam=True)
    err = np.linalg.norm(p2 - points2D, axis=-1) ** 2
    err = np.where(pc[:, 2] <= 0, 1e6, err)
    with np.errstate(invalid="ignore"):
        inlier = err <= (max_reproj_error ** 2)
    valid = inlier.sum(0) >= 2
    if hard_max > 0:
        valid &= (np.abs(points3D) <= hard_max).all(-1)
    tri_any_full = None
    if check_triangle:
        idx = np.nonzero(valid)[0]
        tri_any_full = np.zeros_like(valid)
        S = len(extrinsics)
        c = proj_centers(extrinsics)
        # blocked over points to bound memory; same arithmetic as the (S*S,P) tensor
        for lo in range(0, len(idx), 2048):
            sel = idx[lo:lo + 2048]
            ang = tri_angles_all_pairs(extrinsics, points3D[sel]).reshape(S, S, -1)
            inl = inlier[:, sel]
            grid = inl[:, None, :] & inl[None, :, :]
            tri_any_full[sel] = ((ang >= min_tri_angle) & grid).any(axis=(0, 1))
        ret = tri_any_full & valid
    else:
        ret = valid
    detail = None
    if return_detail:
        detail = inlier.copy()
        if check_triangle:
            detail = tri_any_full[None] & detail
    return ret, detail


# --------------------------------------------------------------------------------------
# LO-RANSAC triangulation                          vggsfm/utils/triangulation.py:677-1017
# --------------------------------------------------------------------------------------
def generate_combinations(S):
    """triangulation_helpers.py:638-645: all C(S,2) index pairs in lexicographic order."""
    return np.array(list(itertools.combinations(range(S), 2)), dtype=np.int64).reshape(-1, 2)


def stable_argsort_desc(x):
    """Stable descending order along the last axis (the tie-break contract)."""
    return np.argsort(-x, axis=-1, kind="stable")


def residual_indicator(err, max_residual, nanvalue):
    """vggsfm/two_view_geo/utils.py:63-87.  err (B,H,S).  Returns (indicator, num, mask, thres)."""
    mask = err <= max_residual
    num = mask.sum(-1)
    with np.errstate(divide="ignore", invalid="ignore"):
        # reference: inlier_mask.float() * residuals  (0 * inf would be nan; errors are finite here)
        ind = (mask.astype(np.float32).astype(err.dtype) * err).sum(-1) / num
    ind = np.nan_to_num(ind, nan=nanvalue, posinf=nanvalue, neginf=nanvalue)
    thres = ind.max() + 1e-6
    ind = (thres - ind) / thres
    return ind + num.astype(np.float64), num, mask, thres


def _local_refine(tn, ext, inlier_mask, lo_num, min_tri_angle, invalid_vis_conf):
    """local_refine_and_compute_error + local_refinement_tri (triangulation.py:959-1017,
    triangulation_helpers.py:648-725).  tn (B,S,2); inlier_mask (B,H,S) -> points (B,lo,3), err (B,lo,S)."""
    B, S, _ = tn.shape
    num = inlier_mask.sum(-1)
    order = stable_argsort_desc(num)[:, :lo_num]
    lo_mask = np.take_along_axis(inlier_mask, order[:, :, None], axis=1)        # (B,lo,S)
    cams = np.broadcast_to(ext[None], (B, S, 3, 4))
    pts = np.zeros((B, lo_num, 3))
    err = np.zeros((B, lo_num, S))
    for j in range(lo_num):
        m = lo_mask[:, j]
        p = np.where(m[..., None], tn, 0.0)
        X, ang, inv_che = dlt_batched(cams, p, mask=m.astype(np.float64), compute_tri_angle=True,
                                      check_cheirality=True)
        with np.errstate(invalid="ignore"):
            tri_ok = (ang >= min_tri_angle).any(-1)
        invalid = (~tri_ok) | inv_che
        with np.errstate(invalid="ignore"):
            e = angular_error(tn.transpose(1, 0, 2), X[None], ext)[0].T               # (B,S)
        e = np.nan_to_num(e, nan=100 * PI, posinf=100 * PI, neginf=100 * PI)
        e = e + np.where(invalid[:, None], PI, 0.0)
        e = e + np.where(invalid_vis_conf, PI, 0.0)
        pts[:, j] = X
        err[:, j] = e
    return pts, err


def triangulate_tracks_chunk(extrinsics, tracks_normalized, pairs, lo_num=50, max_angular_error=2,
                             min_tri_angle=1.5, track_vis=None, track_score=None):
    """triangulate_tracks_single_chunk (triangulation.py:776-956) given the already sampled
    hypothesis pair list `pairs` (H,2) -- the reference draws it with torch.randperm on the
    host (triangulation.py:804-813); the caller reproduces that draw.
    ext (S,3,4), tracks_normalized (S,N,2) -> points (N,3), inlier_num (N), inlier_mask (N,S)."""
    max_rad = max_angular_error * (PI / 180)
    tn = np.asarray(tracks_normalized, dtype=np.float64).transpose(1, 0, 2)   # (B,S,2)
    B, S, _ = tn.shape
    H = len(pairs)
    lo_num = lo_num if H >= lo_num else H
    pr = tn[:, pairs].reshape(B * H, 2, 2)
    cr = np.broadcast_to(extrinsics[pairs][None], (B, H, 2, 3, 4)).reshape(B * H, 2, 3, 4)
    X, ang, inv_che = dlt_batched(cr, pr, compute_tri_angle=True, check_cheirality=True)
    X = X.reshape(B, H, 3)
    with np.errstate(invalid="ignore"):
        invalid = (~(ang >= min_tri_angle).any(-1)).reshape(B, H) | inv_che.reshape(B, H)
        err = angular_error(tn.transpose(1, 0, 2), X.transpose(1, 0, 2), extrinsics).transpose(2, 0, 1)
    err = err + np.where(invalid[:, :, None], PI, 0.0)
    if track_score is not None:
        ivc = (track_vis <= 0.05) | (track_score <= 0.5)
    else:
        ivc = track_vis <= 0.05
    ivc = ivc.T                                                                 # (B,S)
    err = err + np.where(ivc[:, None, :], PI, 0.0)
    with np.errstate(invalid="ignore"):
        inl = err <= max_rad
    lo_pts, lo_err = _local_refine(tn, extrinsics, inl, lo_num, min_tri_angle, ivc)
    lo2 = min(10, lo_num)
    lo_pts2, lo_err2 = _local_refine(tn, extrinsics, lo_err <= max_rad, lo2, min_tri_angle, ivc)
    all_pts = np.concatenate([X, lo_pts, lo_pts2], 1)
    all_err = np.concatenate([err, lo_err, lo_err2], 1)
    ind, num, mask, _ = residual_indicator(all_err, max_rad, nanvalue=2 * PI)
    best = np.argmax(ind, axis=1)
    ar = np.arange(B)
    return all_pts[ar, best], num[ar, best], mask[ar, best]
