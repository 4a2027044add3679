"""ORACLE (test infrastructure, never shipped, never on the product path): the two-view stage in front of the hot path,
``estimate_fundamental`` = 7-point RANSAC + two rounds of 8-point local optimisation
(vggsfm/two_view_geo/fundamental.py:43-183, its helpers in two_view_geo/utils.py:63-298, caller
estimate_preliminary.py:103-152 which turns the winner's inlier mask into ``fmat_inlier_mask`` for the Triangulator).

PARITY PARTLY PINNED.  The reference's implementation imports kornia (not installed here), computes its SVDs and
Sampson residuals in float32 (``autocast``), draws its 7-point samples from numpy's global RNG, and by default
(cfg.use_poselib) is not even used: poselib's LO-RANSAC runs on the CPU instead.  tests/test_oracle_fundamental_vs_reference.py
runs the reference's OWN functions (with kornia's five trivial helpers restated) against this file: the Sampson
distance, the 8-point fit with its masked normalisation, the gathering of the inlier sets for the local optimisation
and the winner selection agree (2e-8 on the matrices in float64; the reference's float32 residuals to 1e-4); so does
the 7-point stage -- null-space pencil, cubic coefficients, scaling, de-normalisation -- run in the reference's float32
with kornia's normalize_points restated and its solve_cubic replaced by numpy.roots (median 1e-4 on unit-norm matrices).
NOT pinned: kornia's own cubic solver and the RNG.
What is restated is the algorithm as the reference states
it, in float64:

    samples (H,7) shared by all pairs                                   utils.py:39-60 (drawn by the caller)
    7-point solver on normalised points, <= 3 matrices per sample       fundamental.py:339-469
    squared Sampson distance, invalid matches = 1e6                     utils.py:90-172, fundamental.py:96-104
    inliers <= max_error^2; hypotheses sorted by inlier count (stable)  fundamental.py:106-112
    8-point fit on the inliers of the best lo_num hypotheses            fundamental.py:118-120, 261-334, utils.py:175-298
    a second round on the lo_num // 2 best refits                       fundamental.py:126-152
    winner = most inliers, then the smallest mean inlier residual       fundamental.py:162-176, utils.py:63-87

with these documented differences in HOW (never in what is selected): the null space of the 7x9 system comes from
Gauss-Jordan elimination with complete pivoting instead of an SVD (any basis spans the same pencil), the pencil is
parametrised F = F1 + lambda F2, the cubic is solved by bisection + deflation (no transcendental functions, so that this
file and csrc/fundamental.hip agree bit for bit), symmetric eigenproblems (9x9 normal matrix, 3x3 rank-2 projection)
use cyclic Jacobi with a fixed number of sweeps, and every sum runs in the fixed order of the kernels' reductions.
Only ``tests/`` may import this module.
"""
import numpy as np

BISECTIONS = 110
SWEEPS9, SWEEPS3 = 10, 8
BIG = 1e6                         # residual of an invalid match (fundamental.py:102-104)


# ----------------------------------------------------------------------------------------------- fixed-order sums
def tree_sum(acc):
    """acc (..., L) with L a power of two: halving tree (t, t + L/2), ..., the order of the kernels' reductions."""
    L = acc.shape[-1]
    while L > 1:
        L //= 2
        acc = acc[..., :L] + acc[..., L:2 * L]
    return acc[..., 0]


def seq_sum(values):
    """values (..., k), small k: plain left-to-right sum (what a single thread does)."""
    acc = values[..., 0].copy()
    for i in range(1, values.shape[-1]):
        acc = acc + values[..., i]
    return acc


def strided_sum(values, lanes):
    """values (..., N): lane t adds elements t, t + lanes, ... in order, then the lanes are tree-summed."""
    N = values.shape[-1]
    acc = np.zeros(values.shape[:-1] + (lanes,), dtype=values.dtype)
    for n0 in range(0, N, lanes):
        chunk = values[..., n0:n0 + lanes]
        acc[..., :chunk.shape[-1]] += chunk
    return tree_sum(acc)


# ----------------------------------------------------------------------------------------------- small solvers
def cubic_real_roots(c3, c2, c1, c0):
    """Real roots of c3 x^3 + c2 x^2 + c1 x + c0 (arrays).  Returns (roots (...,3), valid (...,3)).
    One real root by bisection over the Cauchy bound, the other two from the deflated quadratic; one guarded Newton step."""
    with np.errstate(all="ignore"):
        mx = np.abs(c3) + np.abs(c2) + np.abs(c1) + np.abs(c0)
        is_cubic = np.abs(c3) > 1e-12 * mx
        a, b, c = c2 / c3, c1 / c3, c0 / c3
        # a real root of the monic cubic by bisection over the Cauchy bound (g(-B) < 0 < g(B); only + and *)
        B = 1.0 + np.maximum(np.abs(a), np.maximum(np.abs(b), np.abs(c)))
        lo, hi = -B, B
        for _ in range(BISECTIONS):
            mid = 0.5 * (lo + hi)
            g = ((mid + a) * mid + b) * mid + c
            pos = g > 0.0
            hi = np.where(pos, mid, hi)
            lo = np.where(pos, lo, mid)
        x = 0.5 * (lo + hi)
        # deflate: x^3 + a x^2 + b x + c = (x - r)(x^2 + (a + r) x + (b + (a + r) r))
        # (forward deflation when r is the small root, backward -- from the constant term -- when it is the large one)
        back = np.abs(x * x * x) > np.abs(c)
        qc_b = -c / np.where(back, x, 1.0)
        qb_b = (qc_b - b) / np.where(back, x, 1.0)
        qb_f = a + x
        qc_f = b + qb_f * x
        qb = np.where(back, qb_b, qb_f)
        qc = np.where(back, qc_b, qc_f)
        disc = qb * qb - 4.0 * qc
        okq = disc >= 0.0
        sq = np.sqrt(np.where(okq, disc, 0.0))
        r1 = 0.5 * (-qb + sq)
        r2 = 0.5 * (-qb - sq)
        roots = np.stack([x, r1, r2], -1)
        valid = np.stack([is_cubic, is_cubic & okq, is_cubic & okq], -1)
        # quadratic / linear pencils (c3 ~ 0)
        is_quad = ~is_cubic & (np.abs(c2) > 1e-12 * mx)
        dq = c1 * c1 - 4.0 * c2 * c0
        okd = dq >= 0.0
        sqd = np.sqrt(np.where(okd, dq, 0.0))
        q1 = (-c1 + sqd) / (2.0 * c2)
        q2 = (-c1 - sqd) / (2.0 * c2)
        is_lin = ~is_cubic & ~is_quad & (np.abs(c1) > 1e-12 * mx)
        l1 = -c0 / c1
        roots[..., 0] = np.where(is_cubic, roots[..., 0], np.where(is_quad, q1, l1))
        roots[..., 1] = np.where(is_cubic, roots[..., 1], q2)
        valid[..., 0] = np.where(is_cubic, valid[..., 0], (is_quad & okd) | is_lin)
        valid[..., 1] = np.where(is_cubic, valid[..., 1], is_quad & okd)
        valid[..., 2] = np.where(is_cubic, valid[..., 2], False)
        C3, C2, C1, C0 = (v[..., None] for v in (c3, c2, c1, c0))
        f = ((C3 * roots + C2) * roots + C1) * roots + C0
        df = (3.0 * C3 * roots + 2.0 * C2) * roots + C1
        cand = roots - np.where(df != 0.0, f / df, 0.0)          # one Newton step, kept only if it helps
        fc = ((C3 * cand + C2) * cand + C1) * cand + C0
        roots = np.where(np.abs(fc) < np.abs(f), cand, roots)
        valid &= np.isfinite(roots)
    return roots, valid


def jacobi_eigh(A, sweeps):
    """Cyclic Jacobi on symmetric A (M,n,n), fixed sweeps.  Returns (diag (M,n), V (M,n,n)) with A = V diag V^T.
    Rotation (p,q): tau = (a_qq - a_pp) / (2 a_pq), t = sign(tau) / (|tau| + sqrt(1 + tau^2)), c = 1/sqrt(1+t^2),
    s = t c; columns p,q of A then rows p,q of A then columns p,q of V: new_p = c p - s q, new_q = s p + c q."""
    A = A.copy()
    M, n, _ = A.shape
    V = np.broadcast_to(np.eye(n), (M, n, n)).copy()
    with np.errstate(all="ignore"):
        for _ in range(sweeps):
            for p in range(n - 1):
                for q in range(p + 1, n):
                    apq = A[:, p, q]
                    rot = apq != 0.0
                    tau = (A[:, q, q] - A[:, p, p]) / (2.0 * np.where(rot, apq, 1.0))
                    t = np.where(tau >= 0.0, 1.0, -1.0) / (np.abs(tau) + np.sqrt(1.0 + tau * tau))
                    c = 1.0 / np.sqrt(1.0 + t * t)
                    s = t * c
                    c = np.where(rot, c, 1.0)
                    s = np.where(rot, s, 0.0)
                    cp, cq = A[:, :, p].copy(), A[:, :, q].copy()
                    A[:, :, p] = c[:, None] * cp - s[:, None] * cq
                    A[:, :, q] = s[:, None] * cp + c[:, None] * cq
                    rp, rq = A[:, p, :].copy(), A[:, q, :].copy()
                    A[:, p, :] = c[:, None] * rp - s[:, None] * rq
                    A[:, q, :] = s[:, None] * rp + c[:, None] * rq
                    vp, vq = V[:, :, p].copy(), V[:, :, q].copy()
                    V[:, :, p] = c[:, None] * vp - s[:, None] * vq
                    V[:, :, q] = s[:, None] * vp + c[:, None] * vq
    return np.stack([A[:, i, i] for i in range(n)], -1), V


def _smallest_eigvec(A, sweeps):
    d, V = jacobi_eigh(A, sweeps)
    j = np.argmin(d, axis=1)                                 # first minimum
    return np.take_along_axis(V, j[:, None, None], axis=2)[..., 0]


def null_space_7x9(A):
    """A (M,7,9) -> two null vectors (M,9), (M,9) by Gauss-Jordan with complete pivoting (first maximum in row-major
    order of the remaining block), and ok (M,) = every pivot non-zero."""
    A = A.copy()
    M = A.shape[0]
    perm = np.broadcast_to(np.arange(9), (M, 9)).copy()
    ok = np.ones(M, bool)
    ar = np.arange(M)
    with np.errstate(all="ignore"):
        for k in range(7):
            sub = np.abs(A[:, k:, k:]).reshape(M, -1)
            flat = np.argmax(sub, axis=1)
            r, c = k + flat // (9 - k), k + flat % (9 - k)
            tmp = A[ar, k, :].copy(); A[ar, k, :] = A[ar, r, :]; A[ar, r, :] = tmp       # noqa: E702
            tmp = A[ar, :, k].copy(); A[ar, :, k] = A[ar, :, c]; A[ar, :, c] = tmp       # noqa: E702
            tp = perm[ar, k].copy(); perm[ar, k] = perm[ar, c]; perm[ar, c] = tp         # noqa: E702
            piv = A[:, k, k]
            ok &= piv != 0.0
            A[:, k, :] = A[:, k, :] / np.where(piv != 0.0, piv, 1.0)[:, None]
            for i in range(7):
                if i != k:
                    f = A[:, i, k].copy()
                    A[:, i, :] = A[:, i, :] - f[:, None] * A[:, k, :]
    out = []
    for j in (7, 8):
        x = np.zeros((M, 9))
        x[ar, perm[:, j]] = 1.0
        for i in range(7):
            x[ar, perm[:, i]] = -A[:, i, j]
        out.append(x)
    return out[0], out[1], ok


def det3(F):
    return (F[..., 0, 0] * (F[..., 1, 1] * F[..., 2, 2] - F[..., 1, 2] * F[..., 2, 1])
            - F[..., 0, 1] * (F[..., 1, 0] * F[..., 2, 2] - F[..., 1, 2] * F[..., 2, 0])) \
        + F[..., 0, 2] * (F[..., 1, 0] * F[..., 2, 1] - F[..., 1, 1] * F[..., 2, 0])


def _mat3(a, b):
    """(...,3,3) @ (...,3,3), terms added left to right."""
    out = np.zeros(np.broadcast_shapes(a.shape, b.shape))
    for i in range(3):
        for j in range(3):
            out[..., i, j] = (a[..., i, 0] * b[..., 0, j] + a[..., i, 1] * b[..., 1, j]) + a[..., i, 2] * b[..., 2, j]
    return out


def _denormalize(Fh, T1, T2):
    """T2^T Fh T1 for T = [[s,0,a],[0,s,b],[0,0,1]] (written with general 3x3 products, fixed order)."""
    return _mat3(np.swapaxes(T2, -1, -2), _mat3(Fh, T1))


def _unit(F):
    n = np.sqrt(seq_sum((F * F).reshape(F.shape[:-2] + (9,))))
    with np.errstate(all="ignore"):
        return F / n[..., None, None], n


def _rows(x1, y1, x2, y2):
    one = np.ones_like(x1)
    return np.stack([x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, one], -1)


# ----------------------------------------------------------------------------------------------- 7-point
def seven_point(p1, p2):
    """p1, p2 (M,7,2) -> F (M,3,3,3) unit Frobenius norm, valid (M,3).  fundamental.py:339-469 (kornia-style
    normalisation of the 7 points: mean, scale = sqrt(2) / (mean distance + 1e-8))."""
    with np.errstate(all="ignore"):
        def norm(p):
            mx = seq_sum(p[..., 0]) / 7.0
            my = seq_sum(p[..., 1]) / 7.0
            dx, dy = p[..., 0] - mx[:, None], p[..., 1] - my[:, None]
            d = np.sqrt(dx * dx + dy * dy)
            s = np.sqrt(2.0) / (seq_sum(d) / 7.0 + 1e-8)
            T = np.zeros(p.shape[:-2] + (3, 3))
            T[..., 0, 0], T[..., 1, 1], T[..., 2, 2] = s, s, 1.0
            T[..., 0, 2], T[..., 1, 2] = -s * mx, -s * my
            return s[:, None] * p[..., 0] + T[..., 0, 2][:, None], s[:, None] * p[..., 1] + T[..., 1, 2][:, None], T
        x1, y1, T1 = norm(p1)
        x2, y2, T2 = norm(p2)
        A = _rows(x1, y1, x2, y2)                              # (M,7,9)
        f1, f2, ok = null_space_7x9(A)
        F1, F2 = f1.reshape(-1, 3, 3), f2.reshape(-1, 3, 3)
        c0, c3 = det3(F1), det3(F2)
        dp, dm = det3(F1 + F2), det3(F1 - F2)
        c2 = 0.5 * (dp + dm) - c0
        c1 = 0.5 * (dp - dm) - c3
        lam, okr = cubic_real_roots(c3, c2, c1, c0)            # (M,3)
        Fh = F1[:, None] + lam[..., None, None] * F2[:, None]
        F = _denormalize(Fh, T1[:, None], T2[:, None])
        F, n = _unit(F)
        valid = okr & ok[:, None] & np.isfinite(F).all((-1, -2)) & (n > 0.0)
        F = np.where(valid[..., None, None], F, 0.0)
    return F, valid


# ----------------------------------------------------------------------------------------------- residuals
def sampson_sq(F, x1, x2):
    """F (K,3,3); x1, x2 (N,2) -> squared Sampson distances (K,N) (utils.py:90-172), NaN -> BIG."""
    with np.errstate(all="ignore"):
        u1, v1, u2, v2 = x1[None, :, 0], x1[None, :, 1], x2[None, :, 0], x2[None, :, 1]
        f = lambda i, j: F[:, i, j][:, None]
        l0 = (f(0, 0) * u1 + f(0, 1) * v1) + f(0, 2)            # F x1
        l1 = (f(1, 0) * u1 + f(1, 1) * v1) + f(1, 2)
        l2 = (f(2, 0) * u1 + f(2, 1) * v1) + f(2, 2)
        m0 = (f(0, 0) * u2 + f(1, 0) * v2) + f(2, 0)            # F^T x2
        m1 = (f(0, 1) * u2 + f(1, 1) * v2) + f(2, 1)
        num = (u2 * l0 + v2 * l1) + l2
        den = (l0 * l0 + l1 * l1) + (m0 * m0 + m1 * m1)
        r = (num * num) / den
    return np.where(np.isfinite(r), r, BIG)


def score(F, fvalid, x1, x2, vmask, thr_sq, lanes=64):
    """-> (count (K,) int64 [-1 for an invalid hypothesis], residual sum of the inliers (K,), inlier (K,N))."""
    r = sampson_sq(F, x1, x2)
    inl = (r <= thr_sq) & vmask[None]
    cnt = inl.sum(1).astype(np.int64)
    rs = strided_sum(np.where(inl, r, 0.0), lanes)
    return np.where(fvalid, cnt, -1), rs, inl


# ----------------------------------------------------------------------------------------------- 8-point
def eight_point(x1, x2, masks, lanes=256):
    """x1, x2 (N,2); masks (L,N) bool -> F (L,3,3) unit norm, valid (L,).  fundamental.py:261-334 with
    normalize_points_masked (utils.py:175-253, eps 1e-8), smallest eigenvector of X^T X, rank 2 enforced."""
    L, N = masks.shape
    mf = masks.astype(np.float64)
    with np.errstate(all="ignore"):
        n = mf.sum(1)

        def norm(p):
            mx = strided_sum(mf * p[None, :, 0], lanes) / (n + 1e-8)
            my = strided_sum(mf * p[None, :, 1], lanes) / (n + 1e-8)
            dx, dy = mf * p[None, :, 0] - mx[:, None], mf * p[None, :, 1] - my[:, None]
            d = np.sqrt(dx * dx + dy * dy) * mf
            s = np.sqrt(2.0) / (strided_sum(d, lanes) / (n + 1e-8) + 1e-8)
            T = np.zeros((L, 3, 3))
            T[:, 0, 0], T[:, 1, 1], T[:, 2, 2] = s, s, 1.0
            T[:, 0, 2], T[:, 1, 2] = -s * mx, -s * my
            return s[:, None] * p[None, :, 0] + T[:, 0, 2][:, None], s[:, None] * p[None, :, 1] + T[:, 1, 2][:, None], T
        a1, b1, T1 = norm(x1)
        a2, b2, T2 = norm(x2)
        X = _rows(a1, b1, a2, b2) * mf[..., None]              # (L,N,9)
        Mx = np.zeros((L, 9, 9))
        for i in range(9):
            for j in range(i, 9):
                Mx[:, i, j] = strided_sum(X[..., i] * X[..., j], lanes)
                Mx[:, j, i] = Mx[:, i, j]
        fvec = _smallest_eigvec(Mx, SWEEPS9)
        Fh = fvec.reshape(L, 3, 3)
        # rank 2: remove the component along the right singular vector of the smallest singular value
        G = _mat3(np.swapaxes(Fh, -1, -2), Fh)
        v3 = _smallest_eigvec(G, SWEEPS3)
        Fv = np.stack([(Fh[:, i, 0] * v3[:, 0] + Fh[:, i, 1] * v3[:, 1]) + Fh[:, i, 2] * v3[:, 2] for i in range(3)], -1)
        Fh = Fh - Fv[:, :, None] * v3[:, None, :]
        F = _denormalize(Fh, T1, T2)
        F, nrm = _unit(F)
        valid = (n >= 8) & np.isfinite(F).all((-1, -2)) & (nrm > 0.0)
        F = np.where(valid[:, None, None], F, 0.0)
    return F, valid


# ----------------------------------------------------------------------------------------------- driver
def _order(cnt):
    """torch.sort(descending=True) with the stable tie order (lower index first)."""
    return np.argsort(-cnt, kind="stable")


def _best(cnt, rs):
    """Most inliers, then the smallest mean inlier residual, then the lowest index (utils.py:63-87)."""
    with np.errstate(all="ignore"):
        mean = np.where(cnt > 0, rs / np.maximum(cnt, 1), BIG)
    top = cnt.max()
    cand = np.where(cnt == top, mean, np.inf)
    return int(np.argmin(cand))


def estimate_fundamental_pair(x1, x2, vmask, samples, max_error, lo_num=300, second_refine=True):
    """One image pair.  x1, x2 (N,2) pixels; vmask (N,) bool; samples (H,7).
    Returns dict(fmat (3,3) scaled to F[2,2] = 1 when possible, inlier_num, inlier_mask (N,), residuals (N,), best)."""
    thr = max_error * max_error
    F7, v7 = seven_point(x1[samples], x2[samples])
    Fa, va = F7.reshape(-1, 3, 3), v7.reshape(-1)
    cnt, rs, inl = score(Fa, va, x1, x2, vmask, thr)
    lo = min(lo_num, len(cnt))
    top = _order(cnt)[:lo]
    F8, v8 = eight_point(x1, x2, inl[top] & (cnt[top] >= 0)[:, None])
    c8, r8, i8 = score(F8, v8, x1, x2, vmask, thr)
    allF, allc, allr, alli = [Fa, F8], [cnt, c8], [rs, r8], [inl, i8]
    if second_refine:
        lo2 = min(lo_num // 2, len(c8))
        # fundamental.py:126-152: the second round ranks and refits on residuals_lo BEFORE valid_mask is applied
        c8u, _, i8u = score(F8, v8, x1, x2, np.ones_like(vmask), thr)
        top2 = _order(c8u)[:lo2]
        F9, v9 = eight_point(x1, x2, i8u[top2] & (c8u[top2] >= 0)[:, None])
        c9, r9, i9 = score(F9, v9, x1, x2, vmask, thr)
        allF.append(F9); allc.append(c9); allr.append(r9); alli.append(i9)      # noqa: E702
    Fall, call, rall, iall = np.concatenate(allF), np.concatenate(allc), np.concatenate(allr), np.concatenate(alli)
    b = _best(call, rall)
    F = Fall[b]
    res = sampson_sq(F[None], x1, x2)[0]
    res = np.where(vmask, res, BIG)
    with np.errstate(all="ignore"):
        Fn = F / F[2, 2] if abs(F[2, 2]) > 1e-8 else F          # normalize_transformation
    return dict(fmat=Fn, fmat_unit=F, inlier_num=int(max(call[b], 0)), inlier_mask=iall[b] & (call[b] >= 0), residuals=res,
                best=b, counts=call)
