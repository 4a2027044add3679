"""ORACLE (test infrastructure, never shipped, never on the product path): absolute pose from 2D-3D matches by
P3P RANSAC -- the role ``pycolmap.absolute_pose_estimation`` plays for the reference
(vggsfm/utils/triangulation.py:324-326,400-432 fallback of ``refine_pose``; vggsfm/runners/video_runner.py:987-998
``align_next_window(use_pnp=True)``).

PARITY UNPINNED.  pycolmap 3.10 is a third-party native dependency that is absent here, its
RANSAC draws from COLMAP's own RNG (not reproducible from Python) and it interleaves an EPnP local optimisation; the
reference holds no golden vectors for this call.  What is restated here is the published behaviour of
``EstimateAbsolutePose`` (COLMAP 3.10 src/colmap/estimators/pose.cc): minimal P3P samples, squared reprojection error
on the normalised image plane (points behind the camera never inliers), threshold = max_error / focal, support =
(most inliers, then smallest inlier residual sum), optional focal-length sampling (``estimate_focal_length``: 30
factors 0.2 + 4.8 (i/30)^2), and -- done by the callers -- a final ``RefineAbsolutePose`` on the RANSAC inliers.
Documented deviations of the device path (and of this oracle, which mirrors it operation by operation so that the
two can be compared bit-for-bit on the same samples): a fixed number of hypotheses drawn by the caller instead of the
adaptive trial count, and no EPnP step inside the loop (the non-linear refinement on the inliers follows anyway).

The minimal solver: ratios u = d2/d1, v = d3/d1 of the three depths; the two distance-ratio equations are quadratics
in u whose coefficients are polynomials in v; their resultant is a quartic in v (Grunert / Fischler-Bolles form);
u follows linearly from the common root; the pose from the two triangles' orthonormal frames.
Only ``tests/`` may import this module.
"""
import numpy as np

NEWTON_ITERS = 80        # resolvent cubic: Newton from the Cauchy bound (monotone towards the largest real root)
FOCAL_SAMPLES, FOCAL_MIN, FOCAL_MAX = 30, 0.2, 5.0      # COLMAP AbsolutePoseEstimationOptions defaults


def focal_length_factors(estimate_focal_length):
    """COLMAP EstimateAbsolutePose: quadratic spacing of the focal length factors."""
    if not estimate_focal_length:
        return np.ones(1)
    i = np.arange(FOCAL_SAMPLES, dtype=np.float64) * (1.0 / FOCAL_SAMPLES)
    return FOCAL_MIN + (FOCAL_MAX - FOCAL_MIN) * i * i


def solve_quartic(k4, k3, k2, k1, k0):
    """Real roots of k4 v^4 + ... + k0 (arrays of equal shape).  Returns (roots (...,4), valid (...,4)).
    Ferrari: depressed quartic y^4 + p y^2 + q y + r, largest real root m of the resolvent cubic
    m^3 + p m^2 + (p^2/4 - r) m - q^2/8, two quadratics; two Newton steps on the original quartic per root."""
    with np.errstate(all="ignore"):
        b, c, d, e = k3 / k4, k2 / k4, k1 / k4, k0 / k4
        b2 = b * b
        p = c - 0.375 * b2
        q = d - 0.5 * b * c + 0.125 * b2 * b
        r = e - 0.25 * b * d + 0.0625 * b2 * c - (3.0 / 256.0) * b2 * b2
        c1 = 0.25 * p * p - r
        c0 = -0.125 * q * q
        m = 1.0 + np.maximum(np.abs(p), np.maximum(np.abs(c1), np.abs(c0)))
        for _ in range(NEWTON_ITERS):
            g = ((m + p) * m + c1) * m + c0
            dg = (3.0 * m + 2.0 * p) * m + c1
            step = np.where(dg != 0.0, g / dg, 0.0)
            m = m - step
        scale = 1.0 + np.abs(p) + np.abs(m)
        roots = np.zeros(k4.shape + (4,))
        valid = np.zeros(k4.shape + (4,), bool)
        biq = ~(m > 1e-14 * scale)                        # q ~ 0: biquadratic
        # general case
        s = np.sqrt(2.0 * np.where(biq, 1.0, m))
        qs = q / s
        for j, sg in enumerate((1.0, -1.0)):
            disc = -2.0 * p - 2.0 * m - sg * 2.0 * qs
            disc = np.where((disc < 0.0) & (disc > -1e-10 * scale), 0.0, disc)
            ok = disc >= 0.0
            sq = np.sqrt(np.where(ok, disc, 0.0))
            roots[..., 2 * j] = 0.5 * (sg * s + sq)
            roots[..., 2 * j + 1] = 0.5 * (sg * s - sq)
            valid[..., 2 * j] = ok
            valid[..., 2 * j + 1] = ok
        # biquadratic case: y^2 = (-p +- sqrt(p^2 - 4 r)) / 2
        dq = p * p - 4.0 * r
        dq = np.where((dq < 0.0) & (dq > -1e-10 * scale * scale), 0.0, dq)
        okq = dq >= 0.0
        sq = np.sqrt(np.where(okq, dq, 0.0))
        for j, sg in enumerate((1.0, -1.0)):
            y2 = 0.5 * (-p + sg * sq)
            oky = okq & (y2 >= 0.0)
            y = np.sqrt(np.where(oky, y2, 0.0))
            roots[..., 2 * j] = np.where(biq, y, roots[..., 2 * j])
            roots[..., 2 * j + 1] = np.where(biq, -y, roots[..., 2 * j + 1])
            valid[..., 2 * j] = np.where(biq, oky, valid[..., 2 * j])
            valid[..., 2 * j + 1] = np.where(biq, oky, valid[..., 2 * j + 1])
        v = roots - 0.25 * b[..., None]
        K4, K3, K2, K1, K0 = (x[..., None] for x in (k4, k3, k2, k1, k0))
        for _ in range(2):
            f = (((K4 * v + K3) * v + K2) * v + K1) * v + K0
            df = ((4.0 * K4 * v + 3.0 * K3) * v + 2.0 * K2) * v + K1
            v = v - np.where(df != 0.0, f / df, 0.0)
        lead_ok = np.abs(k4) > 1e-14 * (np.abs(k4) + np.abs(k3) + np.abs(k2) + np.abs(k1) + np.abs(k0))
        valid &= lead_ok[..., None] & np.isfinite(v)
    return v, valid


def _frame(P1, P2, P3):
    """Orthonormal frame of a triangle: e1 along P1->P2, e3 normal, e2 = e3 x e1.  Returns (E (...,3,3) columns, ok)."""
    with np.errstate(all="ignore"):
        dot = lambda a, b: (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]
        cross = lambda a, b: np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                                       a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1)
        a = P2 - P1
        e1 = a / np.sqrt(dot(a, a))[..., None]
        n = cross(e1, P3 - P1)
        nn = np.sqrt(dot(n, n))
        e3 = n / nn[..., None]
        e2 = cross(e3, e1)
    return np.stack([e1, e2, e3], -1), nn


def p3p_solve(x, X):
    """x (H,3,2) normalised image points, X (H,3,3) world points -> poses (H,4,3,4) [R|t] (cam from world), valid (H,4)."""
    with np.errstate(all="ignore"):
        H = x.shape[0]
        bear = np.concatenate([x, np.ones((H, 3, 1))], -1)
        bear = bear / np.sqrt((bear[..., 0] * bear[..., 0] + bear[..., 1] * bear[..., 1]) + 1.0)[..., None]
        b1, b2, b3 = bear[:, 0], bear[:, 1], bear[:, 2]
        X1, X2, X3 = X[:, 0], X[:, 1], X[:, 2]
        dot = lambda a, b: (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]
        a12, a13, a23 = dot(X1 - X2, X1 - X2), dot(X1 - X3, X1 - X3), dot(X2 - X3, X2 - X3)
        c12, c13, c23 = dot(b1, b2), dot(b1, b3), dot(b2, b3)
        ok0 = (a12 > 1e-20) & (a13 > 1e-20) & (a23 > 1e-20)
        inv = 1.0 / a12
        A13, A23 = a13 * inv, a23 * inv
        A1 = A13
        B1 = -2.0 * A13 * c12
        c11, c10 = 2.0 * c13, A13 - 1.0                 # C1 = -v^2 + c11 v + c10
        A2 = A23 - 1.0
        b21, b20 = 2.0 * c23, -2.0 * A23 * c12          # B2 = b21 v + b20 ; C2 = -v^2 + A23
        p2, p1, p0 = A2 - A1, -A2 * c11, A1 * A23 - A2 * c10
        q1, q0 = A1 * b21, A1 * b20 - A2 * B1
        t3, t2, t1, t0 = b21, -B1 - (b21 * c11 - b20), -(b21 * c10 + b20 * c11), B1 * A23 - b20 * c10
        k4 = p2 * p2 - q1 * t3
        k3 = 2.0 * p2 * p1 - (q1 * t2 + q0 * t3)
        k2 = 2.0 * p2 * p0 + p1 * p1 - (q1 * t1 + q0 * t2)
        k1 = 2.0 * p1 * p0 - (q1 * t0 + q0 * t1)
        k0 = p0 * p0 - q0 * t0
        v, okv = solve_quartic(k4, k3, k2, k1, k0)       # (H,4)
        Pv = (p2[:, None] * v + p1[:, None]) * v + p0[:, None]
        Qv = q1[:, None] * v + q0[:, None]
        okq = np.abs(Qv) > 1e-12 * (np.abs(q1) + np.abs(q0))[:, None]
        u = -Pv / np.where(okq, Qv, 1.0)
        den = 1.0 + u * u - 2.0 * u * c12[:, None]
        # (u, v) must satisfy both quadratics (drops spurious roots of the resultant / of a clamped discriminant)
        C1v = (c11[:, None] - v) * v + c10[:, None]
        C2v = A23[:, None] - v * v
        B2v = b21[:, None] * v + b20[:, None]
        r1 = (A1[:, None] * u + B1[:, None]) * u + C1v
        r2 = (A2[:, None] * u + B2v) * u + C2v
        okr = (np.abs(r1) + np.abs(r2)) <= 1e-7 * (1.0 + u * u + v * v)
        ok = okv & okq & okr & (v > 0.0) & (u > 0.0) & (den > 0.0) & ok0[:, None]
        d1 = np.sqrt(a12[:, None] / np.where(ok, den, 1.0))
        d2, d3 = u * d1, v * d1
        Y1 = d1[..., None] * b1[:, None, :]
        Y2 = d2[..., None] * b2[:, None, :]
        Y3 = d3[..., None] * b3[:, None, :]
        E, nE = _frame(X1, X2, X3)                       # (H,3,3)
        C, nC = _frame(Y1, Y2, Y3)                       # (H,4,3,3)
        # R = C E^T and t = Y1 - R X1, written out term by term (fixed summation order, no BLAS / FMA)
        R = np.zeros((H, 4, 3, 3))
        for i in range(3):
            for j in range(3):
                R[..., i, j] = (C[..., i, 0] * E[:, None, j, 0] + C[..., i, 1] * E[:, None, j, 1]) + C[..., i, 2] * E[:, None, j, 2]
        t = np.zeros((H, 4, 3))
        for i in range(3):
            t[..., i] = Y1[..., i] - ((R[..., i, 0] * X1[:, None, 0] + R[..., i, 1] * X1[:, None, 1]) + R[..., i, 2] * X1[:, None, 2])
        ok &= (nE > 1e-12)[:, None] & (nC > 1e-12)
        poses = np.concatenate([R, t[..., None]], -1)
        ok &= np.isfinite(poses).all((-1, -2))
        poses = np.where(ok[..., None, None], poses, 0.0)
    return poses, ok


def score_poses(poses, x, X, mask, thr_sq):
    """poses (M,3,4); x (N,2); X (N,3); mask (N,) bool -> (count (M,), residual_sum (M,), inlier (M,N))."""
    with np.errstate(all="ignore"):
        p = np.zeros((poses.shape[0], X.shape[0], 3))
        for i in range(3):                               # ((R0 X0 + R1 X1) + R2 X2) + t, fixed order
            p[..., i] = ((poses[:, None, i, 0] * X[None, :, 0] + poses[:, None, i, 1] * X[None, :, 1])
                         + poses[:, None, i, 2] * X[None, :, 2]) + poses[:, None, i, 3]
        z = p[..., 2]
        front = z > 1e-12
        zs = np.where(front, z, 1.0)
        e = (p[..., 0] / zs - x[None, :, 0]) ** 2 + (p[..., 1] / zs - x[None, :, 1]) ** 2
        inl = front & (e <= thr_sq) & mask[None]
    return inl.sum(1), np.where(inl, e, 0.0).sum(1), inl


def absolute_pose_ransac(x, X, mask, samples, thr_sq):
    """One (virtual) frame.  x (N,2) normalised points, X (N,3), mask (N,) candidates, samples (H,3) indices.
    Returns dict(pose (3,4), num_inliers, residual_sum, inliers (N,) bool, best = flat index hypothesis*4 + solution)
    -- num_inliers 0 when no hypothesis is valid."""
    poses, ok = p3p_solve(x[samples], X[samples])
    Hn = samples.shape[0]
    flat = poses.reshape(Hn * 4, 3, 4)
    cnt, rs, inl = score_poses(flat, x, X, mask, thr_sq)
    cnt = np.where(ok.reshape(-1), cnt, -1)
    best = -1
    for i in range(Hn * 4):                               # most inliers, then smallest residual sum, then lowest index
        if cnt[i] < 0:
            continue
        if best < 0 or cnt[i] > cnt[best] or (cnt[i] == cnt[best] and rs[i] < rs[best]):
            best = i
    if best < 0 or cnt[best] <= 0:
        return dict(pose=np.zeros((3, 4)), num_inliers=0, residual_sum=0.0, inliers=np.zeros(len(x), bool), best=-1)
    return dict(pose=flat[best], num_inliers=int(cnt[best]), residual_sum=float(rs[best]), inliers=inl[best], best=best)
