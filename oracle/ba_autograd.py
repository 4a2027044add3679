"""ORACLE #2 -- TEST INFRASTRUCTURE ONLY (same rules as oracle/ba_oracle.c: only tests/ may import it).

A second, independently derived statement of the bundle adjustment the reference delegates to
pycolmap.bundle_adjustment / pose_refinement (vggsfm/utils/triangulation.py:213,387,590,1050,1142), written to
pin oracle/ba_oracle.c ("What's missing" 1).  PARITY UNPINNED vs pycolmap itself (absent here); what
this file adds is that the C restatement's mathematics is checked against a derivation that shares none of it:

* NO analytic Jacobian.  The residual of one observation is a differentiable torch function of a tangent
  perturbation (dq, dt, df, dk, dX) of its parameter blocks, `ImgFromCam(intr + d, R(exp(dq) * q) (X + dX) + t + dt) - uv`
  (COLMAP 3.10 cost_functions.h ReprojErrorCostFunction; Ceres EigenQuaternionManifold::Plus), and the 2 x 11
  Jacobian block comes from `torch.func.jacrev` at zero, `vmap`ped over the observations.
* NO Schur complement.  The damped normal equations `(Js^T Js + D^2) y = Js^T r` are formed DENSELY over all
  columns (cameras, intrinsics, points) and handed to numpy.linalg -- algebraically the step Ceres' DENSE_SCHUR /
  SPARSE_SCHUR solvers compute, by another route.
* NO hand-written derivative of the loss: rho'(s) is the autograd derivative of rho(s).
* The trust-region control flow is restated from Ceres 2.x `trust_region_minimizer.cc` /
  `levenberg_marquardt_strategy.cc`, in its own words.

Sizes: the dense Jacobian is (2 O) x (6 C + kd NI + 3 P) -- small problems only (seconds up to ~2000 observations).
"""
import numpy as np
import torch

F64 = torch.float64


# ------------------------------------------------------------------ the model, differentiable
def _exp_quat(d):
    """Ceres EigenQuaternionManifold::Plus uses q(d) = [sin|d|/|d| d, cos|d|] (x,y,z,w).  Around d = 0 (the only
    place the Jacobian is taken) sin(n)/n = 1 - n^2/6 + ..., cos n = 1 - n^2/2 + ...; the truncation is exact to
    first order, which is all `jacrev` at zero sees, and has no 0/0."""
    n2 = (d * d).sum()
    return torch.cat([d * (1.0 - n2 / 6.0), (1.0 - n2 / 2.0).reshape(1)])


def _quat_mul(a, b):
    """Hamilton product of (x,y,z,w) quaternions, a * b."""
    ax, ay, az, aw = a[0], a[1], a[2], a[3]
    bx, by, bz, bw = b[0], b[1], b[2], b[3]
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw,
                        aw * bw - ax * bx - ay * by - az * bz])


def _rotate(q, v):
    """v' = q v q^-1 for a unit quaternion, written as the rotation matrix of q applied to v."""
    x, y, z, w = q[0], q[1], q[2], q[3]
    R = torch.stack([torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)]),
                     torch.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)]),
                     torch.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)])])
    return R @ v


def _residual(dq, dt, dfk, dX, q, t, intr, X, uv, radial):
    """One observation.  intr = (f, cx, cy, k); dfk = (df, dk)."""
    Y = _rotate(_quat_mul(_exp_quat(dq), q), X + dX) + t + dt
    u, v = Y[0] / Y[2], Y[1] / Y[2]
    f = intr[0] + dfk[0]
    if radial:
        k = intr[3] + dfk[1]
        d = 1.0 + k * (u * u + v * v)
    else:
        d = 1.0 + 0.0 * dfk[1]
    return torch.stack([f * (d * u) + intr[1] - uv[0], f * (d * v) + intr[2] - uv[1]])


def _rho(loss, a, s):
    b = a * a
    if loss == 1:                               # Cauchy
        return b * torch.log1p(s / b)
    if loss == 2:                               # Huber
        return torch.where(s > b, 2 * a * torch.sqrt(torch.clamp(s, min=1e-300)) - b, s)
    if loss == 3:                               # SoftL1
        return 2 * b * (torch.sqrt(1 + s / b) - 1)
    return s


def _quat_plus_np(q, d):
    n = np.linalg.norm(d)
    if n == 0.0:
        return q.copy()
    e = np.concatenate([np.sin(n) / n * d, [np.cos(n)]])
    ax, ay, az, aw = e
    bx, by, bz, bw = q
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


class _Problem:
    def __init__(self, cam_intr, row_ptr, obs_cam, obs_uv, camera_model, refine_focal, refine_extra, loss, loss_scale,
                 cam_const, intr_const, pt_const, C, NI, P):
        self.C, self.NI, self.P = C, NI, P
        self.cam_intr = np.asarray(cam_intr, np.int64)
        self.obs_cam = np.asarray(obs_cam, np.int64)
        self.obs_pt = np.repeat(np.arange(P), np.diff(np.asarray(row_ptr, np.int64)))
        self.uv = torch.from_numpy(np.asarray(obs_uv, np.float64).reshape(-1, 2))
        self.radial = camera_model == 1
        self.loss, self.loss_scale = loss, float(loss_scale)
        # tangent layout: [6 per camera (dq, dt)] [kd per intrinsics block] [3 per point]
        self.kcols = [j for j, on in ((0, refine_focal), (1, refine_extra and self.radial)) if on]
        self.kd = len(self.kcols)
        self.n_red = 6 * C + self.kd * NI
        self.n = self.n_red + 3 * P
        act = np.zeros(self.n, bool)
        oi = self.cam_intr[self.obs_cam]
        for c in np.unique(self.obs_cam):
            act[6 * c:6 * c + 6] = True
        for a in np.unique(oi):
            act[6 * C + self.kd * a:6 * C + self.kd * (a + 1)] = True
        for p in np.unique(self.obs_pt):
            act[self.n_red + 3 * p:self.n_red + 3 * p + 3] = True
        self.block_const = dict(cam=np.zeros(C, bool), intr=np.zeros(NI, bool), pt=np.zeros(P, bool))
        if cam_const is not None:
            for c, fl in enumerate(np.asarray(cam_const)):
                if fl & 1:
                    act[6 * c:6 * c + 6] = False
                    self.block_const["cam"][c] = True
                for k in range(3):
                    if fl & (2 << k):
                        act[6 * c + 3 + k] = False
        if intr_const is not None:
            for a, fl in enumerate(np.asarray(intr_const)):
                if fl:
                    act[6 * C + self.kd * a:6 * C + self.kd * (a + 1)] = False
                    self.block_const["intr"][a] = True
        if pt_const is not None:
            for p, fl in enumerate(np.asarray(pt_const)):
                if fl:
                    act[self.n_red + 3 * p:self.n_red + 3 * p + 3] = False
                    self.block_const["pt"][p] = True
        self.active = act
        z3, z2 = torch.zeros(3, dtype=F64), torch.zeros(2, dtype=F64)
        jac = torch.func.jacrev(lambda dq, dt, dfk, dX, q, t, i, X, uv: _residual(dq, dt, dfk, dX, q, t, i, X, uv, self.radial),
                                argnums=(0, 1, 2, 3))
        self._jac = torch.func.vmap(lambda q, t, i, X, uv: jac(z3, z3, z2, z3, q, t, i, X, uv))
        self._res = torch.func.vmap(lambda q, t, i, X, uv: _residual(z3, z3, z2, z3, q, t, i, X, uv, self.radial))

    def _gather(self, q, t, intr, X):
        T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
        return (T(q)[self.obs_cam], T(t)[self.obs_cam], T(intr)[self.cam_intr[self.obs_cam]], T(X)[self.obs_pt], self.uv)

    def cost(self, q, t, intr, X):
        r = self._res(*self._gather(q, t, intr, X))
        return 0.5 * float(_rho(self.loss, self.loss_scale, (r * r).sum(1)).sum())

    def linearize(self, q, t, intr, X):
        """-> cost, corrected residual (2O,), corrected dense Jacobian (2O, n) with constant columns zero."""
        args = self._gather(q, t, intr, X)
        r = self._res(*args)
        Jq, Jt, Jfk, JX = self._jac(*args)
        s = (r * r).sum(1).clone().requires_grad_(True)
        rho = _rho(self.loss, self.loss_scale, s)
        (rho1,) = torch.autograd.grad(rho.sum(), s)
        w = torch.sqrt(rho1.detach())                     # Triggs correction with rho'' <= 0: sqrt(rho') only
        O = len(self.obs_cam)
        J = np.zeros((O, 2, self.n))
        idx = np.arange(O)
        for k in range(3):
            J[idx, :, 6 * self.obs_cam + k] = (w[:, None] * Jq[:, :, k]).numpy()
            J[idx, :, 6 * self.obs_cam + 3 + k] = (w[:, None] * Jt[:, :, k]).numpy()
            J[idx, :, self.n_red + 3 * self.obs_pt + k] = (w[:, None] * JX[:, :, k]).numpy()
        ic = 6 * self.C + self.kd * self.cam_intr[self.obs_cam]
        for j, src in enumerate(self.kcols):
            J[idx, :, ic + j] = (w[:, None] * Jfk[:, :, src]).numpy()
        J = J.reshape(2 * O, self.n)
        J[:, ~self.active] = 0.0
        return 0.5 * float(rho.detach().sum()), (w[:, None] * r).numpy().reshape(-1), J

    def plus(self, q, t, intr, X, delta):
        nq = np.stack([_quat_plus_np(q[c], delta[6 * c:6 * c + 3]) for c in range(self.C)])
        nt = t + delta[:6 * self.C].reshape(self.C, 6)[:, 3:]
        ni = intr.copy()
        for j, src in enumerate(self.kcols):
            ni[:, 3 if src == 1 else 0] += delta[6 * self.C + j:self.n_red:self.kd] if self.kd else 0.0
        nX = X + delta[self.n_red:].reshape(self.P, 3)
        return nq, nt, ni, nX

    def x_norm(self, q, t, intr, X):
        """|x| over the parameter blocks of Ceres' reduced program (constant blocks are removed from it)."""
        keep = ~self.block_const["cam"]
        s = (q[keep] ** 2).sum() + (t[keep] ** 2).sum() + (X[~self.block_const["pt"]] ** 2).sum()
        if self.kd:
            s += (intr[~self.block_const["intr"], :4 if self.radial else 3] ** 2).sum()
        return np.sqrt(s)


def solve_csr(cam_q, cam_t, intr, pts, cam_intr, row_ptr, obs_cam, obs_uv, camera_model, options, refine_focal=True,
              refine_extra=True, loss=0, loss_scale=1.0, cam_const=None, intr_const=None, pt_const=None):
    """Same contract as oracle.ba.solve_csr (arrays updated in place, the same summary dict), dense and autograd."""
    o = options
    pb = _Problem(cam_intr, row_ptr, obs_cam, obs_uv, camera_model, refine_focal, refine_extra, loss, loss_scale, cam_const,
                  intr_const, pt_const, len(cam_t), len(intr), len(pts))
    x = [np.array(cam_q, np.float64), np.array(cam_t, np.float64), np.array(intr, np.float64), np.array(pts, np.float64)]

    def diff(a, b):
        return np.concatenate([(u - v).ravel() for u, v in zip(a, b)])

    def gradient_max_norm(g):
        return np.abs(diff(x, pb.plus(*x, -g))).max()          # Ceres: |x - Plus(x, -g)|_inf

    cost, r, J = pb.linearize(*x)
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0))) if o.jacobi_scaling else np.ones(pb.n)
    g = J.T @ r
    gmax = gradient_max_norm(g)
    radius, shrink, reuse_diagonal, invalid = o.initial_trust_region_radius, 2.0, False, 0
    its = [dict(iteration=0, cost=cost, cost_change=0.0, gradient_max_norm=gmax, step_norm=0.0, relative_decrease=0.0,
                radius=radius, successful=True)]
    out = dict(initial_cost=cost, termination=0, num_successful_steps=0, num_unsuccessful_steps=0, n_reduced=pb.n_red)
    it, last_ok, diag = 0, True, None
    act = pb.active
    while True:
        if it >= o.max_num_iterations:
            out["termination"] = 0
            break
        if last_ok and gmax <= o.gradient_tolerance:
            out["termination"] = 1
            break
        if radius <= o.min_trust_region_radius:
            out["termination"] = 4
            break
        it += 1
        rec = dict(iteration=it, cost=cost, cost_change=0.0, gradient_max_norm=gmax, step_norm=0.0, relative_decrease=0.0,
                   radius=radius, successful=False)
        Js = J[:, act] * scale[act]
        if not reuse_diagonal:
            diag = np.clip((Js * Js).sum(0), o.min_lm_diagonal, o.max_lm_diagonal)
        reuse_diagonal = True
        A = Js.T @ Js + np.diag(diag / radius)
        ok = True
        try:
            y = np.linalg.solve(A, Js.T @ r)
            np.linalg.cholesky(A)
        except np.linalg.LinAlgError:
            ok = False
        if ok:
            step_s = -y
            Jd = Js @ step_s
            model_change = -float(Jd @ (r + 0.5 * Jd))
            ok = np.isfinite(y).all() and model_change > 0.0
        if not ok:
            invalid += 1
            if invalid >= o.max_num_consecutive_invalid_steps:
                out["termination"] = 5
                break
            radius, shrink = radius / shrink, shrink * 2.0
            last_ok = False
            out["num_unsuccessful_steps"] += 1
            rec["radius"] = radius
            its.append(rec)
            continue
        invalid = 0
        delta = np.zeros(pb.n)
        delta[act] = step_s * scale[act]
        cand = pb.plus(*x, delta)
        cand_cost = pb.cost(*cand)
        rec["step_norm"] = step_norm = float(np.linalg.norm(diff(x, cand)))
        rec["cost_change"] = cost - cand_cost
        if not step_norm > o.parameter_tolerance * (pb.x_norm(*x) + o.parameter_tolerance):
            out["termination"] = 3
            its.append(rec)
            break
        if abs(cost - cand_cost) <= o.function_tolerance * cost:
            out["termination"] = 2
            its.append(rec)
            break
        quality = (cost - cand_cost) / model_change
        rec["relative_decrease"] = quality
        if quality > o.min_relative_decrease:
            x = list(cand)
            cost, r, J = pb.linearize(*x)
            g = J.T @ r
            gmax = gradient_max_norm(g)
            radius = min(o.max_trust_region_radius, radius / max(1.0 / 3.0, 1.0 - (2.0 * quality - 1.0) ** 3))
            shrink, reuse_diagonal, last_ok = 2.0, False, True
            out["num_successful_steps"] += 1
            rec.update(successful=True, cost=cost, gradient_max_norm=gmax)
        else:
            radius, shrink = radius / shrink, shrink * 2.0
            last_ok = False
            out["num_unsuccessful_steps"] += 1
        rec["radius"] = radius
        its.append(rec)
    cam_q[...], cam_t[...], intr[...], pts[...] = x
    out.update(final_cost=cost, num_iterations=it, iterations=its)
    return out
