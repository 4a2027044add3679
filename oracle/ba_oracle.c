/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or called from the
 * product path (vggsfm_amd/).  Allowed users: tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py.
 *
 * PARITY UNPINNED.  The reference delegates bundle adjustment and pose refinement to
 * pycolmap==3.10.0 / pyceres==2.3 (reference install.sh:41-42), i.e. COLMAP's
 * BundleAdjuster on Ceres' trust-region Levenberg-Marquardt.  Neither package nor its
 * source is present in /root/reference or installable here (no network), and the reference
 * holds no golden vector for its BA calls.  This file is therefore a CPU restatement of the
 * *published* algorithm (COLMAP 3.10 src/colmap/estimators/{bundle_adjustment,cost_functions,
 * pose}.cc|h; Ceres 2.x internal/ceres/{trust_region_minimizer,levenberg_marquardt_strategy,
 * schur_eliminator_impl,corrector,loss_function,manifold}.cc) anchored on the reference's own
 * call sites:
 *   - problem construction:  vggsfm/utils/tensor_to_pycolmap.py:16-160
 *   - options per call site: vggsfm/utils/triangulation_helpers.py:626-635 (50 iterations,
 *     tolerances x10), vggsfm/models/triangulator.py:254-263 (defaults: 100 iterations)
 *   - bundle_adjustment():   vggsfm/utils/triangulation.py:213,1050,1142
 *   - pose_refinement():     vggsfm/utils/triangulation.py:387,590  (CauchyLoss(1), points fixed)
 * It is cross-checked in tests/ against numeric differentiation and scipy's least_squares
 * optimum; that pins the mathematics, not pycolmap's bits.
 *
 * Model restated (Ceres semantics):
 *   residual_i = ImgFromCam(intr, R(q) X + t) - uv_i            (2-vector, pixels)
 *   cost = 1/2 sum rho(|r_i|^2);  TRIVIAL loss for BA, Cauchy(a) for pose refinement
 *   quaternion manifold: q <- exp(delta) * q, delta in R^3 (EigenQuaternionManifold),
 *   subset manifolds for constant translation components / constant intrinsics,
 *   Jacobi column scaling s_j = 1/(1+|J_j|) frozen at iteration 0,
 *   LM: D^2 = clamp(diag(J^T J), 1e-6, 1e32) / radius, initial radius 1e4, max 1e16,
 *   Schur elimination of the points, dense Cholesky of the reduced camera system,
 *   rho = (cost - cost_new) / model_cost_change, accept if rho > 1e-3,
 *   radius /= max(1/3, 1 - (2 rho - 1)^3) on accept; radius /= factor, factor *= 2 on reject.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "ba_oracle.h"

/* ------------------------------------------------------------------ small helpers */
static void quat_rotate(const double* q, const double* v, double* out) {
  /* Eigen QuaternionBase::_transformVector for a unit quaternion (x,y,z,w) */
  double ux = q[1] * v[2] - q[2] * v[1], uy = q[2] * v[0] - q[0] * v[2], uz = q[0] * v[1] - q[1] * v[0];
  ux += ux; uy += uy; uz += uz;
  out[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
  out[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
  out[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}

static void quat_to_R(const double* q, double* R) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

/* Ceres QuaternionPlusImpl<EigenQuaternionOrder>: x_plus = q(delta) * x */
static void quat_plus(const double* x, const double* d, double* out) {
  const double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (n == 0.0) { memcpy(out, x, 4 * sizeof(double)); return; }
  const double s = sin(n) / n;
  const double qx = s * d[0], qy = s * d[1], qz = s * d[2], qw = cos(n);
  out[3] = qw * x[3] - qx * x[0] - qy * x[1] - qz * x[2];
  out[0] = qw * x[0] + qx * x[3] + qy * x[2] - qz * x[1];
  out[1] = qw * x[1] - qx * x[2] + qy * x[3] + qz * x[0];
  out[2] = qw * x[2] + qx * x[1] - qy * x[0] + qz * x[3];
}

/* residual + analytic Jacobians of one observation.
 * Jp: 2x6 (delta(3), t(3)); Ji: 2x2 (f, k); Jx: 2x3.  Returns 0 always. */
static void obs_eval(int model, const double* q, const double* t, const double* intr, const double* X,
                     double u_obs, double v_obs, double* r, double* Jp, double* Ji, double* Jx) {
  double a[3];
  quat_rotate(q, X, a);
  const double Y0 = a[0] + t[0], Y1 = a[1] + t[1], Y2 = a[2] + t[2];
  const double f = intr[0], cx = intr[1], cy = intr[2], k = (model == 1) ? intr[3] : 0.0;
  const double iz = 1.0 / Y2;
  const double u = Y0 * iz, v = Y1 * iz;
  const double r2 = u * u + v * v;
  const double d = 1.0 + k * r2;
  r[0] = f * (u * d) + cx - u_obs;
  r[1] = f * (v * d) + cy - v_obs;
  if (!Jp) return;
  const double xu = f * (d + 2 * k * u * u), xv = f * (2 * k * u * v), yu = xv, yv = f * (d + 2 * k * v * v);
  /* J_Y = [xu xv; yu yv] * (1/Y2) [1 0 -u; 0 1 -v] */
  double JY[6];
  JY[0] = xu * iz; JY[1] = xv * iz; JY[2] = -(xu * u + xv * v) * iz;
  JY[3] = yu * iz; JY[4] = yv * iz; JY[5] = -(yu * u + yv * v) * iz;
  double R[9];
  quat_to_R(q, R);
  for (int row = 0; row < 2; ++row) {
    const double* j = JY + 3 * row;
    /* d/d delta = 2 * J_Y * (e_k x a) */
    Jp[row * 6 + 0] = 2.0 * (j[2] * a[1] - j[1] * a[2]);
    Jp[row * 6 + 1] = 2.0 * (j[0] * a[2] - j[2] * a[0]);
    Jp[row * 6 + 2] = 2.0 * (j[1] * a[0] - j[0] * a[1]);
    Jp[row * 6 + 3] = j[0]; Jp[row * 6 + 4] = j[1]; Jp[row * 6 + 5] = j[2];
    for (int c = 0; c < 3; ++c) Jx[row * 3 + c] = j[0] * R[c] + j[1] * R[3 + c] + j[2] * R[6 + c];
  }
  Ji[0] = u * d; Ji[1] = f * r2 * u;
  Ji[2] = v * d; Ji[3] = f * r2 * v;
}

/* Ceres LossFunction::Evaluate -> rho[3] */
static void loss_eval(int loss, double a, double s, double* rho) {
  const double b = a * a;
  switch (loss) {
    case BAO_LOSS_CAUCHY: {
      const double sum = 1.0 + s / b, inv = 1.0 / sum;
      rho[0] = b * log(sum); rho[1] = inv > DBL_MIN ? inv : DBL_MIN; rho[2] = -(1.0 / b) * (inv * inv);
      break;
    }
    case BAO_LOSS_HUBER:
      if (s > b) { const double r = sqrt(s); rho[0] = 2 * a * r - b; rho[1] = a / r > DBL_MIN ? a / r : DBL_MIN; rho[2] = -rho[1] / (2 * s); }
      else { rho[0] = s; rho[1] = 1; rho[2] = 0; }
      break;
    case BAO_LOSS_SOFT_L1: {
      const double sum = 1.0 + s / b, tmp = sqrt(sum);
      rho[0] = 2 * b * (tmp - 1); rho[1] = 1 / tmp > DBL_MIN ? 1 / tmp : DBL_MIN; rho[2] = -1 / (2 * b * tmp * sum);
      break;
    }
    default: rho[0] = s; rho[1] = 1; rho[2] = 0;
  }
}

/* Ceres Corrector: rescale residual (2) and Jacobian rows (2 x ncols, row-major blocks) */
typedef struct { double sqrt_rho1, residual_scaling, alpha_sq_norm; } corrector_t;
static corrector_t corrector_make(double sq_norm, const double* rho) {
  corrector_t c;
  c.sqrt_rho1 = sqrt(rho[1]);
  if (sq_norm == 0.0 || rho[2] <= 0.0) { c.residual_scaling = c.sqrt_rho1; c.alpha_sq_norm = 0.0; return c; }
  const double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
  const double alpha = 1.0 - sqrt(D);
  c.residual_scaling = c.sqrt_rho1 / (1 - alpha);
  c.alpha_sq_norm = alpha / sq_norm;
  return c;
}
static void corrector_jac(const corrector_t* c, const double* r, double* J, int ncols) {
  if (c->alpha_sq_norm == 0.0) { for (int i = 0; i < 2 * ncols; ++i) J[i] *= c->sqrt_rho1; return; }
  for (int col = 0; col < ncols; ++col) {
    const double rtj = r[0] * J[col] + r[1] * J[ncols + col];
    J[col] = c->sqrt_rho1 * (J[col] - c->alpha_sq_norm * r[0] * rtj);
    J[ncols + col] = c->sqrt_rho1 * (J[ncols + col] - c->alpha_sq_norm * r[1] * rtj);
  }
}

static int chol3_inv(const double* A, double* inv) { /* A sym 3x3 row-major -> inverse; 0 ok */
  double l00 = A[0]; if (!(l00 > 0)) return 1; l00 = sqrt(l00);
  const double l10 = A[3] / l00, l20 = A[6] / l00;
  double l11 = A[4] - l10 * l10; if (!(l11 > 0)) return 1; l11 = sqrt(l11);
  const double l21 = (A[7] - l20 * l10) / l11;
  double l22 = A[8] - l20 * l20 - l21 * l21; if (!(l22 > 0)) return 1; l22 = sqrt(l22);
  /* inverse of L */
  const double i00 = 1 / l00, i11 = 1 / l11, i22 = 1 / l22;
  const double i10 = -l10 * i00 * i11;
  const double i21 = -l21 * i11 * i22;
  const double i20 = -(l20 * i00 + l21 * i10) * i22;
  /* inv = L^-T L^-1 */
  inv[0] = i00 * i00 + i10 * i10 + i20 * i20; inv[1] = i10 * i11 + i20 * i21; inv[2] = i20 * i22;
  inv[3] = inv[1]; inv[4] = i11 * i11 + i21 * i21; inv[5] = i21 * i22;
  inv[6] = inv[2]; inv[7] = inv[5]; inv[8] = i22 * i22;
  return 0;
}

/* in-place dense Cholesky (lower) + solve; returns 0 ok */
static int dense_chol_solve(double* A, double* b, int n) {
  const int NB = 48;
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int kb = (k0 + NB < n) ? NB : n - k0;
    for (int j = k0; j < k0 + kb; ++j) {               /* factor diagonal block + panel column j */
      double d = A[(size_t)j * n + j];
      for (int k = k0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k];
      if (!(d > 0.0) || !isfinite(d)) return 1;
      d = sqrt(d); A[(size_t)j * n + j] = d;
      const double id = 1.0 / d;
#pragma omp parallel for schedule(static) if (n - j > 256)
      for (int i = j + 1; i < n; ++i) {
        double s = A[(size_t)i * n + j];
        for (int k = k0; k < j; ++k) s -= A[(size_t)i * n + k] * A[(size_t)j * n + k];
        A[(size_t)i * n + j] = s * id;
      }
    }
    /* trailing update: A[i][j] -= sum_k L[i][k] L[j][k], i>=j>=k0+kb */
#pragma omp parallel for schedule(dynamic, 8)
    for (int i = k0 + kb; i < n; ++i) {
      const double* Li = A + (size_t)i * n + k0;
      for (int j = k0 + kb; j <= i; ++j) {
        const double* Lj = A + (size_t)j * n + k0;
        double s = 0;
        for (int k = 0; k < kb; ++k) s += Li[k] * Lj[k];
        A[(size_t)i * n + j] -= s;
      }
    }
  }
  for (int i = 0; i < n; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= A[(size_t)i * n + k] * b[k]; b[i] = s / A[(size_t)i * n + i]; }
  for (int i = n - 1; i >= 0; --i) { double s = b[i]; for (int k = i + 1; k < n; ++k) s -= A[(size_t)k * n + i] * b[k]; b[i] = s / A[(size_t)i * n + i]; }
  return 0;
}

/* ------------------------------------------------------------------ solver state */
typedef struct {
  const bao_problem_t* pb;
  int kd;                 /* refined intrinsics per block: f and/or k */
  int idx_f, idx_k;       /* position of f / k inside the intr tangent block, or -1 */
  int n_red;              /* 6*C + kd*n_intr */
  int n_cols;             /* n_red + 3*P */
  uint8_t* active;        /* per column */
  double* scale;          /* Jacobi scaling per column */
  /* Reproducibility: every sum over observations that feeds a decision (cost, gradient, column norms, the
   * reduced system, the model cost change) is formed slice by slice -- K contiguous point ranges that are a
   * function of the PROBLEM only (balanced by observation count), each summed sequentially into its own
   * buffer -- and the K partial results are then added in slice order.  Threads only decide who computes a
   * slice, never the order of a floating-point addition: the iteration log is bit-identical from run to run
   * and for every OMP_NUM_THREADS (tests/test_oracle_ba.py::test_oracle_is_bit_reproducible). */
  int num_slices;
  int* slice_begin;       /* [num_slices + 1] point index */
  double* slice_lhs;      /* [num_slices][n_red * n_red] */
  double* slice_vec;      /* [num_slices][2 * n_red] */
} ctx_t;

static void make_slices(ctx_t* c) {
  const bao_problem_t* pb = c->pb;
  const long O = pb->row_ptr[pb->num_pts];
  long K = O / 2048;
  if (K > 128) K = 128;
  /* keep the K partial systems below ~3 GB */
  const double per = 8.0 * (double)c->n_red * (double)c->n_red;
  while (K > 1 && per * (double)K > 3.0e9) K /= 2;
  if (K < 1) K = 1;
  if (K > pb->num_pts) K = pb->num_pts > 0 ? pb->num_pts : 1;
  c->num_slices = (int)K;
  c->slice_begin = (int*)malloc(sizeof(int) * (K + 1));
  int p = 0;
  for (long s = 0; s < K; ++s) {
    const long target = O * s / K;
    while (p < pb->num_pts && pb->row_ptr[p] < target) ++p;
    c->slice_begin[s] = p;
  }
  c->slice_begin[0] = 0;
  c->slice_begin[K] = pb->num_pts;
  c->slice_lhs = (double*)malloc(sizeof(double) * (size_t)K * c->n_red * c->n_red);
  c->slice_vec = (double*)malloc(sizeof(double) * (size_t)K * 2 * c->n_red);
}

static int intr_col(const ctx_t* c, int a) { return 6 * c->pb->num_cams + c->kd * a; }
static int pt_col(const ctx_t* c, int p) { return c->n_red + 3 * p; }

/* evaluate observation o of point p at state (q,t,intr,pts): corrected residual r and the
 * *unscaled* corrected tangent Jacobians; returns rho(s)  */
static double eval_corrected(const ctx_t* c, const double* cq, const double* ct, const double* intr,
                             const double* pts, int p, int o, double* r, double* Jp, double* Ji2, double* Jx) {
  const bao_problem_t* pb = c->pb;
  const int cam = pb->obs_cam[o];
  double Ji[4];
  obs_eval(pb->camera_model, cq + 4 * cam, ct + 3 * cam, intr + 4 * pb->cam_intr[cam], pts + 3 * p,
           pb->obs_uv[2 * o], pb->obs_uv[2 * o + 1], r, Jp, Ji, Jx);
  const double s = r[0] * r[0] + r[1] * r[1];
  double rho[3];
  loss_eval(pb->loss, pb->loss_scale, s, rho);
  if (Jp) {
    /* compact intrinsics block to the refined parameters */
    for (int row = 0; row < 2; ++row) {
      if (c->idx_f >= 0) Ji2[row * c->kd + c->idx_f] = Ji[row * 2 + 0];
      if (c->idx_k >= 0) Ji2[row * c->kd + c->idx_k] = Ji[row * 2 + 1];
    }
    if (pb->loss != BAO_LOSS_TRIVIAL) {
      corrector_t cr = corrector_make(s, rho);
      corrector_jac(&cr, r, Jp, 6);
      if (c->kd) corrector_jac(&cr, r, Ji2, c->kd);
      corrector_jac(&cr, r, Jx, 3);
      r[0] *= cr.residual_scaling; r[1] *= cr.residual_scaling;
    }
  }
  return rho[0];
}

static double total_cost(const ctx_t* c, const double* cq, const double* ct, const double* intr, const double* pts) {
  const bao_problem_t* pb = c->pb;
  const int K = c->num_slices;
  double part[128];
#pragma omp parallel for schedule(dynamic, 1)
  for (int s = 0; s < K; ++s) {
    double cs = 0;
    for (int p = c->slice_begin[s]; p < c->slice_begin[s + 1]; ++p) {
      double cp = 0;
      for (int o = pb->row_ptr[p]; o < pb->row_ptr[p + 1]; ++o) {
        double r[2];
        cp += eval_corrected(c, cq, ct, intr, pts, p, o, r, NULL, NULL, NULL);
      }
      cs += cp;
    }
    part[s] = cs;
  }
  double cost = 0;
  for (int s = 0; s < K; ++s) cost += part[s];
  return 0.5 * cost;
}

/* zero the Jacobian columns of constant tangent dimensions */
static void mask_jac(const ctx_t* c, int cam, int p, double* Jp, double* Ji, double* Jx) {
  const bao_problem_t* pb = c->pb;
  for (int k = 0; k < 6; ++k) if (!c->active[6 * cam + k]) { Jp[k] = 0; Jp[6 + k] = 0; }
  const int ic = intr_col(c, pb->cam_intr[cam]);
  for (int k = 0; k < c->kd; ++k) if (!c->active[ic + k]) { Ji[k] = 0; Ji[c->kd + k] = 0; }
  for (int k = 0; k < 3; ++k) if (!c->active[pt_col(c, p) + k]) { Jx[k] = 0; Jx[3 + k] = 0; }
}

/* gradient (unscaled) and squared column norms (unscaled) */
static void grad_and_colnorm(const ctx_t* c, const double* cq, const double* ct, const double* intr,
                             const double* pts, double* grad, double* colsq) {
  const bao_problem_t* pb = c->pb;
  const int nr = c->n_red;
  memset(grad, 0, sizeof(double) * c->n_cols);
  memset(colsq, 0, sizeof(double) * c->n_cols);
  const int K = c->num_slices;
#pragma omp parallel for schedule(dynamic, 1)
  for (int s = 0; s < K; ++s) {
    double* lg = c->slice_vec + (size_t)s * 2 * nr;                 /* camera-side partials of this slice */
    double* lc = lg + nr;
    memset(lg, 0, sizeof(double) * 2 * nr);
    for (int p = c->slice_begin[s]; p < c->slice_begin[s + 1]; ++p)
      for (int o = pb->row_ptr[p]; o < pb->row_ptr[p + 1]; ++o) {
        const int cam = pb->obs_cam[o];
        double r[2], Jp[12], Ji[4], Jx[6];
        eval_corrected(c, cq, ct, intr, pts, p, o, r, Jp, Ji, Jx);
        mask_jac(c, cam, p, Jp, Ji, Jx);
        for (int k = 0; k < 6; ++k) { lg[6 * cam + k] += Jp[k] * r[0] + Jp[6 + k] * r[1]; lc[6 * cam + k] += Jp[k] * Jp[k] + Jp[6 + k] * Jp[6 + k]; }
        const int ic = intr_col(c, pb->cam_intr[cam]);
        for (int k = 0; k < c->kd; ++k) { lg[ic + k] += Ji[k] * r[0] + Ji[c->kd + k] * r[1]; lc[ic + k] += Ji[k] * Ji[k] + Ji[c->kd + k] * Ji[c->kd + k]; }
        const int pc = pt_col(c, p);
        for (int k = 0; k < 3; ++k) { grad[pc + k] += Jx[k] * r[0] + Jx[3 + k] * r[1]; colsq[pc + k] += Jx[k] * Jx[k] + Jx[3 + k] * Jx[3 + k]; }
      }
  }
  for (int s = 0; s < K; ++s) {                                     /* slice order: fixed */
    const double* lg = c->slice_vec + (size_t)s * 2 * nr;
    const double* lc = lg + nr;
    for (int i = 0; i < nr; ++i) { grad[i] += lg[i]; colsq[i] += lc[i]; }
  }
}

/* x (+) delta for the whole state */
static void state_plus(const ctx_t* c, const double* cq, const double* ct, const double* intr, const double* pts,
                       const double* delta, double* nq, double* nt, double* nintr, double* npts) {
  const bao_problem_t* pb = c->pb;
  for (int cam = 0; cam < pb->num_cams; ++cam) {
    quat_plus(cq + 4 * cam, delta + 6 * cam, nq + 4 * cam);
    for (int k = 0; k < 3; ++k) nt[3 * cam + k] = ct[3 * cam + k] + delta[6 * cam + 3 + k];
  }
  memcpy(nintr, intr, sizeof(double) * 4 * pb->num_intr);
  for (int a = 0; a < pb->num_intr; ++a) {
    const int ic = intr_col(c, a);
    if (c->idx_f >= 0) nintr[4 * a + 0] += delta[ic + c->idx_f];
    if (c->idx_k >= 0) nintr[4 * a + 3] += delta[ic + c->idx_k];
  }
  for (int i = 0; i < 3 * pb->num_pts; ++i) npts[i] = pts[i] + delta[c->n_red + i];
}

static double ambient_diff_norm(const ctx_t* c, const double* q0, const double* t0, const double* i0, const double* p0,
                                const double* q1, const double* t1, const double* i1, const double* p1, double* maxabs) {
  const bao_problem_t* pb = c->pb;
  double s = 0, m = 0;
#define ACC(a, b) { const double d_ = (a) - (b); s += d_ * d_; if (fabs(d_) > m) m = fabs(d_); }
  for (int i = 0; i < 4 * pb->num_cams; ++i) ACC(q0[i], q1[i]);
  for (int i = 0; i < 3 * pb->num_cams; ++i) ACC(t0[i], t1[i]);
  for (int i = 0; i < 4 * pb->num_intr; ++i) ACC(i0[i], i1[i]);
  for (int i = 0; i < 3 * pb->num_pts; ++i) ACC(p0[i], p1[i]);
#undef ACC
  if (maxabs) *maxabs = m;
  return sqrt(s);
}

/* test hook: when set, the first reduced system (after damping / constant-column handling, before the
 * factorisation) is copied out: lhs n*n row-major (full symmetric), rhs n */
static double* g_dump_lhs = NULL;
static double* g_dump_rhs = NULL;
void bao_debug_dump_system(double* lhs, double* rhs) { g_dump_lhs = lhs; g_dump_rhs = rhs; }

/* Build and solve the damped normal equations by Schur elimination; y solves
 * (Js^T Js + D^2) y = Js^T r with Js = J diag(scale).  Also returns
 * model_cost_change for step = -y.  Returns 0 ok / 1 linear solver failure. */
static int schur_solve(const ctx_t* c, const double* cq, const double* ct, const double* intr, const double* pts,
                       const double* D, double* y, double* lhs, double* rhs, double* model_cost_change) {
  const bao_problem_t* pb = c->pb;
  const int n = c->n_red, P = pb->num_pts, kd = c->kd, bw = 6 + kd;
  double* einv = (double*)malloc(sizeof(double) * 9 * (size_t)P);
  double* ge = (double*)malloc(sizeof(double) * 3 * (size_t)P);
  int fail = 0;
  const int K = c->num_slices;
#pragma omp parallel for schedule(dynamic, 1)
  for (int s = 0; s < K; ++s) {
    double* L = c->slice_lhs + (size_t)s * n * n;
    double* Rh = c->slice_vec + (size_t)s * 2 * n;
    memset(L, 0, sizeof(double) * (size_t)n * n);
    memset(Rh, 0, sizeof(double) * n);
    int cap = 64;
    double* FtE = (double*)malloc(sizeof(double) * cap * bw * 3);
    int* cols = (int*)malloc(sizeof(int) * cap * bw);
    for (int p = c->slice_begin[s]; p < c->slice_begin[s + 1]; ++p) {
      const int o0 = pb->row_ptr[p], no = pb->row_ptr[p + 1] - o0;
      if (no > cap) { cap = no; FtE = (double*)realloc(FtE, sizeof(double) * cap * bw * 3); cols = (int*)realloc(cols, sizeof(int) * cap * bw); }
      const int pc = pt_col(c, p);
      const int eliminate = c->active[pc] || c->active[pc + 1] || c->active[pc + 2];
      double ete[9] = {0}, g[3] = {0};
      for (int i = 0; i < no; ++i) {
        const int o = o0 + i, cam = pb->obs_cam[o], ic = intr_col(c, pb->cam_intr[cam]);
        double r[2], Jp[12], Ji[4], Jx[6], F[2 * 8];
        eval_corrected(c, cq, ct, intr, pts, p, o, r, Jp, Ji, Jx);
        mask_jac(c, cam, p, Jp, Ji, Jx);
        int* cl = cols + i * bw;
        for (int k = 0; k < 6; ++k) { cl[k] = 6 * cam + k; F[k] = Jp[k] * c->scale[cl[k]]; F[bw + k] = Jp[6 + k] * c->scale[cl[k]]; }
        for (int k = 0; k < kd; ++k) { cl[6 + k] = ic + k; F[6 + k] = Ji[k] * c->scale[ic + k]; F[bw + 6 + k] = Ji[kd + k] * c->scale[ic + k]; }
        for (int k = 0; k < 3; ++k) { Jx[k] *= c->scale[pc + k]; Jx[3 + k] *= c->scale[pc + k]; }
        /* F^T F and F^T r */
        for (int a = 0; a < bw; ++a) {
          Rh[cl[a]] += F[a] * r[0] + F[bw + a] * r[1];
          for (int b = 0; b < bw; ++b) L[(size_t)cl[a] * n + cl[b]] += F[a] * F[b] + F[bw + a] * F[bw + b];
        }
        if (eliminate) {
          for (int a = 0; a < 3; ++a) { g[a] += Jx[a] * r[0] + Jx[3 + a] * r[1]; for (int b = 0; b < 3; ++b) ete[3 * a + b] += Jx[a] * Jx[b] + Jx[3 + a] * Jx[3 + b]; }
          double* W = FtE + (size_t)i * bw * 3;
          for (int a = 0; a < bw; ++a) for (int b = 0; b < 3; ++b) W[a * 3 + b] = F[a] * Jx[b] + F[bw + a] * Jx[3 + b];
        }
      }
      if (!eliminate) { memset(einv + 9 * (size_t)p, 0, 72); memset(ge + 3 * (size_t)p, 0, 24); continue; }
      for (int k = 0; k < 3; ++k) ete[4 * k] += D[pc + k] * D[pc + k];
      for (int k = 0; k < 3; ++k) if (!c->active[pc + k]) ete[4 * k] = 1.0;   /* constant coordinate (unused today) */
      double* inv = einv + 9 * (size_t)p;
      if (chol3_inv(ete, inv)) {
#pragma omp atomic write
        fail = 1;
        continue;
      }
      memcpy(ge + 3 * (size_t)p, g, 24);
      double ig[3];
      for (int a = 0; a < 3; ++a) ig[a] = inv[3 * a] * g[0] + inv[3 * a + 1] * g[1] + inv[3 * a + 2] * g[2];
      for (int i = 0; i < no; ++i) {
        const double* Wi = FtE + (size_t)i * bw * 3;
        const int* ci = cols + i * bw;
        double WiV[8 * 3];
        for (int a = 0; a < bw; ++a) for (int b = 0; b < 3; ++b) WiV[a * 3 + b] = Wi[a * 3] * inv[b] + Wi[a * 3 + 1] * inv[3 + b] + Wi[a * 3 + 2] * inv[6 + b];
        for (int a = 0; a < bw; ++a) Rh[ci[a]] -= Wi[a * 3] * ig[0] + Wi[a * 3 + 1] * ig[1] + Wi[a * 3 + 2] * ig[2];
        for (int j = 0; j < no; ++j) {
          const double* Wj = FtE + (size_t)j * bw * 3;
          const int* cj = cols + j * bw;
          for (int a = 0; a < bw; ++a) {
            double* row = L + (size_t)ci[a] * n;
            for (int b = 0; b < bw; ++b) row[cj[b]] -= WiV[a * 3] * Wj[b * 3] + WiV[a * 3 + 1] * Wj[b * 3 + 1] + WiV[a * 3 + 2] * Wj[b * 3 + 2];
          }
        }
      }
    }
    free(FtE); free(cols);
  }
  /* the K partial systems, added in slice order (element-wise parallel: the order per element is fixed) */
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)n * n; ++i) {
    double a = 0;
    for (int s = 0; s < K; ++s) a += c->slice_lhs[(size_t)s * n * n + i];
    lhs[i] = a;
  }
  for (int i = 0; i < n; ++i) {
    double a = 0;
    for (int s = 0; s < K; ++s) a += c->slice_vec[(size_t)s * 2 * n + i];
    rhs[i] = a;
  }
  if (!fail) {
    for (int i = 0; i < n; ++i) {
      if (c->active[i]) lhs[(size_t)i * n + i] += D[i] * D[i];
      else { for (int j = 0; j < n; ++j) { lhs[(size_t)i * n + j] = 0; lhs[(size_t)j * n + i] = 0; } lhs[(size_t)i * n + i] = 1.0; rhs[i] = 0; }
    }
    if (g_dump_lhs) { memcpy(g_dump_lhs, lhs, sizeof(double) * (size_t)n * n); memcpy(g_dump_rhs, rhs, sizeof(double) * n); g_dump_lhs = NULL; g_dump_rhs = NULL; }
    memcpy(y, rhs, sizeof(double) * n);
    fail = dense_chol_solve(lhs, y, n);
  }
  if (!fail) {
    /* back-substitution y_e = (E^T E + D_e^2)^-1 (E^T b - E^T F y_f) and model cost change */
    double mpart[128];
#pragma omp parallel for schedule(dynamic, 1)
    for (int s = 0; s < K; ++s) {
    double mcc = 0;
    for (int p = c->slice_begin[s]; p < c->slice_begin[s + 1]; ++p) {
      const int pc = pt_col(c, p);
      const int eliminate = c->active[pc] || c->active[pc + 1] || c->active[pc + 2];
      double acc[3] = {0, 0, 0};
      const int o0 = pb->row_ptr[p], o1 = pb->row_ptr[p + 1];
      if (eliminate) {
        for (int o = o0; o < o1; ++o) {
          const int cam = pb->obs_cam[o], ic = intr_col(c, pb->cam_intr[cam]);
          double r[2], Jp[12], Ji[4], Jx[6];
          eval_corrected(c, cq, ct, intr, pts, p, o, r, Jp, Ji, Jx);
          mask_jac(c, cam, p, Jp, Ji, Jx);
          double fy0 = 0, fy1 = 0;
          for (int k = 0; k < 6; ++k) { const double s = c->scale[6 * cam + k] * y[6 * cam + k]; fy0 += Jp[k] * s; fy1 += Jp[6 + k] * s; }
          for (int k = 0; k < kd; ++k) { const double s = c->scale[ic + k] * y[ic + k]; fy0 += Ji[k] * s; fy1 += Ji[kd + k] * s; }
          for (int k = 0; k < 3; ++k) acc[k] += c->scale[pc + k] * (Jx[k] * fy0 + Jx[3 + k] * fy1);
        }
        const double* inv = einv + 9 * (size_t)p;
        const double* g = ge + 3 * (size_t)p;
        const double b0 = g[0] - acc[0], b1 = g[1] - acc[1], b2 = g[2] - acc[2];
        for (int a = 0; a < 3; ++a) y[pc + a] = inv[3 * a] * b0 + inv[3 * a + 1] * b1 + inv[3 * a + 2] * b2;
      } else {
        y[pc] = y[pc + 1] = y[pc + 2] = 0;
      }
      /* model_cost_change = -(J s)^T (r + J s / 2) with s = -y */
      for (int o = o0; o < o1; ++o) {
        const int cam = pb->obs_cam[o], ic = intr_col(c, pb->cam_intr[cam]);
        double r[2], Jp[12], Ji[4], Jx[6];
        eval_corrected(c, cq, ct, intr, pts, p, o, r, Jp, Ji, Jx);
        mask_jac(c, cam, p, Jp, Ji, Jx);
        double m0 = 0, m1 = 0;
        for (int k = 0; k < 6; ++k) { const double s = -c->scale[6 * cam + k] * y[6 * cam + k]; m0 += Jp[k] * s; m1 += Jp[6 + k] * s; }
        for (int k = 0; k < kd; ++k) { const double s = -c->scale[ic + k] * y[ic + k]; m0 += Ji[k] * s; m1 += Ji[kd + k] * s; }
        for (int k = 0; k < 3; ++k) { const double s = -c->scale[pc + k] * y[pc + k]; m0 += Jx[k] * s; m1 += Jx[3 + k] * s; }
        mcc += -(m0 * (r[0] + m0 / 2.0) + m1 * (r[1] + m1 / 2.0));
      }
    }
    mpart[s] = mcc;
    }
    double mcc = 0;
    for (int s = 0; s < K; ++s) mcc += mpart[s];
    *model_cost_change = mcc;
    for (int i = 0; i < c->n_cols; ++i) if (!isfinite(y[i])) { fail = 1; break; }
  }
  free(einv); free(ge);
  return fail;
}

/* ------------------------------------------------------------------ public entry */
int bao_solve(const bao_problem_t* pb, const bao_options_t* opt, double* cam_q, double* cam_t, double* intr,
              double* pts, bao_summary_t* sum, bao_iter_t* log, int log_cap) {
  ctx_t c;
  memset(&c, 0, sizeof(c));
  c.pb = pb;
  c.idx_f = c.idx_k = -1;
  c.kd = 0;
  if (pb->refine_focal) c.idx_f = c.kd++;
  if (pb->refine_extra && pb->camera_model == 1) c.idx_k = c.kd++;
  c.n_red = 6 * pb->num_cams + c.kd * pb->num_intr;
  c.n_cols = c.n_red + 3 * pb->num_pts;
  const int C = pb->num_cams, P = pb->num_pts, NI = pb->num_intr, n = c.n_cols;
  make_slices(&c);
  c.active = (uint8_t*)malloc(n);
  c.scale = (double*)malloc(sizeof(double) * n);
  /* Ceres removes parameter blocks (and subset-manifold coordinates) that are constant, and
   * COLMAP only adds cameras/points that carry observations */
  memset(c.active, 0, n);
  for (int p = 0; p < P; ++p)
    for (int o = pb->row_ptr[p]; o < pb->row_ptr[p + 1]; ++o) {
      const int cam = pb->obs_cam[o];
      for (int k = 0; k < 6; ++k) c.active[6 * cam + k] = 1;
      for (int k = 0; k < c.kd; ++k) c.active[intr_col(&c, pb->cam_intr[cam]) + k] = 1;
      for (int k = 0; k < 3; ++k) c.active[pt_col(&c, p) + k] = 1;
    }
  for (int cam = 0; cam < C; ++cam) {
    const uint8_t f = pb->cam_const ? pb->cam_const[cam] : 0;
    for (int k = 0; k < 3; ++k) if (f & 1) c.active[6 * cam + k] = 0;
    for (int k = 0; k < 3; ++k) if ((f & 1) || (f & (2u << k))) c.active[6 * cam + 3 + k] = 0;
  }
  if (pb->intr_const) for (int a = 0; a < NI; ++a) if (pb->intr_const[a]) for (int k = 0; k < c.kd; ++k) c.active[intr_col(&c, a) + k] = 0;
  if (pb->pt_const) for (int p = 0; p < P; ++p) if (pb->pt_const[p]) for (int k = 0; k < 3; ++k) c.active[pt_col(&c, p) + k] = 0;

  double* grad = (double*)malloc(sizeof(double) * n);
  double* colsq = (double*)malloc(sizeof(double) * n);
  double* diag = (double*)malloc(sizeof(double) * n);
  double* D = (double*)malloc(sizeof(double) * n);
  double* y = (double*)malloc(sizeof(double) * n);
  double* delta = (double*)malloc(sizeof(double) * n);
  double* lhs = (double*)malloc(sizeof(double) * (size_t)c.n_red * c.n_red);
  double* rhs = (double*)malloc(sizeof(double) * c.n_red);
  double* nq = (double*)malloc(sizeof(double) * 4 * C);
  double* nt = (double*)malloc(sizeof(double) * 3 * C);
  double* ni = (double*)malloc(sizeof(double) * 4 * NI);
  double* np_ = (double*)malloc(sizeof(double) * 3 * P);

  double radius = opt->initial_trust_region_radius, decrease_factor = 2.0;
  int reuse_diagonal = 0, invalid_streak = 0;
  memset(sum, 0, sizeof(*sum));
  sum->termination = BAO_NO_CONVERGENCE;

  /* iteration zero */
  double x_cost = total_cost(&c, cam_q, cam_t, intr, pts);
  grad_and_colnorm(&c, cam_q, cam_t, intr, pts, grad, colsq);
  for (int i = 0; i < n; ++i) c.scale[i] = opt->jacobi_scaling ? 1.0 / (1.0 + sqrt(colsq[i])) : 1.0;
  sum->initial_cost = x_cost;
  double gmax;
  {
    for (int i = 0; i < n; ++i) delta[i] = c.active[i] ? -grad[i] : 0.0;
    state_plus(&c, cam_q, cam_t, intr, pts, delta, nq, nt, ni, np_);
    ambient_diff_norm(&c, cam_q, cam_t, intr, pts, nq, nt, ni, np_, &gmax);
  }
  int it = 0, step_successful = 1;
  if (log && log_cap > 0) { bao_iter_t z = {0, x_cost, 0, gmax, 0, 0, radius, 1}; log[0] = z; }
  sum->num_log = 1;
  for (;;) {
    /* FinalizeIterationAndCheckIfMinimizerCanContinue */
    if (it >= opt->max_num_iterations) { sum->termination = BAO_NO_CONVERGENCE; break; }
    if (step_successful && gmax <= opt->gradient_tolerance) { sum->termination = BAO_CONVERGENCE_GRADIENT; break; }
    if (radius <= opt->min_trust_region_radius) { sum->termination = BAO_CONVERGENCE_RADIUS; break; }
    ++it;
    bao_iter_t li = {it, x_cost, 0, gmax, 0, 0, radius, 0};
    /* LevenbergMarquardtStrategy::ComputeStep */
    if (!reuse_diagonal) {
      for (int i = 0; i < n; ++i) {
        double d = colsq[i] * c.scale[i] * c.scale[i];
        d = d < opt->min_lm_diagonal ? opt->min_lm_diagonal : d;
        diag[i] = d > opt->max_lm_diagonal ? opt->max_lm_diagonal : d;
      }
    }
    for (int i = 0; i < n; ++i) D[i] = sqrt(diag[i] / radius);
    double mcc = 0;
    int bad = schur_solve(&c, cam_q, cam_t, intr, pts, D, y, lhs, rhs, &mcc);
    reuse_diagonal = 1;
    int step_valid = !bad && (mcc > 0.0);
    if (!step_valid) {
      if (++invalid_streak >= opt->max_num_consecutive_invalid_steps) { sum->termination = BAO_FAILURE; break; }
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = 1;
      step_successful = 0;
      ++sum->num_unsuccessful_steps;
      li.radius = radius;
      if (log && it < log_cap) log[it] = li;
      sum->num_log = it + 1;
      continue;
    }
    invalid_streak = 0;
    for (int i = 0; i < n; ++i) delta[i] = c.active[i] ? -y[i] * c.scale[i] : 0.0;
    state_plus(&c, cam_q, cam_t, intr, pts, delta, nq, nt, ni, np_);
    const double cand_cost = total_cost(&c, nq, nt, ni, np_);
    const double step_norm = ambient_diff_norm(&c, cam_q, cam_t, intr, pts, nq, nt, ni, np_, NULL);
    li.step_norm = step_norm;
    li.cost_change = x_cost - cand_cost;
    /* ParameterToleranceReached */
    {
      /* |x| over the parameter blocks Ceres keeps in the reduced program (constant blocks are removed) */
      double xs = 0;
      for (int cam = 0; cam < C; ++cam) {
        if (pb->cam_const && (pb->cam_const[cam] & 1)) continue;
        for (int k = 0; k < 4; ++k) xs += cam_q[4 * cam + k] * cam_q[4 * cam + k];
        for (int k = 0; k < 3; ++k) xs += cam_t[3 * cam + k] * cam_t[3 * cam + k];
      }
      if (c.kd > 0) {
        const int np = pb->camera_model == 1 ? 4 : 3;
        for (int a = 0; a < NI; ++a) {
          if (pb->intr_const && pb->intr_const[a]) continue;
          for (int k = 0; k < np; ++k) xs += intr[4 * a + k] * intr[4 * a + k];
        }
      }
      for (int p = 0; p < P; ++p) {
        if (pb->pt_const && pb->pt_const[p]) continue;
        for (int k = 0; k < 3; ++k) xs += pts[3 * p + k] * pts[3 * p + k];
      }
      const double tol = opt->parameter_tolerance * (sqrt(xs) + opt->parameter_tolerance);
      if (!(step_norm > tol)) { sum->termination = BAO_CONVERGENCE_PARAMETER; if (log && it < log_cap) log[it] = li; sum->num_log = it + 1; break; }
    }
    /* FunctionToleranceReached */
    if (fabs(x_cost - cand_cost) <= opt->function_tolerance * x_cost) {
      sum->termination = BAO_CONVERGENCE_FUNCTION; if (log && it < log_cap) log[it] = li; sum->num_log = it + 1; break;
    }
    const double rel = (x_cost - cand_cost) / mcc;
    li.relative_decrease = rel;
    if (rel > opt->min_relative_decrease) {
      memcpy(cam_q, nq, sizeof(double) * 4 * C); memcpy(cam_t, nt, sizeof(double) * 3 * C);
      memcpy(intr, ni, sizeof(double) * 4 * NI); memcpy(pts, np_, sizeof(double) * 3 * P);
      x_cost = cand_cost;
      grad_and_colnorm(&c, cam_q, cam_t, intr, pts, grad, colsq);
      for (int i = 0; i < n; ++i) delta[i] = c.active[i] ? -grad[i] : 0.0;
      state_plus(&c, cam_q, cam_t, intr, pts, delta, nq, nt, ni, np_);
      ambient_diff_norm(&c, cam_q, cam_t, intr, pts, nq, nt, ni, np_, &gmax);
      radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rel - 1.0, 3));
      radius = fmin(opt->max_trust_region_radius, radius);
      decrease_factor = 2.0; reuse_diagonal = 0;
      step_successful = 1;
      ++sum->num_successful_steps;
      li.successful = 1; li.cost = x_cost; li.gradient_max_norm = gmax;
    } else {
      radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = 1;
      step_successful = 0;
      ++sum->num_unsuccessful_steps;
    }
    li.radius = radius;
    if (log && it < log_cap) log[it] = li;
    sum->num_log = it + 1;
  }
  sum->final_cost = x_cost;
  sum->num_iterations = it;
  sum->n_reduced = c.n_red;
  free(c.slice_begin); free(c.slice_lhs); free(c.slice_vec);
  free(c.active); free(c.scale); free(grad); free(colsq); free(diag); free(D); free(y); free(delta);
  free(lhs); free(rhs); free(nq); free(nt); free(ni); free(np_);
  return 0;
}

int bao_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* expose one residual/Jacobian evaluation for the finite-difference tests */
void bao_obs_eval(int model, const double* q, const double* t, const double* intr, const double* X, double u, double v,
                  double* r, double* Jp, double* Ji, double* Jx) {
  obs_eval(model, q, t, intr, X, u, v, r, Jp, Ji, Jx);
}
void bao_quat_plus(const double* x, const double* d, double* out) { quat_plus(x, d, out); }
