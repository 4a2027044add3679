// Per-observation reprojection residual, analytic Jacobians, robust-loss corrector and the
// quaternion manifold -- the arithmetic every BA / pose-refinement kernel shares.
//
// Semantics follow what the reference obtains from pycolmap==3.10 / Ceres (reference call sites
// vggsfm/utils/triangulation.py:213,387,590,1050,1142): residual = ImgFromCam(params, R(q) X + t) - uv
// for SIMPLE_PINHOLE (f,cx,cy) and SIMPLE_RADIAL (f,cx,cy,k); quaternion (x,y,z,w) updated on the
// manifold q <- exp(delta) * q; robust losses applied through Ceres' corrector.  Derivation of the
// Jacobians: SURVEY.md Appendix A.
#pragma once
#include "common.hpp"

namespace vgg {

enum : int { kPinhole = 0, kSimpleRadial = 1 };
enum : int { kLossTrivial = 0, kLossCauchy = 1, kLossHuber = 2, kLossSoftL1 = 3 };

struct Pose {  // one camera: unit quaternion (x,y,z,w) + translation
  double q[4];
  double t[3];
};

__device__ __forceinline__ void quat_rotate(const double* q, const double* v, double* out) {
  double ux = q[1] * v[2] - q[2] * v[1], uy = q[2] * v[0] - q[0] * v[2], uz = q[0] * v[1] - q[1] * v[0];
  ux += ux; uy += uy; uz += uz;
  out[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
  out[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
  out[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}

__device__ __forceinline__ void quat_to_R(const double* q, double* R) {
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
  R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

// x_plus = q(delta) * x, q(delta) = [sin|d|/|d| d, cos|d|]
__device__ __forceinline__ void quat_plus(const double* x, const double* d, double* out) {
  const double n = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (n == 0.0) { out[0] = x[0]; out[1] = x[1]; out[2] = x[2]; out[3] = x[3]; return; }
  const double s = sin(n) / n;
  const double qx = s * d[0], qy = s * d[1], qz = s * d[2], qw = cos(n);
  out[3] = qw * x[3] - qx * x[0] - qy * x[1] - qz * x[2];
  out[0] = qw * x[0] + qx * x[3] + qy * x[2] - qz * x[1];
  out[1] = qw * x[1] - qx * x[2] + qy * x[3] + qz * x[0];
  out[2] = qw * x[2] + qx * x[1] - qy * x[0] + qz * x[3];
}

// residual only
__device__ __forceinline__ void obs_residual(int model, const double* q, const double* t, const double* intr,
                                             const double* X, double u_obs, double v_obs, double* r) {
  double a[3];
  quat_rotate(q, X, a);
  const double Y0 = a[0] + t[0], Y1 = a[1] + t[1], Y2 = a[2] + t[2];
  const double k = (model == kSimpleRadial) ? intr[3] : 0.0;
  const double iz = 1.0 / Y2;
  const double u = Y0 * iz, v = Y1 * iz;
  const double d = 1.0 + k * (u * u + v * v);
  r[0] = intr[0] * (u * d) + intr[1] - u_obs;
  r[1] = intr[0] * (v * d) + intr[2] - v_obs;
}

// residual + Jacobians.  Jp 2x6 (rotation tangent delta(3), translation(3)), Ji 2x2 (f, k), Jx 2x3.
__device__ __forceinline__ void obs_eval(int model, const double* q, const double* t, const double* intr,
                                         const double* X, double u_obs, double v_obs, double* r, double* Jp,
                                         double* Ji, double* Jx) {
  double a[3];
  quat_rotate(q, X, a);
  const double Y0 = a[0] + t[0], Y1 = a[1] + t[1], Y2 = a[2] + t[2];
  const double f = intr[0];
  const double k = (model == kSimpleRadial) ? intr[3] : 0.0;
  const double iz = 1.0 / Y2;
  const double u = Y0 * iz, v = Y1 * iz;
  const double r2 = u * u + v * v;
  const double d = 1.0 + k * r2;
  r[0] = f * (u * d) + intr[1] - u_obs;
  r[1] = f * (v * d) + intr[2] - v_obs;
  const double xu = f * (d + 2 * k * u * u), xv = f * (2 * k * u * v), yv = f * (d + 2 * k * v * v);
  double JY[6];
  JY[0] = xu * iz; JY[1] = xv * iz; JY[2] = -(xu * u + xv * v) * iz;
  JY[3] = xv * iz; JY[4] = yv * iz; JY[5] = -(xv * u + yv * v) * iz;
  double R[9];
  quat_to_R(q, R);
#pragma unroll
  for (int row = 0; row < 2; ++row) {
    const double* j = JY + 3 * row;
    Jp[row * 6 + 0] = 2.0 * (j[2] * a[1] - j[1] * a[2]);
    Jp[row * 6 + 1] = 2.0 * (j[0] * a[2] - j[2] * a[0]);
    Jp[row * 6 + 2] = 2.0 * (j[1] * a[0] - j[0] * a[1]);
    Jp[row * 6 + 3] = j[0]; Jp[row * 6 + 4] = j[1]; Jp[row * 6 + 5] = j[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) Jx[row * 3 + c] = j[0] * R[c] + j[1] * R[3 + c] + j[2] * R[6 + c];
  }
  Ji[0] = u * d; Ji[1] = f * r2 * u;
  Ji[2] = v * d; Ji[3] = f * r2 * v;
}

// The same two with the rotation given as a matrix (row-major R[9], a = R X): the BA kernels keep R per camera (9 FMAs
// instead of the 18-instruction quaternion sandwich + quat_to_R per observation).
__device__ __forceinline__ void obs_residual_R(int model, const double* R, const double* t, const double* intr,
                                               const double* X, double u_obs, double v_obs, double* r) {
  const double Y0 = (R[0] * X[0] + R[1] * X[1] + R[2] * X[2]) + t[0];
  const double Y1 = (R[3] * X[0] + R[4] * X[1] + R[5] * X[2]) + t[1];
  const double Y2 = (R[6] * X[0] + R[7] * X[1] + R[8] * X[2]) + t[2];
  const double k = (model == kSimpleRadial) ? intr[3] : 0.0;
  const double iz = 1.0 / Y2;
  const double u = Y0 * iz, v = Y1 * iz;
  const double d = 1.0 + k * (u * u + v * v);
  r[0] = intr[0] * (u * d) + intr[1] - u_obs;
  r[1] = intr[0] * (v * d) + intr[2] - v_obs;
}

__device__ __forceinline__ void obs_eval_R(int model, const double* R, const double* t, const double* intr,
                                           const double* X, double u_obs, double v_obs, double* r, double* Jp,
                                           double* Ji, double* Jx) {
  double a[3];
  a[0] = R[0] * X[0] + R[1] * X[1] + R[2] * X[2];
  a[1] = R[3] * X[0] + R[4] * X[1] + R[5] * X[2];
  a[2] = R[6] * X[0] + R[7] * X[1] + R[8] * X[2];
  const double Y0 = a[0] + t[0], Y1 = a[1] + t[1], Y2 = a[2] + t[2];
  const double f = intr[0];
  const double k = (model == kSimpleRadial) ? intr[3] : 0.0;
  const double iz = 1.0 / Y2;
  const double u = Y0 * iz, v = Y1 * iz;
  const double r2 = u * u + v * v;
  const double d = 1.0 + k * r2;
  r[0] = f * (u * d) + intr[1] - u_obs;
  r[1] = f * (v * d) + intr[2] - v_obs;
  const double xu = f * (d + 2 * k * u * u), xv = f * (2 * k * u * v), yv = f * (d + 2 * k * v * v);
  double JY[6];
  JY[0] = xu * iz; JY[1] = xv * iz; JY[2] = -(xu * u + xv * v) * iz;
  JY[3] = xv * iz; JY[4] = yv * iz; JY[5] = -(xv * u + yv * v) * iz;
#pragma unroll
  for (int row = 0; row < 2; ++row) {
    const double* j = JY + 3 * row;
    Jp[row * 6 + 0] = 2.0 * (j[2] * a[1] - j[1] * a[2]);
    Jp[row * 6 + 1] = 2.0 * (j[0] * a[2] - j[2] * a[0]);
    Jp[row * 6 + 2] = 2.0 * (j[1] * a[0] - j[0] * a[1]);
    Jp[row * 6 + 3] = j[0]; Jp[row * 6 + 4] = j[1]; Jp[row * 6 + 5] = j[2];
#pragma unroll
    for (int c = 0; c < 3; ++c) Jx[row * 3 + c] = j[0] * R[c] + j[1] * R[3 + c] + j[2] * R[6 + c];
  }
  Ji[0] = u * d; Ji[1] = f * r2 * u;
  Ji[2] = v * d; Ji[3] = f * r2 * v;
}

// Ceres LossFunction::Evaluate
__device__ __forceinline__ void loss_eval(int loss, double a, double s, double* rho) {
  const double b = a * a;
  if (loss == kLossCauchy) {
    const double sum = 1.0 + s / b, inv = 1.0 / sum;
    rho[0] = b * log(sum); rho[1] = fmax(inv, 2.2250738585072014e-308); rho[2] = -(1.0 / b) * (inv * inv);
  } else if (loss == kLossHuber) {
    if (s > b) { const double r = sqrt(s); rho[0] = 2 * a * r - b; rho[1] = fmax(a / r, 2.2250738585072014e-308); rho[2] = -rho[1] / (2 * s); }
    else { rho[0] = s; rho[1] = 1; rho[2] = 0; }
  } else if (loss == kLossSoftL1) {
    const double sum = 1.0 + s / b, tmp = sqrt(sum);
    rho[0] = 2 * b * (tmp - 1); rho[1] = fmax(1 / tmp, 2.2250738585072014e-308); rho[2] = -1 / (2 * b * tmp * sum);
  } else {
    rho[0] = s; rho[1] = 1; rho[2] = 0;
  }
}

struct Corrector {
  double sqrt_rho1, residual_scaling, alpha_sq_norm;
  __device__ __forceinline__ Corrector(double sq_norm, const double* rho) {
    sqrt_rho1 = sqrt(rho[1]);
    if (sq_norm == 0.0 || rho[2] <= 0.0) { residual_scaling = sqrt_rho1; alpha_sq_norm = 0.0; return; }
    const double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
    const double alpha = 1.0 - sqrt(D);
    residual_scaling = sqrt_rho1 / (1 - alpha);
    alpha_sq_norm = alpha / sq_norm;
  }
  template <int NC>
  __device__ __forceinline__ void jac(const double* r, double* J) const {
    if (alpha_sq_norm == 0.0) {
#pragma unroll
      for (int i = 0; i < 2 * NC; ++i) J[i] *= sqrt_rho1;
      return;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const double rtj = r[0] * J[c] + r[1] * J[NC + c];
      J[c] = sqrt_rho1 * (J[c] - alpha_sq_norm * r[0] * rtj);
      J[NC + c] = sqrt_rho1 * (J[NC + c] - alpha_sq_norm * r[1] * rtj);
    }
  }
};

}  // namespace vgg
