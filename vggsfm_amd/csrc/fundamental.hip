// Two-view stage in front of the hot path: fundamental matrices by 7-point RANSAC + 8-point local optimisation for
// every (query frame, other frame) pair at once -- vggsfm/two_view_geo/fundamental.py:43-183 (estimate_fundamental),
// its helpers two_view_geo/utils.py:63-298, caller estimate_preliminary.py:103-152 (-> fmat_inlier_mask, the input
// of the Triangulator).  SURVEY.md section 8(f).3.
//
// PARITY PARTLY PINNED (oracle/fundamental.py header: Sampson distance, 8-point fit and winner selection are checked
// against the reference's own functions; the 7-point solver and the RNG are not); this file and the oracle agree bit for bit: float64, no FMA contraction
// (-ffp-contract=off), no transcendental functions, every sum in a fixed order.
//
//   fmat7_kernel        one thread per (pair, 7-point sample): normalise, null space of the 7x9 system by Gauss-Jordan
//                       with complete pivoting, cubic det(F1 + lambda F2) = 0 by bisection + deflation, <= 3 matrices
//   fmat_score_kernel   one wavefront per (pair, hypothesis): squared Sampson distance of all N matches in one sweep
//                       (coalesced, the pair's points stay in L2 for its K hypotheses), inlier count + residual sum
//   fmat8_kernel        one workgroup per (pair, selected hypothesis): its inliers are recomputed on the fly, three
//                       sweeps (means, scales, the 45 sums of X^T X) with fixed-order reductions, then one wavefront
//                       runs cyclic Jacobi on the 9x9 matrix (lane = row), rank 2, denormalise
//   fmat_residuals_kernel  residuals of the winner (B x N)
// Selection of the lo_num best hypotheses (a stable sort of K integers per pair) is left to the caller.
#include "common.hpp"
#include "../../include/vggsfm_amd.h"

namespace vgg {

constexpr int kBisections = 110;     // BISECTIONS in oracle/fundamental.py
constexpr int kSweeps9 = 10, kSweeps3 = 8;
constexpr double kBig = 1e6;         // residual of an invalid match

// ------------------------------------------------------------------------------------------------ small solvers
// real roots of c3 x^3 + c2 x^2 + c1 x + c0 (oracle: cubic_real_roots)
__device__ inline void cubic_real_roots(double c3, double c2, double c1, double c0, double r[3], bool ok[3]) {
  const double mx = fabs(c3) + fabs(c2) + fabs(c1) + fabs(c0);
  const bool is_cubic = fabs(c3) > 1e-12 * mx;
  if (is_cubic) {
    const double a = c2 / c3, b = c1 / c3, c = c0 / c3;
    const double B = 1.0 + fmax(fabs(a), fmax(fabs(b), fabs(c)));
    double lo = -B, hi = B;
    for (int it = 0; it < kBisections; ++it) {
      const double mid = 0.5 * (lo + hi);
      const double g = ((mid + a) * mid + b) * mid + c;
      if (g > 0.0) hi = mid; else lo = mid;
    }
    const double x = 0.5 * (lo + hi);
    const bool back = fabs(x * x * x) > fabs(c);
    double qb, qc;
    if (back) { qc = -c / x; qb = (qc - b) / x; }
    else { qb = a + x; qc = b + qb * x; }
    const double disc = qb * qb - 4.0 * qc;
    const bool okq = disc >= 0.0;
    const double sq = sqrt(okq ? disc : 0.0);
    r[0] = x; r[1] = 0.5 * (-qb + sq); r[2] = 0.5 * (-qb - sq);
    ok[0] = true; ok[1] = okq; ok[2] = okq;
  } else {
    const bool is_quad = fabs(c2) > 1e-12 * mx;
    const double dq = c1 * c1 - 4.0 * c2 * c0;
    const bool okd = dq >= 0.0;
    const double sqd = sqrt(okd ? dq : 0.0);
    const bool is_lin = !is_quad && fabs(c1) > 1e-12 * mx;
    r[0] = is_quad ? (-c1 + sqd) / (2.0 * c2) : -c0 / c1;
    r[1] = (-c1 - sqd) / (2.0 * c2);
    r[2] = 0.0;                          // (slot unused)
    ok[0] = (is_quad && okd) || is_lin; ok[1] = is_quad && okd; ok[2] = false;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double x = r[i];
    const double f = ((c3 * x + c2) * x + c1) * x + c0;
    const double df = (3.0 * c3 * x + 2.0 * c2) * x + c1;
    const double cand = x - ((df != 0.0) ? f / df : 0.0);
    const double fc = ((c3 * cand + c2) * cand + c1) * cand + c0;
    const double xr = (fabs(fc) < fabs(f)) ? cand : x;
    r[i] = xr;
    ok[i] = ok[i] && isfinite(xr);
  }
}

__device__ inline double det3(const double* F) {
  return (F[0] * (F[4] * F[8] - F[5] * F[7]) - F[1] * (F[3] * F[8] - F[5] * F[6])) + F[2] * (F[3] * F[7] - F[4] * F[6]);
}
// C = A B (3x3 row-major), terms added left to right
__device__ inline void mat3(const double* A, const double* B, double* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = (A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j]) + A[3 * i + 2] * B[6 + j];
}
// F = T2^T Fh T1 for T = [[s,0,a],[0,s,b],[0,0,1]] given as full 3x3, then unit Frobenius norm; returns the norm
__device__ inline double denormalize_unit(const double* Fh, const double* T1, const double* T2, double* F) {
  double tmp[9], T2t[9];
  mat3(Fh, T1, tmp);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) T2t[3 * i + j] = T2[3 * j + i];
  mat3(T2t, tmp, F);
  double n2 = F[0] * F[0];
#pragma unroll
  for (int i = 1; i < 9; ++i) n2 = n2 + F[i] * F[i];
  const double n = sqrt(n2);
#pragma unroll
  for (int i = 0; i < 9; ++i) F[i] = F[i] / n;
  return n;
}

__device__ inline void jacobi_cs(double app, double aqq, double apq, double& c, double& s) {
  const bool rot = apq != 0.0;
  const double tau = (aqq - app) / (2.0 * (rot ? apq : 1.0));
  const double t = ((tau >= 0.0) ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
  const double cc = 1.0 / sqrt(1.0 + t * t);
  c = rot ? cc : 1.0;
  s = rot ? t * cc : 0.0;
}

// cyclic Jacobi of a symmetric 3x3 (scalar): eigenvector of the smallest eigenvalue (first minimum)
__device__ inline void smallest_eigvec3(const double* G, double* v3) {
  double A[3][3], V[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) { A[i][j] = G[3 * i + j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
  for (int sw = 0; sw < kSweeps3; ++sw) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int q = p + 1; q < 3; ++q) {
        double c, s;
        jacobi_cs(A[p][p], A[q][q], A[p][q], c, s);
#pragma unroll
        for (int i = 0; i < 3; ++i) { const double cp = A[i][p], cq = A[i][q]; A[i][p] = c * cp - s * cq; A[i][q] = s * cp + c * cq; }
#pragma unroll
        for (int j = 0; j < 3; ++j) { const double rp = A[p][j], rq = A[q][j]; A[p][j] = c * rp - s * rq; A[q][j] = s * rp + c * rq; }
#pragma unroll
        for (int i = 0; i < 3; ++i) { const double vp = V[i][p], vq = V[i][q]; V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq; }
      }
    }
  }
  int jm = 0;
  double dm = A[0][0];
  if (A[1][1] < dm) { dm = A[1][1]; jm = 1; }
  if (A[2][2] < dm) { dm = A[2][2]; jm = 2; }
#pragma unroll
  for (int i = 0; i < 3; ++i) v3[i] = (jm == 0) ? V[i][0] : ((jm == 1) ? V[i][1] : V[i][2]);
}

// ------------------------------------------------------------------------------------------------ 7-point
__global__ __launch_bounds__(64) void fmat7_kernel(const double* __restrict__ pts1, const double* __restrict__ pts2,
                                                  const int32_t* __restrict__ samples, int B, int N, int H,
                                                  double* __restrict__ outF, uint8_t* __restrict__ out_valid) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (h >= H) return;
  const int32_t* smp = samples + (size_t)h * 7;
  double x1[7], y1[7], x2[7], y2[7];
  bool idx_ok = true;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int id = smp[i];
    idx_ok = idx_ok && id >= 0 && id < N;
    const size_t o = ((size_t)b * N + (idx_ok ? id : 0)) * 2;
    x1[i] = pts1[o]; y1[i] = pts1[o + 1]; x2[i] = pts2[o]; y2[i] = pts2[o + 1];
  }
  double T1[9], T2[9];
  auto normalize = [&](double* x, double* y, double* T) {
    double sx = x[0], sy = y[0];
#pragma unroll
    for (int i = 1; i < 7; ++i) { sx = sx + x[i]; sy = sy + y[i]; }
    const double mx = sx / 7.0, my = sy / 7.0;
    double sd = 0.0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const double dx = x[i] - mx, dy = y[i] - my;
      const double d = sqrt(dx * dx + dy * dy);
      sd = (i == 0) ? d : sd + d;
    }
    const double s = sqrt(2.0) / (sd / 7.0 + 1e-8);
    T[0] = s; T[1] = 0.0; T[2] = -s * mx; T[3] = 0.0; T[4] = s; T[5] = -s * my; T[6] = 0.0; T[7] = 0.0; T[8] = 1.0;
#pragma unroll
    for (int i = 0; i < 7; ++i) { x[i] = s * x[i] + T[2]; y[i] = s * y[i] + T[5]; }
  };
  normalize(x1, y1, T1);
  normalize(x2, y2, T2);
  double A[7][9];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    A[i][0] = x2[i] * x1[i]; A[i][1] = x2[i] * y1[i]; A[i][2] = x2[i];
    A[i][3] = y2[i] * x1[i]; A[i][4] = y2[i] * y1[i]; A[i][5] = y2[i];
    A[i][6] = x1[i]; A[i][7] = y1[i]; A[i][8] = 1.0;
  }
  // Gauss-Jordan with complete pivoting (first maximum in row-major order of the remaining block)
  int perm[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) perm[j] = j;
  bool ok = idx_ok;
  for (int k = 0; k < 7; ++k) {
    int pr = k, pc = k;
    double best = -1.0;
    for (int i = k; i < 7; ++i)
      for (int j = k; j < 9; ++j) {
        const double v = fabs(A[i][j]);
        if (v > best) { best = v; pr = i; pc = j; }
      }
    for (int j = 0; j < 9; ++j) { const double t = A[k][j]; A[k][j] = A[pr][j]; A[pr][j] = t; }
    for (int i = 0; i < 7; ++i) { const double t = A[i][k]; A[i][k] = A[i][pc]; A[i][pc] = t; }
    { const int t = perm[k]; perm[k] = perm[pc]; perm[pc] = t; }
    const double piv = A[k][k];
    ok = ok && piv != 0.0;
    const double pd = (piv != 0.0) ? piv : 1.0;
    for (int j = 0; j < 9; ++j) A[k][j] = A[k][j] / pd;
    for (int i = 0; i < 7; ++i) {
      if (i == k) continue;
      const double f = A[i][k];
      for (int j = 0; j < 9; ++j) A[i][j] = A[i][j] - f * A[k][j];
    }
  }
  double F1[9], F2[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) { F1[j] = 0.0; F2[j] = 0.0; }
  for (int j = 0; j < 9; ++j) {
    if (perm[7] == j) F1[j] = 1.0;
    if (perm[8] == j) F2[j] = 1.0;
  }
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j < 9; ++j)
      if (perm[i] == j) { F1[j] = -A[i][7]; F2[j] = -A[i][8]; }
  double Fp[9], Fm[9];
#pragma unroll
  for (int j = 0; j < 9; ++j) { Fp[j] = F1[j] + F2[j]; Fm[j] = F1[j] - F2[j]; }
  const double c0 = det3(F1), c3 = det3(F2), dp = det3(Fp), dm = det3(Fm);
  const double c2 = 0.5 * (dp + dm) - c0;
  const double c1 = 0.5 * (dp - dm) - c3;
  double lam[3];
  bool okr[3];
  cubic_real_roots(c3, c2, c1, c0, lam, okr);
  double* out = outF + ((size_t)b * H + h) * 27;
  uint8_t* vout = out_valid + ((size_t)b * H + h) * 3;
  for (int s = 0; s < 3; ++s) {
    double Fh[9], F[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) Fh[j] = F1[j] + lam[s] * F2[j];
    const double n = denormalize_unit(Fh, T1, T2, F);
    bool v = okr[s] && ok && n > 0.0;
#pragma unroll
    for (int j = 0; j < 9; ++j) v = v && isfinite(F[j]);
#pragma unroll
    for (int j = 0; j < 9; ++j) out[9 * s + j] = v ? F[j] : 0.0;
    vout[s] = v ? 1 : 0;
  }
}

// ------------------------------------------------------------------------------------------------ residuals
__device__ inline double sampson_sq(const double* __restrict__ F, double u1, double v1, double u2, double v2) {
  const double l0 = (F[0] * u1 + F[1] * v1) + F[2];
  const double l1 = (F[3] * u1 + F[4] * v1) + F[5];
  const double l2 = (F[6] * u1 + F[7] * v1) + F[8];
  const double m0 = (F[0] * u2 + F[3] * v2) + F[6];
  const double m1 = (F[1] * u2 + F[4] * v2) + F[7];
  const double num = (u2 * l0 + v2 * l1) + l2;
  const double den = (l0 * l0 + l1 * l1) + (m0 * m0 + m1 * m1);
  const double r = (num * num) / den;
  return isfinite(r) ? r : kBig;
}

constexpr int kHypPerWave = 4;   // hypotheses scored per sweep of a wavefront: the points are read once for all of them

__global__ __launch_bounds__(256) void fmat_score_kernel(const double* __restrict__ pts1, const double* __restrict__ pts2,
                                                        const uint8_t* __restrict__ vmask, const double* __restrict__ Fall,
                                                        const uint8_t* __restrict__ fvalid, int B, int N, int K, double thr_sq,
                                                        int32_t* __restrict__ counts, double* __restrict__ rsums) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int k0 = (blockIdx.x * 4 + wave) * kHypPerWave, b = blockIdx.y;
  if (k0 >= K) return;
  double F[kHypPerWave][9];
  bool live[kHypPerWave];
  bool any = false;
#pragma unroll
  for (int h = 0; h < kHypPerWave; ++h) {
    const int k = k0 + h;
    live[h] = k < K && fvalid[(size_t)b * K + (k < K ? k : 0)];
    any = any || live[h];
#pragma unroll
    for (int i = 0; i < 9; ++i) F[h][i] = live[h] ? Fall[((size_t)b * K + k) * 9 + i] : 0.0;
  }
  const double* p1 = pts1 + (size_t)b * N * 2;
  const double* p2 = pts2 + (size_t)b * N * 2;
  const uint8_t* vm = vmask ? vmask + (size_t)b * N : nullptr;
  int c[kHypPerWave];
  double s[kHypPerWave];
#pragma unroll
  for (int h = 0; h < kHypPerWave; ++h) { c[h] = 0; s[h] = 0.0; }
  if (any) {
    for (int n = lane; n < N; n += 64) {
      const double u1 = p1[2 * n], v1 = p1[2 * n + 1], u2 = p2[2 * n], v2 = p2[2 * n + 1];
      const bool usable = !vm || vm[n];
#pragma unroll
      for (int h = 0; h < kHypPerWave; ++h) {
        const double r = sampson_sq(F[h], u1, v1, u2, v2);
        const bool in = r <= thr_sq && usable;
        c[h] += in ? 1 : 0;
        s[h] = s[h] + (in ? r : 0.0);
      }
    }
  }
#pragma unroll
  for (int h = 0; h < kHypPerWave; ++h) {
    const int ch = wave_sum_i(c[h]);
    const double sh = wave_sum(s[h]);
    if (lane == 0 && k0 + h < K) {
      counts[(size_t)b * K + k0 + h] = live[h] ? ch : -1;
      rsums[(size_t)b * K + k0 + h] = live[h] ? sh : 0.0;
    }
  }
}

__global__ __launch_bounds__(256) void fmat_residuals_kernel(const double* __restrict__ pts1, const double* __restrict__ pts2,
                                                            const uint8_t* __restrict__ vmask, const double* __restrict__ Fb,
                                                            int B, int N, double* __restrict__ out) {
  const int n = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (n >= N) return;
  double F[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) F[i] = Fb[(size_t)b * 9 + i];
  const size_t o = ((size_t)b * N + n) * 2;
  const double r = sampson_sq(F, pts1[o], pts1[o + 1], pts2[o], pts2[o + 1]);
  out[(size_t)b * N + n] = (!vmask || vmask[(size_t)b * N + n]) ? r : kBig;
}

// ------------------------------------------------------------------------------------------------ 8-point
// fixed-order reduction of Q quantities over the 256 threads: halving tree (t, t + 128), (t, t + 64), ...
template <int Q>
__device__ inline void block_tree_sum(double (*red)[256], const double* val, double* out) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int q = 0; q < Q; ++q) red[q][tid] = val[q];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) {
#pragma unroll
      for (int q = 0; q < Q; ++q) red[q][tid] = red[q][tid] + red[q][tid + st];
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) out[q] = red[q][0];
  __syncthreads();
}

__global__ __launch_bounds__(256) void fmat8_kernel(const double* __restrict__ pts1, const double* __restrict__ pts2,
                                                   const uint8_t* __restrict__ vmask, const double* __restrict__ Fsrc,
                                                   const int32_t* __restrict__ src_counts, const int32_t* __restrict__ sel,
                                                   int B, int N, int Ksrc, int L, double thr_sq, double* __restrict__ outF,
                                                   uint8_t* __restrict__ out_valid) {
  __shared__ double red[9][256];
  __shared__ double Ms[9][9];
  __shared__ double fvec[9];
  const int tid = threadIdx.x, l = blockIdx.x, b = blockIdx.y;
  const int src = sel[(size_t)b * L + l];
  const bool src_ok = src >= 0 && src < Ksrc && src_counts[(size_t)b * Ksrc + (src >= 0 && src < Ksrc ? src : 0)] >= 0;
  double Fs[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Fs[i] = src_ok ? Fsrc[((size_t)b * Ksrc + src) * 9 + i] : 0.0;
  const double* p1 = pts1 + (size_t)b * N * 2;
  const double* p2 = pts2 + (size_t)b * N * 2;
  const uint8_t* vm = vmask ? vmask + (size_t)b * N : nullptr;
  auto inlier = [&](int n, double& u1, double& v1, double& u2, double& v2) -> bool {
    u1 = p1[2 * n]; v1 = p1[2 * n + 1]; u2 = p2[2 * n]; v2 = p2[2 * n + 1];
    const double r = sampson_sq(Fs, u1, v1, u2, v2);
    return src_ok && r <= thr_sq && (!vm || vm[n]);
  };
  // sweep 1: count and coordinate sums
  double v5[5] = {0, 0, 0, 0, 0}, o5[5];
  for (int n = tid; n < N; n += 256) {
    double u1, v1, u2, v2;
    const bool in = inlier(n, u1, v1, u2, v2);
    v5[0] = v5[0] + (in ? 1.0 : 0.0);
    v5[1] = v5[1] + (in ? u1 : 0.0); v5[2] = v5[2] + (in ? v1 : 0.0);
    v5[3] = v5[3] + (in ? u2 : 0.0); v5[4] = v5[4] + (in ? v2 : 0.0);
  }
  block_tree_sum<5>(red, v5, o5);
  const double cnt = o5[0];
  const double mx1 = o5[1] / (cnt + 1e-8), my1 = o5[2] / (cnt + 1e-8), mx2 = o5[3] / (cnt + 1e-8), my2 = o5[4] / (cnt + 1e-8);
  // sweep 2: mean distances to the means
  double v2s[2] = {0, 0}, o2[2];
  for (int n = tid; n < N; n += 256) {
    double u1, v1, u2, v2;
    const bool in = inlier(n, u1, v1, u2, v2);
    const double dx1 = u1 - mx1, dy1 = v1 - my1, dx2 = u2 - mx2, dy2 = v2 - my2;
    v2s[0] = v2s[0] + (in ? sqrt(dx1 * dx1 + dy1 * dy1) : 0.0);
    v2s[1] = v2s[1] + (in ? sqrt(dx2 * dx2 + dy2 * dy2) : 0.0);
  }
  block_tree_sum<2>(red, v2s, o2);
  const double s1 = sqrt(2.0) / (o2[0] / (cnt + 1e-8) + 1e-8), s2 = sqrt(2.0) / (o2[1] / (cnt + 1e-8) + 1e-8);
  const double T1[9] = {s1, 0.0, -s1 * mx1, 0.0, s1, -s1 * my1, 0.0, 0.0, 1.0};
  const double T2[9] = {s2, 0.0, -s2 * mx2, 0.0, s2, -s2 * my2, 0.0, 0.0, 1.0};
  // sweep 3: the upper triangle of X^T X (45 sums per thread)
  double acc[45];
#pragma unroll
  for (int i = 0; i < 45; ++i) acc[i] = 0.0;
  for (int n = tid; n < N; n += 256) {
    double u1, v1, u2, v2;
    const bool in = inlier(n, u1, v1, u2, v2);
    const double a1 = s1 * u1 + T1[2], b1 = s1 * v1 + T1[5], a2 = s2 * u2 + T2[2], b2 = s2 * v2 + T2[5];
    const double m = in ? 1.0 : 0.0;
    const double row[9] = {(a2 * a1) * m, (a2 * b1) * m, a2 * m, (b2 * a1) * m, (b2 * b1) * m, b2 * m, a1 * m, b1 * m, 1.0 * m};
    int e = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
      for (int j = i; j < 9; ++j) { acc[e] = acc[e] + row[i] * row[j]; ++e; }
  }
  {
    int e = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      double vals[9], outs[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) vals[j] = (j >= i) ? acc[e + (j - i)] : 0.0;
      block_tree_sum<9>(red, vals, outs);
      if (tid == 0) {
#pragma unroll
        for (int j = 0; j < 9; ++j)
          if (j >= i) { Ms[i][j] = outs[j]; Ms[j][i] = outs[j]; }
      }
      e += 9 - i;
    }
  }
  __syncthreads();
  // cyclic Jacobi on the 9x9 matrix by one wavefront: lane = row of A and of V
  if (tid < 64) {
    const int lane = tid;
    const int rowi = lane < 9 ? lane : 0;
    double a[9], v[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) { a[j] = Ms[rowi][j]; v[j] = (j == rowi) ? 1.0 : 0.0; }
    for (int sw = 0; sw < kSweeps9; ++sw) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
#pragma unroll
        for (int q = p + 1; q < 9; ++q) {
          const double app = __shfl(a[p], p, 64), aqq = __shfl(a[q], q, 64), apq = __shfl(a[q], p, 64);
          double c, s;
          jacobi_cs(app, aqq, apq, c, s);
          { const double cp = a[p], cq = a[q]; a[p] = c * cp - s * cq; a[q] = s * cp + c * cq; }      // columns p, q
          double rp[9], rq[9];
#pragma unroll
          for (int j = 0; j < 9; ++j) { rp[j] = __shfl(a[j], p, 64); rq[j] = __shfl(a[j], q, 64); }
          if (lane == p) {
#pragma unroll
            for (int j = 0; j < 9; ++j) a[j] = c * rp[j] - s * rq[j];
          } else if (lane == q) {
#pragma unroll
            for (int j = 0; j < 9; ++j) a[j] = s * rp[j] + c * rq[j];
          }
          { const double vp = v[p], vq = v[q]; v[p] = c * vp - s * vq; v[q] = s * vp + c * vq; }      // V columns
        }
      }
    }
    // first minimum of the diagonal -> its eigenvector (column of V)
    double dmin = __shfl(a[0], 0, 64);
    int jm = 0;
#pragma unroll
    for (int j = 1; j < 9; ++j) {
      const double dj = __shfl(a[j], j, 64);
      if (dj < dmin) { dmin = dj; jm = j; }
    }
    double comp = v[0];
#pragma unroll
    for (int j = 1; j < 9; ++j) comp = (jm == j) ? v[j] : comp;
    if (lane < 9) fvec[lane] = comp;
  }
  __syncthreads();
  if (tid == 0) {
    double Fh[9], Ft[9], G[9], v3[3], Fv[3], F[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Fh[i] = fvec[i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Ft[3 * i + j] = Fh[3 * j + i];
    mat3(Ft, Fh, G);
    smallest_eigvec3(G, v3);
#pragma unroll
    for (int i = 0; i < 3; ++i) Fv[i] = (Fh[3 * i] * v3[0] + Fh[3 * i + 1] * v3[1]) + Fh[3 * i + 2] * v3[2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Fh[3 * i + j] = Fh[3 * i + j] - Fv[i] * v3[j];
    const double nrm = denormalize_unit(Fh, T1, T2, F);
    bool ok = cnt >= 8.0 && nrm > 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) ok = ok && isfinite(F[i]);
    double* o = outF + ((size_t)b * L + l) * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) o[i] = ok ? F[i] : 0.0;
    out_valid[(size_t)b * L + l] = ok ? 1 : 0;
  }
}

}  // namespace vgg

using namespace vgg;

extern "C" {

int vgg_fmat_seven_point(const double* points1, const double* points2, const int32_t* samples, int num_pairs, int num_points,
                         int num_samples, double* out_fmat, uint8_t* out_valid, void* stream) {
  if (num_pairs < 0 || num_points < 7 || num_samples <= 0) return VGG_ERR_INVALID_ARGUMENT;
  if (num_pairs == 0) return VGG_OK;
  if (!points1 || !points2 || !samples || !out_fmat || !out_valid) return VGG_ERR_INVALID_ARGUMENT;
  if (num_pairs > 65535) return VGG_ERR_UNSUPPORTED;
  fmat7_kernel<<<dim3(div_up(num_samples, 64), num_pairs), 64, 0, (hipStream_t)stream>>>(points1, points2, samples, num_pairs,
                                                                                        num_points, num_samples, out_fmat, out_valid);
  VGG_LAUNCH_CHECK();
  return VGG_OK;
}

int vgg_fmat_score(const double* points1, const double* points2, const uint8_t* valid_mask, const double* fmat,
                   const uint8_t* fmat_valid, int num_pairs, int num_points, int num_hypotheses, double max_error_sq,
                   int32_t* out_counts, double* out_residual_sums, void* stream) {
  if (num_pairs < 0 || num_points <= 0 || num_hypotheses <= 0) return VGG_ERR_INVALID_ARGUMENT;
  if (num_pairs == 0) return VGG_OK;
  if (!points1 || !points2 || !fmat || !fmat_valid || !out_counts || !out_residual_sums) return VGG_ERR_INVALID_ARGUMENT;
  if (num_pairs > 65535) return VGG_ERR_UNSUPPORTED;
  fmat_score_kernel<<<dim3(div_up(num_hypotheses, 4 * kHypPerWave), num_pairs), 256, 0, (hipStream_t)stream>>>(
      points1, points2, valid_mask, fmat, fmat_valid, num_pairs, num_points, num_hypotheses, max_error_sq, out_counts, out_residual_sums);
  VGG_LAUNCH_CHECK();
  return VGG_OK;
}

int vgg_fmat_eight_point(const double* points1, const double* points2, const uint8_t* valid_mask, const double* src_fmat,
                         const int32_t* src_counts, const int32_t* selected, int num_pairs, int num_points, int num_src,
                         int num_selected, double max_error_sq, double* out_fmat, uint8_t* out_valid, void* stream) {
  if (num_pairs < 0 || num_points <= 0 || num_src <= 0 || num_selected <= 0) return VGG_ERR_INVALID_ARGUMENT;
  if (num_pairs == 0) return VGG_OK;
  if (!points1 || !points2 || !src_fmat || !src_counts || !selected || !out_fmat || !out_valid) return VGG_ERR_INVALID_ARGUMENT;
  if (num_pairs > 65535) return VGG_ERR_UNSUPPORTED;
  fmat8_kernel<<<dim3(num_selected, num_pairs), 256, 0, (hipStream_t)stream>>>(points1, points2, valid_mask, src_fmat, src_counts,
                                                                               selected, num_pairs, num_points, num_src, num_selected,
                                                                               max_error_sq, out_fmat, out_valid);
  VGG_LAUNCH_CHECK();
  return VGG_OK;
}

int vgg_fmat_residuals(const double* points1, const double* points2, const uint8_t* valid_mask, const double* fmat, int num_pairs,
                       int num_points, double* out_residuals, void* stream) {
  if (num_pairs < 0 || num_points <= 0) return VGG_ERR_INVALID_ARGUMENT;
  if (num_pairs == 0) return VGG_OK;
  if (!points1 || !points2 || !fmat || !out_residuals) return VGG_ERR_INVALID_ARGUMENT;
  if (num_pairs > 65535) return VGG_ERR_UNSUPPORTED;
  fmat_residuals_kernel<<<dim3(div_up(num_points, 256), num_pairs), 256, 0, (hipStream_t)stream>>>(points1, points2, valid_mask, fmat,
                                                                                                   num_pairs, num_points, out_residuals);
  VGG_LAUNCH_CHECK();
  return VGG_OK;
}

}  // extern "C"
