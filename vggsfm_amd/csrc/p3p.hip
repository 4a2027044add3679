// Absolute pose from 2D-3D matches by P3P RANSAC on the device -- the role pycolmap.absolute_pose_estimation plays
// for the reference (vggsfm/utils/triangulation.py:324-326,400-432: fallback of refine_pose when a frame has <= 100
// inliers or a wild focal length; vggsfm/runners/video_runner.py:987-998: align_next_window(use_pnp=True)).
// pycolmap (COLMAP 3.10, EstimateAbsolutePose) is a third-party native dependency, absent from the reference tree:
// restated from its published behaviour -- minimal P3P samples, squared reprojection error on the normalised image
// plane, points behind the camera never inliers, support = (most inliers, then smallest inlier residual sum) -- with
// two documented deviations: a fixed number of hypotheses drawn by the caller (no adaptive trial count; COLMAP's own
// RNG cannot be reproduced anyway) and no EPnP local optimisation inside the loop (the callers run the non-linear
// pose refinement on the RANSAC inliers afterwards, as COLMAP does).  oracle/p3p.py mirrors this file operation by
// operation (this file is compiled with -ffp-contract=off), so the two agree bit for bit on the same samples.
//
// Minimal solver: ratios u = d2/d1, v = d3/d1 of the three depths; the two distance-ratio equations are quadratics
// in u whose coefficients are polynomials in v; their resultant is a quartic in v; u follows linearly from the
// common root; the pose from the orthonormal frames of the two triangles.
//
// Kernels (all frames of a call run concurrently; a "frame" may be a virtual one = (frame, focal length factor)):
//   p3p_hypotheses_kernel  one thread per (frame, sample): <= 4 poses
//   p3p_score_kernel       one wavefront per (frame, sample): its 4 poses scored against all N points in one sweep
//                          (points read once per wavefront, coalesced; counts / residual sums by wave reduction)
//   p3p_select_kernel      one workgroup per frame: arg-best over the 4H scores, then the inlier mask of the winner
#include "common.hpp"
#include "../../include/vggsfm_amd.h"

namespace vgg {

constexpr int kNewtonIters = 80;   // NEWTON_ITERS in oracle/p3p.py

// Real roots of k4 v^4 + k3 v^3 + k2 v^2 + k1 v + k0 (Ferrari; oracle/p3p.py solve_quartic).
__device__ inline void solve_quartic(double k4, double k3, double k2, double k1, double k0, double v[4], bool ok[4]) {
  const double b = k3 / k4, c = k2 / k4, d = k1 / k4, e = k0 / k4;
  const double b2 = b * b;
  const double p = c - 0.375 * b2;
  const double q = d - 0.5 * b * c + 0.125 * b2 * b;
  const double r = e - 0.25 * b * d + 0.0625 * b2 * c - (3.0 / 256.0) * b2 * b2;
  const double c1 = 0.25 * p * p - r;
  const double c0 = -0.125 * q * q;
  double m = 1.0 + fmax(fabs(p), fmax(fabs(c1), fabs(c0)));
  for (int it = 0; it < kNewtonIters; ++it) {
    const double g = ((m + p) * m + c1) * m + c0;
    const double dg = (3.0 * m + 2.0 * p) * m + c1;
    const double step = (dg != 0.0) ? g / dg : 0.0;
    m = m - step;
  }
  const double scale = 1.0 + fabs(p) + fabs(m);
  const bool biq = !(m > 1e-14 * scale);
  double roots[4];
  if (!biq) {
    const double s = sqrt(2.0 * m);
    const double qs = q / s;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const double sg = j == 0 ? 1.0 : -1.0;
      double disc = -2.0 * p - 2.0 * m - sg * 2.0 * qs;
      if (disc < 0.0 && disc > -1e-10 * scale) disc = 0.0;
      const bool okd = disc >= 0.0;
      const double sq = sqrt(okd ? disc : 0.0);
      roots[2 * j] = 0.5 * (sg * s + sq);
      roots[2 * j + 1] = 0.5 * (sg * s - sq);
      ok[2 * j] = okd; ok[2 * j + 1] = okd;
    }
  } else {
    double dq = p * p - 4.0 * r;
    if (dq < 0.0 && dq > -1e-10 * scale * scale) dq = 0.0;
    const bool okq = dq >= 0.0;
    const double sq = sqrt(okq ? dq : 0.0);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const double sg = j == 0 ? 1.0 : -1.0;
      const double y2 = 0.5 * (-p + sg * sq);
      const bool oky = okq && y2 >= 0.0;
      const double y = sqrt(oky ? y2 : 0.0);
      roots[2 * j] = y; roots[2 * j + 1] = -y;
      ok[2 * j] = oky; ok[2 * j + 1] = oky;
    }
  }
  const bool lead_ok = fabs(k4) > 1e-14 * (fabs(k4) + fabs(k3) + fabs(k2) + fabs(k1) + fabs(k0));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    double x = roots[j] - 0.25 * b;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const double f = (((k4 * x + k3) * x + k2) * x + k1) * x + k0;
      const double df = ((4.0 * k4 * x + 3.0 * k3) * x + 2.0 * k2) * x + k1;
      x = x - ((df != 0.0) ? f / df : 0.0);
    }
    v[j] = x;
    ok[j] = ok[j] && lead_ok && isfinite(x);
  }
}

__device__ inline double dot3(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
__device__ inline void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}
// orthonormal frame of a triangle: e1 along P1->P2, e3 normal, e2 = e3 x e1; returns |e1 x (P3 - P1)|
__device__ inline double tri_frame(const double* P1, const double* P2, const double* P3, double* e1, double* e2, double* e3) {
  double a[3] = {P2[0] - P1[0], P2[1] - P1[1], P2[2] - P1[2]};
  const double na = sqrt(dot3(a, a));
  e1[0] = a[0] / na; e1[1] = a[1] / na; e1[2] = a[2] / na;
  double w[3] = {P3[0] - P1[0], P3[1] - P1[1], P3[2] - P1[2]}, n[3];
  cross3(e1, w, n);
  const double nn = sqrt(dot3(n, n));
  e3[0] = n[0] / nn; e3[1] = n[1] / nn; e3[2] = n[2] / nn;
  cross3(e3, e1, e2);
  return nn;
}

// x [F][N][2] normalised image points, X [N][3], samples [F/group][H][3] (virtual frames of one real frame share them)
__global__ __launch_bounds__(128) void p3p_hypotheses_kernel(const double* __restrict__ x, const double* __restrict__ X,
                                                            const int32_t* __restrict__ samples, int F, int N, int H, int group,
                                                            double* __restrict__ poses, uint8_t* __restrict__ valid) {
  const int h = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
  if (h >= H) return;
  const int32_t* smp = samples + ((size_t)(f / group) * H + h) * 3;
  double bear[3][3], Xw[3][3];
  bool idx_ok = true;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int id = smp[i];
    idx_ok = idx_ok && id >= 0 && id < N;
    const int ic = idx_ok ? id : 0;
    const double u = x[((size_t)f * N + ic) * 2], w = x[((size_t)f * N + ic) * 2 + 1];
    const double nb = sqrt((u * u + w * w) + 1.0);
    bear[i][0] = u / nb; bear[i][1] = w / nb; bear[i][2] = 1.0 / nb;
    Xw[i][0] = X[3 * (size_t)ic]; Xw[i][1] = X[3 * (size_t)ic + 1]; Xw[i][2] = X[3 * (size_t)ic + 2];
  }
  double d12[3], d13[3], d23[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { d12[k] = Xw[0][k] - Xw[1][k]; d13[k] = Xw[0][k] - Xw[2][k]; d23[k] = Xw[1][k] - Xw[2][k]; }
  const double a12 = dot3(d12, d12), a13 = dot3(d13, d13), a23 = dot3(d23, d23);
  const double c12 = dot3(bear[0], bear[1]), c13 = dot3(bear[0], bear[2]), c23 = dot3(bear[1], bear[2]);
  const bool ok0 = idx_ok && a12 > 1e-20 && a13 > 1e-20 && a23 > 1e-20;
  const double inv = 1.0 / a12;
  const double A13 = a13 * inv, A23 = a23 * inv;
  const double A1 = A13;
  const double B1 = -2.0 * A13 * c12;
  const double c11 = 2.0 * c13, c10 = A13 - 1.0;
  const double A2 = A23 - 1.0;
  const double b21 = 2.0 * c23, b20 = -2.0 * A23 * c12;
  const double p2 = A2 - A1, p1 = -A2 * c11, p0 = A1 * A23 - A2 * c10;
  const double q1 = A1 * b21, q0 = A1 * b20 - A2 * B1;
  const double t3 = b21, t2 = -B1 - (b21 * c11 - b20), t1 = -(b21 * c10 + b20 * c11), t0 = B1 * A23 - b20 * c10;
  const double k4 = p2 * p2 - q1 * t3;
  const double k3 = 2.0 * p2 * p1 - (q1 * t2 + q0 * t3);
  const double k2 = 2.0 * p2 * p0 + p1 * p1 - (q1 * t1 + q0 * t2);
  const double k1 = 2.0 * p1 * p0 - (q1 * t0 + q0 * t1);
  const double k0 = p0 * p0 - q0 * t0;
  double v[4];
  bool okv[4];
  solve_quartic(k4, k3, k2, k1, k0, v, okv);
  double e1[3], e2[3], e3[3];
  const double nE = tri_frame(Xw[0], Xw[1], Xw[2], e1, e2, e3);
  double* out = poses + ((size_t)f * H + h) * 48;
  uint8_t* vout = valid + ((size_t)f * H + h) * 4;
  for (int s = 0; s < 4; ++s) {
    const double vs = v[s];
    const double Pv = (p2 * vs + p1) * vs + p0;
    const double Qv = q1 * vs + q0;
    const bool okq = fabs(Qv) > 1e-12 * (fabs(q1) + fabs(q0));
    const double u = -Pv / (okq ? Qv : 1.0);
    const double den = 1.0 + u * u - 2.0 * u * c12;
    // (u, v) must satisfy both quadratics (drops spurious roots of the resultant / of a clamped discriminant)
    const double C1v = (c11 - vs) * vs + c10;
    const double C2v = A23 - vs * vs;
    const double B2v = b21 * vs + b20;
    const double r1 = (A1 * u + B1) * u + C1v;
    const double r2 = (A2 * u + B2v) * u + C2v;
    const bool okr = (fabs(r1) + fabs(r2)) <= 1e-7 * (1.0 + u * u + vs * vs);
    bool ok = okv[s] && okq && okr && vs > 0.0 && u > 0.0 && den > 0.0 && ok0;
    const double dd1 = sqrt(a12 / (ok ? den : 1.0));
    const double dd2 = u * dd1, dd3 = vs * dd1;
    double Y1[3], Y2[3], Y3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { Y1[k] = dd1 * bear[0][k]; Y2[k] = dd2 * bear[1][k]; Y3[k] = dd3 * bear[2][k]; }
    double g1[3], g2[3], g3[3];
    const double nC = tri_frame(Y1, Y2, Y3, g1, g2, g3);
    double R[9], t[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) R[3 * i + j] = (g1[i] * e1[j] + g2[i] * e2[j]) + g3[i] * e3[j];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = Y1[i] - ((R[3 * i] * Xw[0][0] + R[3 * i + 1] * Xw[0][1]) + R[3 * i + 2] * Xw[0][2]);
    ok = ok && nE > 1e-12 && nC > 1e-12;
    bool fin = true;
#pragma unroll
    for (int i = 0; i < 9; ++i) fin = fin && isfinite(R[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) fin = fin && isfinite(t[i]);
    ok = ok && fin;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      out[12 * s + 4 * i] = ok ? R[3 * i] : 0.0; out[12 * s + 4 * i + 1] = ok ? R[3 * i + 1] : 0.0;
      out[12 * s + 4 * i + 2] = ok ? R[3 * i + 2] : 0.0; out[12 * s + 4 * i + 3] = ok ? t[i] : 0.0;
    }
    vout[s] = ok ? 1 : 0;
  }
}

__device__ inline bool point_error(const double* __restrict__ P, double X0, double X1, double X2, double u, double w, double thr_sq,
                                   double& e) {
  const double px = ((P[0] * X0 + P[1] * X1) + P[2] * X2) + P[3];
  const double py = ((P[4] * X0 + P[5] * X1) + P[6] * X2) + P[7];
  const double pz = ((P[8] * X0 + P[9] * X1) + P[10] * X2) + P[11];
  const bool front = pz > 1e-12;
  const double zs = front ? pz : 1.0;
  const double ex = px / zs - u, ey = py / zs - w;
  e = ex * ex + ey * ey;
  return front && e <= thr_sq;
}

// one wavefront per (frame, sample); counts [F][4H] (-1 for an invalid pose), sums [F][4H]
__global__ __launch_bounds__(256) void p3p_score_kernel(const double* __restrict__ x, const double* __restrict__ X,
                                                       const uint8_t* __restrict__ mask, int F, int N, int H,
                                                       const double* __restrict__ thr_sq, const double* __restrict__ poses,
                                                       const uint8_t* __restrict__ valid, int32_t* __restrict__ counts,
                                                       double* __restrict__ sums) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int h = blockIdx.x * 4 + wave, f = blockIdx.y;
  if (h >= H) return;
  const double* P = poses + ((size_t)f * H + h) * 48;
  const uint8_t* vl = valid + ((size_t)f * H + h) * 4;
  const bool v0 = vl[0], v1 = vl[1], v2 = vl[2], v3 = vl[3];
  int32_t* cnt = counts + ((size_t)f * H + h) * 4;
  double* sm = sums + ((size_t)f * H + h) * 4;
  if (!(v0 || v1 || v2 || v3)) {
    if (lane < 4) { cnt[lane] = -1; sm[lane] = 0.0; }
    return;
  }
  double pose[48];
#pragma unroll
  for (int i = 0; i < 48; ++i) pose[i] = P[i];
  const double thr = thr_sq[f];
  int c[4] = {0, 0, 0, 0};
  double s[4] = {0, 0, 0, 0};
  const double* xf = x + (size_t)f * N * 2;
  const uint8_t* mf = mask ? mask + (size_t)f * N : nullptr;
  for (int n = lane; n < N; n += 64) {
    if (mf && !mf[n]) continue;
    const double X0 = X[3 * (size_t)n], X1 = X[3 * (size_t)n + 1], X2 = X[3 * (size_t)n + 2];
    const double u = xf[2 * (size_t)n], w = xf[2 * (size_t)n + 1];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      double e;
      if (point_error(pose + 12 * k, X0, X1, X2, u, w, thr, e)) { c[k] += 1; s[k] += e; }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) { c[k] = wave_sum_i(c[k]); s[k] = wave_sum(s[k]); }
  if (lane == 0) {
    cnt[0] = v0 ? c[0] : -1; cnt[1] = v1 ? c[1] : -1; cnt[2] = v2 ? c[2] : -1; cnt[3] = v3 ? c[3] : -1;
    sm[0] = s[0]; sm[1] = s[1]; sm[2] = s[2]; sm[3] = s[3];
  }
}

// better support: more inliers; then the smaller residual sum; then the lower index
__device__ inline bool better(int ca, double sa, int ia, int cb, double sb, int ib) {
  if (ca != cb) return ca > cb;
  if (sa != sb) return sa < sb;
  return ia < ib;
}

__global__ __launch_bounds__(256) void p3p_select_kernel(const double* __restrict__ x, const double* __restrict__ X,
                                                        const uint8_t* __restrict__ mask, int N, int H,
                                                        const double* __restrict__ thr_sq, const double* __restrict__ poses,
                                                        const int32_t* __restrict__ counts, const double* __restrict__ sums,
                                                        double* __restrict__ out_pose, int32_t* __restrict__ out_num,
                                                        double* __restrict__ out_sum, int32_t* __restrict__ out_best,
                                                        uint8_t* __restrict__ out_mask) {
  __shared__ int sc[256];
  __shared__ double ss[256];
  __shared__ int si[256];
  const int f = blockIdx.x, tid = threadIdx.x;
  const int M = 4 * H;
  const int32_t* cnt = counts + (size_t)f * M;
  const double* sm = sums + (size_t)f * M;
  int bc = -1, bi = 0x7fffffff;
  double bs = 0.0;
  for (int i = tid; i < M; i += 256) {
    const int ci = cnt[i];
    if (ci < 0) continue;
    if (bi == 0x7fffffff || better(ci, sm[i], i, bc, bs, bi)) { bc = ci; bs = sm[i]; bi = i; }
  }
  sc[tid] = bc; ss[tid] = bs; si[tid] = bi;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) {
      const int oi = si[tid + st];
      if (oi != 0x7fffffff && (si[tid] == 0x7fffffff || better(sc[tid + st], ss[tid + st], oi, sc[tid], ss[tid], si[tid]))) {
        sc[tid] = sc[tid + st]; ss[tid] = ss[tid + st]; si[tid] = oi;
      }
    }
    __syncthreads();
  }
  const int best = si[0], best_c = sc[0];
  const bool found = best != 0x7fffffff && best_c > 0;
  if (tid < 12) out_pose[12 * (size_t)f + tid] = found ? poses[(size_t)f * M * 12 + (size_t)best * 12 + tid] : 0.0;
  if (tid == 0) { out_num[f] = found ? best_c : 0; out_sum[f] = found ? ss[0] : 0.0; out_best[f] = found ? best : -1; }
  uint8_t* om = out_mask + (size_t)f * N;
  if (!found) {
    for (int n = tid; n < N; n += 256) om[n] = 0;
    return;
  }
  double pose[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) pose[i] = poses[(size_t)f * M * 12 + (size_t)best * 12 + i];
  const double thr = thr_sq[f];
  const double* xf = x + (size_t)f * N * 2;
  const uint8_t* mf = mask ? mask + (size_t)f * N : nullptr;
  for (int n = tid; n < N; n += 256) {
    double e;
    const bool in = (!mf || mf[n]) &&
                    point_error(pose, X[3 * (size_t)n], X[3 * (size_t)n + 1], X[3 * (size_t)n + 2], xf[2 * (size_t)n], xf[2 * (size_t)n + 1], thr, e);
    om[n] = in ? 1 : 0;
  }
}

}  // namespace vgg

using namespace vgg;

extern "C" {

size_t vgg_p3p_ransac_workspace_bytes(int num_frames, int num_hypotheses) {
  if (num_frames <= 0 || num_hypotheses <= 0) return 0;
  const size_t fh = (size_t)num_frames * num_hypotheses;
  return fh * 48 * sizeof(double) + fh * 4 * sizeof(double) + fh * 4 * sizeof(int32_t) + fh * 4 + 1024;
}

int vgg_p3p_ransac(const double* points2D_normalized, const double* points3D, const uint8_t* candidate_mask,
                   const int32_t* samples, int num_frames, int frames_per_sample_set, int num_points, int num_hypotheses,
                   const double* max_error_sq, double* out_pose, int32_t* out_num_inliers, double* out_residual_sum,
                   int32_t* out_best, uint8_t* out_inlier_mask, void* workspace, void* stream) {
  const int F = num_frames, N = num_points, H = num_hypotheses, G = frames_per_sample_set;
  if (F < 0 || N < 0 || H <= 0 || G <= 0 || F % G != 0) return VGG_ERR_INVALID_ARGUMENT;
  if (F == 0) return VGG_OK;
  if (N < 3) return VGG_ERR_INVALID_ARGUMENT;
  if (!points2D_normalized || !points3D || !samples || !max_error_sq || !out_pose || !out_num_inliers || !out_residual_sum ||
      !out_best || !out_inlier_mask || !workspace)
    return VGG_ERR_INVALID_ARGUMENT;
  if (F > 65535) return VGG_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const size_t fh = (size_t)F * H;
  char* base = (char*)workspace;
  double* poses = (double*)base; base += fh * 48 * sizeof(double);
  double* sums = (double*)base; base += fh * 4 * sizeof(double);
  int32_t* counts = (int32_t*)base; base += fh * 4 * sizeof(int32_t);
  uint8_t* valid = (uint8_t*)base;
  p3p_hypotheses_kernel<<<dim3(div_up(H, 128), F), 128, 0, st>>>(points2D_normalized, points3D, samples, F, N, H, G, poses, valid);
  p3p_score_kernel<<<dim3(div_up(H, 4), F), 256, 0, st>>>(points2D_normalized, points3D, candidate_mask, F, N, H, max_error_sq,
                                                         poses, valid, counts, sums);
  p3p_select_kernel<<<F, 256, 0, st>>>(points2D_normalized, points3D, candidate_mask, N, H, max_error_sq, poses, counts, sums,
                                       out_pose, out_num_inliers, out_residual_sum, out_best, out_inlier_mask);
  VGG_LAUNCH_CHECK();
  return VGG_OK;
}

}  // extern "C"
