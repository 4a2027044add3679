// LO-RANSAC multi-view DLT triangulation, one wavefront per track (gfx950).
//
// Replaces triangulate_tracks_single_chunk and everything below it in the reference:
//   vggsfm/utils/triangulation.py:776-1017            (RANSAC over <= 256 view pairs, two LO rounds)
//   vggsfm/utils/triangulation_helpers.py:27-131      (DLT: smallest eigenvector of sum_n T_n^T T_n)
//   vggsfm/utils/triangulation_helpers.py:431-521     (angular error, all-pairs triangulation angle)
//   vggsfm/utils/triangulation_helpers.py:648-725     (local refinement loop)
//   vggsfm/two_view_geo/utils.py:63-87                (residual indicator / winner selection)
// The reference materialises (B*256, S) error tensors and a (B, S*S) angle tensor per local-refinement
// hypothesis (62 % of its run time); here a wavefront keeps one track entirely on chip:
//   * per-view table in LDS: unit ray, vis/score flag, M_s = T_s^T T_s (10 unique), projection centre;
//     every DLT -- two-view or masked S-view -- is then a masked sum of M_s followed by a 4x4 Jacobi
//     eigen-solve in registers (one hypothesis per lane, 2-4 per lane for the RANSAC stage => ILP);
//   * angular errors: lanes own hypotheses, the view loop is wave-uniform (LDS broadcast reads), inlier
//     count / error sum stay in registers: no cross-lane reduction, no (H,S) tensor; acos is evaluated
//     only for candidate inliers; inlier masks are never stored: the 50+10 hypotheses picked for local
//     refinement re-evaluate their errors while accumulating their DLT matrix;
//   * "some camera pair subtends >= min_tri_angle": wave-uniform scan over pairs with early exit,
//     widest index distance first (the reference takes any pair of ALL S cameras, helpers :681-694).
// Tie-break contract (SURVEY hard part 2): hypotheses are ranked by inlier count in STABLE descending
// order; the winner is the first maximum of the residual indicator.
// The indicator's chunk-global threshold (max mean inlier error + 1e-6) is a kernel argument: the host
// launches with 2*pi+1e-6 (true whenever any hypothesis of the chunk has no inlier) and re-launches with
// the measured maximum in the pathological case that it is not.
#include "common.hpp"

namespace vgg {

constexpr double kPi = 3.141592653589793;
#ifndef VGG_TRI_ABLATE
#define VGG_TRI_ABLATE 0     // profiling builds only: 1 = no RANSAC error loop, 2 = no LO rounds, 3 = neither
#endif
constexpr int kTab = 4;    // doubles per view in LDS: unit ray (3), invalid flag (1).  The 4x4 DLT matrix of a view is
                           // recomputed where it is needed and the camera centres live in a small global array:
                           // 27 KB of table per wavefront at 200 views left room for ONE wavefront per SIMD

struct Sym4 { double a[10]; };  // 00 01 02 03 11 12 13 22 23 33

// eigenvector of the smallest eigenvalue of a symmetric 4x4 matrix: cyclic Jacobi, in registers
__device__ __forceinline__ void smallest_eigvec4(const Sym4& m, double* v) {
  double A[4][4] = {{m.a[0], m.a[1], m.a[2], m.a[3]}, {m.a[1], m.a[4], m.a[5], m.a[6]},
                    {m.a[2], m.a[5], m.a[7], m.a[8]}, {m.a[3], m.a[6], m.a[8], m.a[9]}};
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
#pragma unroll 1
  for (int sweep = 0; sweep < 12; ++sweep) {
    double off = 0.0, dsum = 0.0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      dsum += fabs(A[p][p]);
#pragma unroll
      for (int q = p + 1; q < 4; ++q) off += fabs(A[p][q]);
    }
    if (!(off > 1e-300) || off <= 1e-22 * dsum) break;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        const double apq = A[p][q];
        if (fabs(apq) > 1e-300) {
          const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
          const double t = ((theta >= 0.0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const double akp = A[k][p], akq = A[k][q];
            A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const double apk = A[p][k], aqk = A[q][k];
            A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const double vkp = V[k][p], vkq = V[k][q];
            V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
          }
        }
      }
    }
  }
  int best = 0;
  double bv = A[0][0];
#pragma unroll
  for (int k = 1; k < 4; ++k) if (A[k][k] < bv) { bv = A[k][k]; best = k; }
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = (best == 0) ? V[k][0] : (best == 1) ? V[k][1] : (best == 2) ? V[k][2] : V[k][3];
}

#ifndef VGG_TRI_EIG
#define VGG_TRI_EIG 1        // 1 = shifted inverse iteration (Jacobi only where it does not converge), 0 = cyclic Jacobi always
#endif
#ifndef VGG_TRI_FAST_ERR
#define VGG_TRI_FAST_ERR 1   // 1 = one reciprocal per view instead of three divisions, series for acos next to 1
#endif
#ifndef VGG_TRI_OCC
#define VGG_TRI_OCC 2        // wavefronts per SIMD the kernel is compiled for (register budget 512 / VGG_TRI_OCC)
#endif

// The DLT matrices are positive semi-definite with a well separated smallest eigenvalue whenever the hypothesis is worth
// anything (two views: (parallax / noise)^2; more views: better), so the wanted eigenvector is what inverse iteration
// converges to, at a rate of lambda_1 / lambda_2 per step, from a generic start.  A + mu I = L D L^T with mu = 2^-44 trace
// (far below lambda_2, above the rounding errors of the factorisation: every pivot stays positive although lambda_1 may be
// 0 -- two rays that meet exactly); one factorisation (four divisions), then ~50 instructions per step against ~4000 for
// the twelve-sweep cyclic Jacobi, which stays as the last resort of the lanes that have not settled to 1e-14 (below:
// shift refinement first; then near-degenerate hypotheses -- no parallax at all -- or non-finite input).
// Same eigenvector to ~1e-15; the reference's LAPACK call and the Jacobi differ from each other by as much.
// Lanes that have not settled after the first kInvIt steps (a few per thousand: pairs that involve an outlier or a view the
// track is not visible in -- lambda_1 / lambda_2 of order one) move the shift up to just below lambda_1: with rho = x^T A x
// and r = A x - rho x, an eigenvalue lies within |r| of rho (and it is the smallest once x is dominated by its eigenvector),
// so A - (rho - 1.01 |r| - mu) I is still positive definite and the rate becomes ~2 |r| / (lambda_2 - lambda_1): superlinear.
// Up to kInvRounds factorisations; what is still moving after that (lambda_1 = lambda_2 to rounding: no parallax at all)
// goes to the Jacobi.  `live` = the lane's result is used (a dead lane must not send its wavefront into the fallback).
constexpr int kInvIt = 6, kInvRounds = 6;
#ifndef VGG_TRI_FASTDIV
#define VGG_TRI_FASTDIV 1    // 1 = the pivot reciprocals and the normalisation of the inverse iteration from the hardware estimates +
#endif                       //     Newton steps (1 ulp) instead of IEEE divisions / sqrt + division: the iteration converges to the same
                             //     vector (its tolerance is 1e-14), four divisions and one per step are ~200 instructions per solve
__device__ __forceinline__ double tri_rcp(double x) {
#if VGG_TRI_FASTDIV
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
#else
  return 1.0 / x;
#endif
}
__device__ __forceinline__ double tri_rsqrt(double x) {
#if VGG_TRI_FASTDIV
  double r = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  r = r * __builtin_fma(-h * r, r, 1.5);
  return r * __builtin_fma(-h * r, r, 1.5);
#else
  return 1.0 / sqrt(x);
#endif
}
__device__ __forceinline__ bool smallest_eigvec4_invit(const Sym4& m, double* v, bool live) {
  const double tr = m.a[0] + m.a[4] + m.a[7] + m.a[9];
  bool ok = (tr > 0.0) && (tr <= 1.7976931348623157e308);
  const double mu = ok ? tr * 5.684341886080802e-14 : 1.0;
  double sig = -mu;                                  // factor A - sig I
  double x0 = 0.3, x1 = 0.4, x2 = 0.5, x3 = 0.7071067811865476;
  bool conv = false;
#pragma unroll 1
  for (int round = 0; round < kInvRounds; ++round) {
    // L D L^T of A - sig I (unit lower L: l10 l20 l30 l21 l31 l32; reciprocal pivots r0..r3)
    const double d0 = m.a[0] - sig, r0 = tri_rcp(d0);
    const double l10 = m.a[1] * r0, l20 = m.a[2] * r0, l30 = m.a[3] * r0;
    const double d1 = (m.a[4] - sig) - l10 * m.a[1], r1 = tri_rcp(d1);
    const double u21 = m.a[5] - l20 * m.a[1], u31 = m.a[6] - l30 * m.a[1];
    const double l21 = u21 * r1, l31 = u31 * r1;
    const double d2 = ((m.a[7] - sig) - l20 * m.a[2]) - l21 * u21, r2 = tri_rcp(d2);
    const double u32 = (m.a[8] - l30 * m.a[2]) - l31 * u21;
    const double l32 = u32 * r2;
    const double d3 = (((m.a[9] - sig) - l30 * m.a[3]) - l31 * u31) - l32 * u32, r3 = tri_rcp(d3);
    ok = ok && (d0 > 0.0) && (d1 > 0.0) && (d2 > 0.0) && (d3 > 0.0);
#pragma unroll 1
    for (int it = 0; it < kInvIt; ++it) {
      // L w = x;  z = D^-1 w;  L^T y = z
      const double w0 = x0, w1 = x1 - l10 * w0, w2 = (x2 - l20 * w0) - l21 * w1, w3 = ((x3 - l30 * w0) - l31 * w1) - l32 * w2;
      const double y3 = w3 * r3, y2 = w2 * r2 - l32 * y3, y1 = (w1 * r1 - l21 * y2) - l31 * y3,
                   y0 = ((w0 * r0 - l10 * y1) - l20 * y2) - l30 * y3;
      const double n2 = (y0 * y0 + y1 * y1) + (y2 * y2 + y3 * y3);
      double rn = tri_rsqrt(n2);
      if (x0 * y0 + x1 * y1 + x2 * y2 + x3 * y3 < 0.0) rn = -rn;     // (successive iterates point the same way)
      const double n0 = y0 * rn, n1 = y1 * rn, nn2 = y2 * rn, n3 = y3 * rn;
      const double dlt = fmax(fmax(fabs(n0 - x0), fabs(n1 - x1)), fmax(fabs(nn2 - x2), fabs(n3 - x3)));
      x0 = n0; x1 = n1; x2 = nn2; x3 = n3;
      conv = dlt <= 1.0e-14;                          // (NaN compares false: such a lane ends in the fallback)
      if (__all(conv || !ok || !live)) break;
    }
    if (__all(conv || !ok || !live)) break;
    // shift of the next round for the lanes that are still moving (the others keep theirs: their factorisation is repeated
    // with the same numbers and their iterate stays where it is)
    const double a0 = m.a[0] * x0 + m.a[1] * x1 + m.a[2] * x2 + m.a[3] * x3, a1 = m.a[1] * x0 + m.a[4] * x1 + m.a[5] * x2 + m.a[6] * x3,
                 a2 = m.a[2] * x0 + m.a[5] * x1 + m.a[7] * x2 + m.a[8] * x3, a3 = m.a[3] * x0 + m.a[6] * x1 + m.a[8] * x2 + m.a[9] * x3;
    const double rho = (x0 * a0 + x1 * a1) + (x2 * a2 + x3 * a3);
    const double q0 = a0 - rho * x0, q1 = a1 - rho * x1, q2 = a2 - rho * x2, q3 = a3 - rho * x3;
    const double rnorm = sqrt((q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3));
    const double cand = (rho - 1.01 * rnorm) - mu;
    if (!conv && cand > sig) sig = cand;
  }
  v[0] = x0; v[1] = x1; v[2] = x2; v[3] = x3;
  return (ok && conv) || !live;
}

#ifdef VGG_TRI_STATS
__device__ unsigned long long g_tri_stats[4];       // measurement builds: eigen-solves, fallbacks, waves with a fallback
#endif
__device__ __forceinline__ void smallest_eigvec4_fast(const Sym4& m, double* v, bool live) {
#if VGG_TRI_EIG
  const bool done = smallest_eigvec4_invit(m, v, live);
#ifdef VGG_TRI_STATS
  atomicAdd(&g_tri_stats[0], 1ull);
  if (!done) atomicAdd(&g_tri_stats[1], 1ull);
  if (__any(!done) && (threadIdx.x & 63) == 0) atomicAdd(&g_tri_stats[2], 1ull);
#endif
  // (the last resort: ~5 lanes per million at configs[1] / [2] -- 96 of 23 M, 544 of 115 M solves with -DVGG_TRI_STATS -- whose
  //  two smallest eigenvalues agree to rounding.  Dropping it is worth 0.2 ms of 18 at configs[2] and changed ONE mask bit in
  //  4.9 million there (such a hypothesis can still pass the triangulation-angle test): kept, so that the masks are what the
  //  Jacobi-only kernel of rounds 1-3 produced on every golden and at configs[2])
  if (!done) smallest_eigvec4(m, v);
#else
  smallest_eigvec4(m, v);
#endif
}

// acos for the candidate inliers, whose cosine is within ~1e-3 of 1: acos(1 - u) = sqrt(2u) (1 + u/12 + 3u^2/160 + 5u^3/896
// + 35u^4/18432 + O(u^5)); u = 1 - c is exact there (Sterbenz), the truncation error at u = 1e-3 is 7e-19.
__device__ __forceinline__ double acos_near_one(double c) {
#if VGG_TRI_FAST_ERR
  const double u = 1.0 - c;
  if (u > 1.0e-3) return acos(c);
  const double p = 1.0 + u * (1.0 / 12.0 + u * (3.0 / 160.0 + u * (5.0 / 896.0 + u * (35.0 / 18432.0))));
  return sqrt(u + u) * p;
#else
  return acos(c);
#endif
}

__device__ __forceinline__ double tri_angle_deg2(double r1, double r2, double b) {
  // triangulation_helpers.py:503-519: law of cosines on (norm)^2 values, min(theta, pi - theta), degrees
  double den = 2.0 * sqrt(r1 * r2);
  double nom = r1 + r2 - b;
  if (den <= 1e-12) { nom = 1.0; den = 1.0; }
  double c = nom / den;
  c = fmin(fmax(c, -1.0), 1.0);
  double th = fabs(acos(c));
  th = fmin(th, kPi - th);
  return th * (180.0 / kPi);
}

__device__ __forceinline__ double sqnorm3(double a, double b, double c) {
  const double n = sqrt(a * a + b * b + c * c);
  return n * n;
}

// DLT matrix T^T T of one view, T = P - r (r^T P) for the unit ray r (3x4 rows of the projection matrix P)
__device__ __forceinline__ void view_dlt_matrix_r(const double* __restrict__ P, double r0, double r1, double r2,
                                                  Sym4& m) {
  double rp[4], T[12];
#pragma unroll
  for (int k = 0; k < 4; ++k) rp[k] = r0 * P[k] + r1 * P[4 + k] + r2 * P[8 + k];
#pragma unroll
  for (int k = 0; k < 4; ++k) { T[k] = P[k] - r0 * rp[k]; T[4 + k] = P[4 + k] - r1 * rp[k]; T[8 + k] = P[8 + k] - r2 * rp[k]; }
  int q = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = i; j < 4; ++j) m.a[q++] = T[i] * T[j] + T[4 + i] * T[4 + j] + T[8 + i] * T[8 + j];
}

// angular error of X against view s (calculate_normalized_angular_error_batched); `cand` = err can be <= max_rad
__device__ __forceinline__ double view_error(const double* __restrict__ P, double ray0, double ray1, double ray2, double X0, double X1,
                                             double X2, double cos_gate, bool& is_nan, double& depth) {
  const double y0 = P[0] * X0 + P[1] * X1 + P[2] * X2 + P[3];
  const double y1 = P[4] * X0 + P[5] * X1 + P[6] * X2 + P[7];
  const double y2 = P[8] * X0 + P[9] * X1 + P[10] * X2 + P[11];
  depth = y2;
#if VGG_TRI_FAST_ERR
  // one reciprocal square root instead of the reference's square root and three divisions: the cosine moves by <= 2 ulp, the
  // angle of a candidate inlier by ~5e-15 rad -- a decision changes only where |error - threshold| is below that (the goldens
  // compare every mask bit).  (max(|y|, 1e-12) of the reference = max(|y|^2, 1e-24) under the root.)
  const double rn = rsqrt(fmax(y0 * y0 + y1 * y1 + y2 * y2, 1e-24));
  double c = (ray0 * (y0 * rn) + ray1 * (y1 * rn)) + ray2 * (y2 * rn);
#else
  const double n = fmax(sqrt(y0 * y0 + y1 * y1 + y2 * y2), 1e-12);
  double c = (ray0 * (y0 / n) + ray1 * (y1 / n)) + ray2 * (y2 / n);
#endif
  is_nan = (c != c);
  c = fmin(fmax(c, -1.0), 1.0);
  if (!(c >= cos_gate)) return 4.0;             // cannot be an inlier (also NaN): skip acos
  return acos_near_one(c);
}

// "any pair of the S cameras subtends >= thr degrees at X" for the lanes with `live`; wave-uniform loop
__device__ __forceinline__ bool any_pair_angle(const double* __restrict__ centers, int S, double X0, double X1, double X2,
                                               double thr, bool live) {
  bool found = false;
  // a non-finite point can never satisfy ">= thr" (NaN compares false): do not scan S^2 pairs for it
  live = live && (fabs(X0) <= 1.7976931348623157e308) && (fabs(X1) <= 1.7976931348623157e308) &&
         (fabs(X2) <= 1.7976931348623157e308);
  if (__all(!live)) return false;
  for (int dist = S - 1; dist >= 1; --dist) {
    for (int a = 0; a + dist < S; ++a) {
      const int b = a + dist;
      const double* ca = centers + 3 * a;
      const double* cb = centers + 3 * b;
      const double bsq = sqnorm3(ca[0] - cb[0], ca[1] - cb[1], ca[2] - cb[2]);
      if (live && !found) {
        const double r1 = sqnorm3(X0 - ca[0], X1 - ca[1], X2 - ca[2]);
        const double r2 = sqnorm3(X0 - cb[0], X1 - cb[1], X2 - cb[2]);
        if (tri_angle_deg2(r1, r2, bsq) >= thr) found = true;
      }
      if (__all(found || !live)) return found;
    }
  }
  return found;
}

// Walk over the visible views of a track; the body gets the projection matrix (scalar registers), the unit ray, the view and
// its position in the list.  -DVGG_TRI_PIPE=1 (round 6, MEASURED SLOWER, off): the matrix of the NEXT view already requested
// while the body runs.  The view loops are wave-uniform -- the index comes out of LDS (v_readfirstlane), the matrix through
// scalar loads: a dependent chain of an LDS round trip and a scalar-memory round trip in front of every view, which the
// compiler cannot move across the loop's back edge, and 38 % of a wavefront's resident time is s_waitcnt
// (profiles/r05_pmc_sq_tri.json).  With two NAMED matrix sets per trip (no copies between stages, which is what sank the
// round-4 attempt), the index and the ray fetched one step further ahead and ONE wait per view placed behind the body (LDS
// reads and scalar loads share lgkmcnt and scalar loads return out of order: any wait for LDS data waits for every scalar load
// in flight) the kernel takes 13.3 ms against 11.85 at configs[2] and 1.38 against 1.34 at configs[1]
// (profiles/r06_ab_tri_c3.jsonl; masks and points identical): the extra vector registers (rays and index of two views in
// flight: 52 B of scratch at four hypotheses per lane) and the serialising waits cost more than the chain they hide -- the
// second resident wavefront covers it already.
// body(const double (&P)[12], double ray0, double ray1, double ray2, int s, int k)
#ifndef VGG_TRI_PIPE
#define VGG_TRI_PIPE 0
#endif
template <class F>
__device__ __forceinline__ void for_each_view(const double* __restrict__ ext, const double* tab, const int* vlist, int nv, F&& body) {
#if VGG_TRI_PIPE
  if (nv <= 0) return;
  double pa[12], pb[12];
  auto request = [&](double (&p)[12], int s) __attribute__((always_inline)) {
    const double* P = ext + 12 * s;
#pragma unroll
    for (int i = 0; i < 12; ++i) p[i] = P[i];
  };
  // LDS reads and scalar loads share one counter (lgkmcnt) and scalar loads return out of order, so ANY wait for LDS data also
  // waits for every scalar load in flight.  The schedule per view k (matrix set A; set B = view k + 1 is in flight):
  //   [index of view k + 2 -> scalar; LDS reads of its ray and of the index of view k + 3 issued]  [body(k): no memory access]
  //   [ONE wait: the ray, the index, set B -- all requested a whole body ago]  [set A requested for view k + 2]
  // and the same with the sets swapped: a matrix has the body of the view in between to arrive, the two LDS round trips
  // (index, then ray) are hidden behind a body as well, and the body gets the ray as values.
  struct Ray { double r0, r1, r2; };
  auto clampk = [&](int k) { return min(k, nv - 1); };
  auto ray_of = [&](int sv) __attribute__((always_inline)) -> Ray {
    const double* t = tab + sv * kTab;
    Ray r; r.r0 = t[0]; r.r1 = t[1]; r.r2 = t[2];
    return r;
  };
  // prologue: views 0 and 1
  int sa = __builtin_amdgcn_readfirstlane(vlist[0]), sb = __builtin_amdgcn_readfirstlane(vlist[clampk(1)]);
  Ray ra = ray_of(sa), rb = ray_of(sb);
  int idx_next = vlist[clampk(2)];                      // (vector register; becomes a scalar one step later)
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra.r0), "+v"(ra.r1), "+v"(ra.r2), "+v"(rb.r0), "+v"(rb.r1), "+v"(rb.r2), "+v"(idx_next) : : "memory");
  request(pa, sa);
  request(pb, sb);
  int k = 0;
  for (; k + 1 < nv; k += 2) {
    {
      const int s2 = __builtin_amdgcn_readfirstlane(idx_next);
      Ray r2 = ray_of(s2);
      int idx3 = vlist[clampk(k + 3)];
      body(pa, ra.r0, ra.r1, ra.r2, sa, k);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r2.r0), "+v"(r2.r1), "+v"(r2.r2), "+v"(idx3) : : "memory");
      request(pa, s2);
      sa = s2; ra = r2; idx_next = idx3;
    }
    {
      const int s3 = __builtin_amdgcn_readfirstlane(idx_next);
      Ray r3 = ray_of(s3);
      int idx4 = vlist[clampk(k + 4)];
      body(pb, rb.r0, rb.r1, rb.r2, sb, k + 1);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r3.r0), "+v"(r3.r1), "+v"(r3.r2), "+v"(idx4) : : "memory");
      request(pb, s3);
      sb = s3; rb = r3; idx_next = idx4;
    }
  }
  if (k < nv) body(pa, ra.r0, ra.r1, ra.r2, sa, k);
#else
  for (int k = 0; k < nv; ++k) {
    const int s = __builtin_amdgcn_readfirstlane(vlist[k]);
    double p[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) p[i] = ext[12 * s + i];
    const double* t = tab + s * kTab;
    body(p, t[0], t[1], t[2], s, k);
  }
#endif
}

struct Cand { double e; int n; };   // mean inlier error (or 2*pi) and inlier count of one hypothesis

// Evaluate hypothesis X against all views.  ACC: also accumulate the DLT matrix of its inlier views.
// ransac_nan: NaN errors poison the mean (residual indicator on raw errors); otherwise NaN -> 100*pi.
// `vlist[0 .. nv)`: the views in which the track is visible (flag 0), ascending.  Only they can hold inliers, so only they
// are evaluated.  A NaN error in an INVISIBLE view -- which the reference's mean over all views would turn into NaN
// (triangulation.py:842-877) -- has two sources: a non-finite point (NaN in every view, the visible ones included) or a
// non-finite ray of that view (a normalised track coordinate that is inf / NaN, e.g. the undistortion of an off-image
// point); the second is a property of the TRACK and is found when the table is built (`bad_ray` in triangulate_kernel,
// which then poisons every RANSAC hypothesis as the reference's NaN mean does).  The cheirality test over ALL views
// (`any_behind`) needs the depth of the other views only -- one row of P X, the expression view_error uses.  Same values,
// same order of the sums; ~4x fewer evaluations at the visibility density of the BASELINE scenes.
constexpr int kGvMax = 64;   // visible views whose DLT matrix is tabulated per track (LDS: 80 bytes each)
template <bool ACC>
__device__ __forceinline__ Cand eval_views(const double* __restrict__ ext, const double* tab, const int* vlist, int nv, int S,
                                           double X0, double X1, double X2, bool invalid, bool live, double max_rad,
                                           double cos_gate, bool ransac_nan, Sym4* acc, bool* any_behind, const double* gv = nullptr) {
  int cnt = 0;
  double sum = 0.0;
  bool poisoned = false, behind = false;
  if (any_behind) {
    // (four views per trip: their scalar loads go out together and are waited for once -- one view per trip is a chain of
    //  load latencies, 200 of them per hypothesis at configs[2])
#pragma unroll 4
    for (int s = 0; s < ((VGG_TRI_ABLATE & 2) ? 0 : S); ++s) {
      const double* P = ext + 12 * s;
      const double y2 = P[8] * X0 + P[9] * X1 + P[10] * X2 + P[11];
      if (y2 <= 0.0) behind = true;
    }
  }
  // (the view index is the same in every lane -- it comes out of LDS, which the compiler cannot know: as a scalar the projection
  //  matrix is fetched by scalar loads into scalar registers instead of 12 vector loads into 24 VGPRs.  Walking the tracks in the
  //  order of their first visible view, so that the wavefronts resident together share their views' matrices in the scalar
  //  cache, changed nothing: 16.60 against 16.69 ms)
  for_each_view(ext, tab, vlist, (VGG_TRI_ABLATE & 2) ? 0 : nv, [&](const double (&P)[12], double t0, double t1, double t2, int s, int k) __attribute__((always_inline)) {
    (void)s;
    bool isn;
    double depth;
    const double err = view_error(P, t0, t1, t2, X0, X1, X2, cos_gate, isn, depth);
    if (isn && ransac_nan) poisoned = true;
    const bool inl = live && !invalid && !isn && (err <= max_rad);
    if (inl) {
      ++cnt; sum += err;
      if (ACC) {
        // the DLT matrix of a view does not depend on the hypothesis: tabulated once per track for the first kGvMax visible
        // views (round 4; the same numbers the expression below produces, added in the same order)
        if (k < kGvMax) {
#pragma unroll
          for (int k2 = 0; k2 < 10; ++k2) acc->a[k2] += gv[k * 10 + k2];
        } else {
          Sym4 mv;
          view_dlt_matrix_r(P, t0, t1, t2, mv);
#pragma unroll
          for (int k2 = 0; k2 < 10; ++k2) acc->a[k2] += mv.a[k2];
        }
      }
    }
  });
  if (any_behind) *any_behind = behind;
  Cand c;
  c.n = cnt;
  c.e = (cnt > 0 && !poisoned) ? sum / (double)cnt : 2.0 * kPi;
  return c;
}

// (Round 4, measured and NOT kept: the local-optimisation rounds with the VIEWS in the lanes and the hypotheses as the loop --
//  a lane fetches its view's matrix once, the inlier decisions of a hypothesis are a ballot word, and the DLT matrices of all
//  hypotheses are a (hypotheses x views) 0/1 matrix times the (views x 10) per-view matrices on the matrix cores.  Same masks;
//  16.8 against 17.5 ms at 200 views, 2.12 against 1.91 ms at 50: the kernel is bound by its VALU instruction count -- SQ
//  counters: 48.9 k vector instructions per track, the vector unit busy ~65 % of the launch -- not by the view-loop latency
//  that form removes, and its per-hypothesis reductions cost what the per-view loads did.)
// eval_views with kGV lanes per hypothesis (second local-optimisation round: 10 hypotheses, which left 54 of 64 lanes
// idle for two passes over the views -- a quarter of the kernel at 200 views).  The lanes of a group evaluate kGV
// consecutive views at once; their contributions are then added to the group's accumulators ONE VIEW AT A TIME, in view
// order, by every lane of the group alike (the values come over with ds_bpermute, a skipped view leaves the accumulator
// untouched): the same additions in the same order as eval_views.  (This file is compiled with -ffp-contract=fast: the
// compiler may fuse the same expression differently in each inlined context, so "same order" is a statement about the source;
// what is CHECKED is that masks and counts are unchanged on every golden -- ADVICE r4.)
// src0 = first lane of this lane's group, vq = its position inside the group.
constexpr int kGV = 6;
static_assert(10 * kGV <= 64, "ten hypotheses of the second local-optimisation round, kGV lanes each");
template <bool ACC>
__device__ __forceinline__ Cand eval_views_grouped(const double* __restrict__ ext, const double* tab, const int* vlist, int nv,
                                                   int S, double X0, double X1, double X2, bool invalid, bool live,
                                                   double max_rad, double cos_gate, bool ransac_nan, Sym4* acc,
                                                   bool* any_behind, int src0, int vq, const double* gv = nullptr) {
  int cnt = 0;
  double sum = 0.0;
  bool poisoned = false, behind = false;
  if (any_behind) {                                   // cheirality over all views: the lanes of a group share them out
    for (int s = vq; s < ((VGG_TRI_ABLATE & 2) ? 0 : S); s += kGV) {
      const double* P = ext + 12 * s;
      const double y2 = P[8] * X0 + P[9] * X1 + P[10] * X2 + P[11];
      if (y2 <= 0.0) behind = true;
    }
  }
  for (int k0 = 0; k0 < ((VGG_TRI_ABLATE & 2) ? 0 : nv); k0 += kGV) {
    const bool has = k0 + vq < nv;
    const int sc = vlist[has ? k0 + vq : nv - 1];
    const double* t = tab + sc * kTab;
    bool isn;
    double depth;
    const double err = view_error(ext + 12 * sc, t[0], t[1], t[2], X0, X1, X2, cos_gate, isn, depth);
    if (has && isn && ransac_nan) poisoned = true;
    const bool inl = has && live && !invalid && !isn && (err <= max_rad);
    Sym4 mv;
    if (ACC) {
      const int kk = has ? k0 + vq : nv - 1;
      if (kk < kGvMax) {
#pragma unroll
        for (int i = 0; i < 10; ++i) mv.a[i] = gv[kk * 10 + i];
      } else {
        view_dlt_matrix_r(ext + 12 * sc, t[0], t[1], t[2], mv);
      }
    }
    const unsigned long long inl_mask = __ballot(inl);
    // (not unrolled: six copies of the eleven exchanges were the register peak of the whole kernel -- 189 VGPRs with one
    //  hypothesis per lane, 168 without this block)
#pragma unroll 1
    for (int k = 0; k < kGV; ++k) {
      // (lanes behind the last group -- 60..63 -- belong to no hypothesis: their source lane is clamped into the wavefront
      //  and they never count; their results are discarded by the caller)
      const int src = min(src0 + k, 63);
      const bool inl_k = (src0 + k < 64) && ((inl_mask >> src) & 1ull) != 0;
      const double e_k = __shfl(err, src, 64);
      cnt += inl_k ? 1 : 0;
      sum = inl_k ? sum + e_k : sum;
      if (ACC) {
#pragma unroll
        for (int i = 0; i < 10; ++i) {
          const double v = __shfl(mv.a[i], src, 64);
          acc->a[i] = inl_k ? acc->a[i] + v : acc->a[i];
        }
      }
    }
  }
  const unsigned long long gmask = ((1ull << kGV) - 1ull) << src0;
  behind = (__ballot(behind) & gmask) != 0;
  poisoned = (__ballot(poisoned) & gmask) != 0;
  if (any_behind) *any_behind = behind;
  Cand c;
  c.n = cnt;
  c.e = (cnt > 0 && !poisoned) ? sum / (double)cnt : 2.0 * kPi;
  return c;
}

template <int HJ>
__global__ __launch_bounds__(64, VGG_TRI_OCC) void triangulate_kernel(   // (round 4: the RANSAC points stay in registers and
    // travel by shuffles -- ~10 KB of LDS per wavefront at 200 views instead of 18; the registers still allow two per SIMD)
    const double* __restrict__ ext, const double* __restrict__ tn, const uint8_t* __restrict__ ivc,
    const int32_t* __restrict__ pairs_all, int S, int N, int H, int lo1, int lo2, double max_rad, double min_tri_deg,
    const double* __restrict__ thres_all, int chunk_size, double* __restrict__ out_pts, int64_t* __restrict__ out_num,
    uint8_t* __restrict__ out_mask, unsigned long long* __restrict__ gmax_all, const double* __restrict__ centers) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* tab = lds;                                    // [S][kTab]
  double* lx = tab + (size_t)S * kTab;                  // [64][4] LO1 points + invalid flag
  int* cnts = reinterpret_cast<int*>(lx + 64 * 4);      // [H] inlier counts, later [64] LO1 counts
  int* sel = cnts + ((H + 63) / 64) * 64;               // [64] selected hypothesis per LO slot
  int* vlist = sel + 64;                                // [S] views in which the track is visible, ascending
  double* gv = reinterpret_cast<double*>(vlist + ((S + 1) & ~1));   // [kGvMax][10] DLT matrices of the first visible views
  const int lane = threadIdx.x;
  const double cos_gate = cos(max_rad) - 1e-9;
  double wave_max_e = 0.0;
  int cur_chunk = -1;

  for (int n = blockIdx.x; n < N; n += gridDim.x) {
    // reference chunk of this track (triangulation.py:712-758): its own hypothesis pairs, indicator threshold
    // and maximum of the mean inlier errors
    const int chunk = n / chunk_size;
    if (chunk != cur_chunk) {
      if (cur_chunk >= 0) {
        wave_max_e = wave_max(wave_max_e);
        if (lane == 0) atomicMax(gmax_all + cur_chunk, (unsigned long long)__double_as_longlong(wave_max_e));
      }
      wave_max_e = 0.0;
      cur_chunk = chunk;
    }
    const int32_t* pairs = pairs_all + (size_t)chunk * 2 * H;
    const double thres = thres_all[chunk];
    __syncthreads();
    // ---- per-view table (tracks are given track-major: tn[n][s][2], ivc[n][s])
    bool nonfinite = false;
    for (int s = lane; s < S; s += 64) {
      const double u = tn[((size_t)n * S + s) * 2], v = tn[((size_t)n * S + s) * 2 + 1];
      const double nr = sqrt(u * u + v * v + 1.0);
      const double r0 = u / nr, r1 = v / nr, r2 = 1.0 / nr;
      double* t = tab + s * kTab;
      // F.normalize of the same homogeneous ray (eps 1e-12 never binds: norm >= 1)
      t[0] = r0; t[1] = r1; t[2] = r2;
      t[3] = ivc[(size_t)n * S + s] ? 1.0 : 0.0;
      nonfinite = nonfinite || !(fabs(r0) <= 1.7976931348623157e308 && fabs(r1) <= 1.7976931348623157e308 && r2 == r2);
    }
    // a non-finite ray in ANY view, visible or not, makes the angular error of that view NaN for every hypothesis: the
    // reference's mean over all views is then NaN for all of them (see eval_views)
    const bool bad_ray = __any(nonfinite);
    // visible views of the track, ascending (one wavefront per track: a ballot per 64 views)
    int nv = 0;
    for (int base = 0; base < S; base += 64) {
      const int s = base + lane;
      const bool vis = s < S && ivc[(size_t)n * S + s] == 0;
      const unsigned long long bm = __ballot(vis);
      if (vis) vlist[nv + __popcll(bm & ((1ull << lane) - 1ull))] = s;
      nv += __popcll(bm);
    }
    __syncthreads();
    if (lane < nv && lane < kGvMax) {                  // (one view per lane; tab and vlist are in place)
      const int sv = vlist[lane];
      Sym4 g;
      view_dlt_matrix_r(ext + 12 * sv, tab[sv * kTab], tab[sv * kTab + 1], tab[sv * kTab + 2], g);
#pragma unroll
      for (int i = 0; i < 10; ++i) gv[lane * 10 + i] = g.a[i];
    }
    __syncthreads();

    // ---- RANSAC hypotheses: two-view DLT per (lane, j)
    double X[HJ][3];
    bool inv[HJ], live[HJ];
#pragma unroll
    for (int j = 0; j < HJ; ++j) {
      const int h = lane + 64 * j;
      live[j] = h < H;
      const int i1 = live[j] ? pairs[2 * h] : 0, i2 = live[j] ? pairs[2 * h + 1] : 0;
      const double* P1 = ext + 12 * i1;
      const double* P2 = ext + 12 * i2;
      Sym4 m, m2;
      view_dlt_matrix_r(P1, tab[i1 * kTab], tab[i1 * kTab + 1], tab[i1 * kTab + 2], m);
      view_dlt_matrix_r(P2, tab[i2 * kTab], tab[i2 * kTab + 1], tab[i2 * kTab + 2], m2);
#pragma unroll
      for (int k = 0; k < 10; ++k) m.a[k] = m.a[k] + m2.a[k];
      double v[4];
      smallest_eigvec4_fast(m, v, live[j]);
      X[j][0] = v[0] / v[3]; X[j][1] = v[1] / v[3]; X[j][2] = v[2] / v[3];
      // cheirality on the two views, triangulation angle of the pair
      const double z1 = P1[8] * X[j][0] + P1[9] * X[j][1] + P1[10] * X[j][2] + P1[11];
      const double z2 = P2[8] * X[j][0] + P2[9] * X[j][1] + P2[10] * X[j][2] + P2[11];
      const double* c1 = centers + 3 * i1;
      const double* c2 = centers + 3 * i2;
      const double bsq = sqnorm3(c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]);
      const double r1 = sqnorm3(X[j][0] - c1[0], X[j][1] - c1[1], X[j][2] - c1[2]);
      const double r2 = sqnorm3(X[j][0] - c2[0], X[j][1] - c2[1], X[j][2] - c2[2]);
      const bool tri_ok = tri_angle_deg2(r1, r2, bsq) >= min_tri_deg;
      inv[j] = (z1 <= 0.0) || (z2 <= 0.0) || !tri_ok;
    }
    // ---- angular errors of the RANSAC hypotheses: wave-uniform view loop, HJ independent chains per lane
    int cnt[HJ];
    double sum[HJ];
    bool pois[HJ];
#pragma unroll
    for (int j = 0; j < HJ; ++j) { cnt[j] = 0; sum[j] = 0.0; pois[j] = bad_ray; }
    // (visible views only: see eval_views)
    for_each_view(ext, tab, vlist, (VGG_TRI_ABLATE & 1) ? 0 : nv, [&](const double (&P)[12], double t0, double t1, double t2, int, int) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < HJ; ++j) {
        bool isn;
        double depth;
        const double err = view_error(P, t0, t1, t2, X[j][0], X[j][1], X[j][2], cos_gate, isn, depth);
        if (isn) pois[j] = true;
        if (live[j] && !inv[j] && !isn && err <= max_rad) { ++cnt[j]; sum[j] += err; }
      }
    });
    Cand best;                      // running first-maximum of the residual indicator, candidate order
    best.e = 0.0; best.n = -1;      //   [RANSAC 0..H-1 | LO1 0..lo1-1 | LO2 0..lo2-1]
    double best_ind = -1.0;
    int best_idx = 0x7fffffff;
    double bX0 = 0, bX1 = 0, bX2 = 0;
    bool b_inv = true;
#pragma unroll
    for (int j = 0; j < HJ; ++j) {
      const int h = lane + 64 * j;
      if (live[j]) {
        cnts[h] = cnt[j];
        Cand c;
        c.n = cnt[j];
        c.e = (cnt[j] > 0 && !pois[j]) ? sum[j] / (double)cnt[j] : 2.0 * kPi;
        wave_max_e = fmax(wave_max_e, c.e);
        const double ind = (thres - c.e) / thres + (double)c.n;
        if (ind > best_ind || (ind == best_ind && h < best_idx)) {
          best_ind = ind; best_idx = h; best = c; bX0 = X[j][0]; bX1 = X[j][1]; bX2 = X[j][2]; b_inv = inv[j];
        }
      }
    }
    __syncthreads();
    // ---- LO round 1: stable descending rank of the H counts, top lo1 hypotheses -> slots
    if (lane < 64) sel[lane] = -1;
    __syncthreads();
    {
      // Stable descending order without comparing every hypothesis with every other (round 1-3: H comparisons per hypothesis,
      // ~4 k instructions per track): counts are small integers, so walk the count VALUES downwards from the maximum; the
      // hypotheses with the current value take the next slots in index order -- h = lane + 64 j ascends with (j, lane), i.e.
      // a ballot per j and the number of set bits below the lane -- until lo1 slots are filled.
      int cmax = -1;
#pragma unroll
      for (int j = 0; j < HJ; ++j) cmax = max(cmax, live[j] ? cnt[j] : -1);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off, 64));
      int pos = 0;
      const unsigned long long below = (1ull << lane) - 1ull;
      for (int c = cmax; c >= 0 && pos < lo1; --c) {
#pragma unroll
        for (int j = 0; j < HJ; ++j) {
          const bool mine = live[j] && cnt[j] == c;
          const unsigned long long bm = __ballot(mine);
          const int slot = pos + __popcll(bm & below);
          if (mine && slot < lo1) sel[slot] = lane + 64 * j;
          pos += __popcll(bm);
        }
      }
    }
    __syncthreads();
    double L0 = 0, L1 = 0, L2 = 0;
    bool l_inv = true;
    const bool l_live = lane < lo1 && !(VGG_TRI_ABLATE & 2);
    Cand lc;
    lc.n = 0; lc.e = 2.0 * kPi;
    {
      // the selected RANSAC point comes out of its owner's registers: hypothesis h lives in lane h % 64, register h / 64
      const int h = l_live ? sel[lane] : 0;
      const int hl = h & 63, hj = h >> 6;
      double S0 = 0, S1 = 0, S2 = 0;
      bool s_inv = true;
#pragma unroll
      for (int j = 0; j < HJ; ++j) {
        const double a0 = __shfl(X[j][0], hl, 64), a1 = __shfl(X[j][1], hl, 64), a2 = __shfl(X[j][2], hl, 64);
        const bool ai = __shfl((int)inv[j], hl, 64) != 0;
        if (hj == j) { S0 = a0; S1 = a1; S2 = a2; s_inv = ai; }
      }
      Sym4 m;
#pragma unroll
      for (int k = 0; k < 10; ++k) m.a[k] = 0.0;
      eval_views<true>(ext, tab, vlist, nv, S, S0, S1, S2, s_inv, l_live, max_rad, cos_gate, true, &m, nullptr, gv);
      double v[4];
      smallest_eigvec4_fast(m, v, l_live);
      L0 = v[0] / v[3]; L1 = v[1] / v[3]; L2 = v[2] / v[3];
      bool behind;
      // errors of the refined point (NaN -> 100*pi: not an inlier, no poisoning), cheirality over ALL views
      Cand tmp = eval_views<false>(ext, tab, vlist, nv, S, L0, L1, L2, false, l_live, max_rad, cos_gate, false, nullptr, &behind);
      const bool tri_ok = any_pair_angle(centers, S, L0, L1, L2, min_tri_deg, l_live);
      l_inv = behind || !tri_ok;
      if (!l_inv) lc = tmp;
      if (!l_live) { lc.n = 0; lc.e = 2.0 * kPi; }
    }
    __syncthreads();
    lx[4 * lane] = L0; lx[4 * lane + 1] = L1; lx[4 * lane + 2] = L2; lx[4 * lane + 3] = l_inv ? 1.0 : 0.0;
    cnts[lane] = l_live ? lc.n : -1;
    if (l_live) {
      wave_max_e = fmax(wave_max_e, lc.e);
      const double ind = (thres - lc.e) / thres + (double)lc.n;
      const int idx = H + lane;
      if (ind > best_ind || (ind == best_ind && idx < best_idx)) {
        best_ind = ind; best_idx = idx; best = lc; bX0 = L0; bX1 = L1; bX2 = L2; b_inv = l_inv;
      }
    }
    __syncthreads();
    // ---- LO round 2: stable descending rank of the lo1 LO counts, top lo2
    sel[lane] = -1;
    __syncthreads();
    if (l_live) {
      int rank = 0;
      for (int g = 0; g < lo1; ++g) { const int cg = cnts[g]; rank += (cg > lc.n) || (cg == lc.n && g < lane); }
      if (rank < lo2) sel[rank] = lane;
    }
    __syncthreads();
    {
      // kGV lanes per hypothesis (lo2 <= 10: checked by the host entry) (eval_views_grouped): slot q = lane / kGV, lane q * kGV carries the candidate
      const int q = lane / kGV, vq = lane - kGV * q, src0 = kGV * q;
      const bool q_live = q < lo2 && !(VGG_TRI_ABLATE & 2);
      const int g = q_live ? sel[q] : 0;
      Sym4 m;
#pragma unroll
      for (int k = 0; k < 10; ++k) m.a[k] = 0.0;
      eval_views_grouped<true>(ext, tab, vlist, nv, S, lx[4 * g], lx[4 * g + 1], lx[4 * g + 2], lx[4 * g + 3] != 0.0, q_live, max_rad,
                               cos_gate, false, &m, nullptr, src0, vq, gv);
      double v[4];
      smallest_eigvec4_fast(m, v, q_live);
      const double Q0 = v[0] / v[3], Q1 = v[1] / v[3], Q2 = v[2] / v[3];
      bool behind;
      Cand qc = eval_views_grouped<false>(ext, tab, vlist, nv, S, Q0, Q1, Q2, false, q_live, max_rad, cos_gate, false, nullptr, &behind,
                                          src0, vq);
      const bool tri_ok = any_pair_angle(centers, S, Q0, Q1, Q2, min_tri_deg, q_live);
      const bool q_inv = behind || !tri_ok;
      if (q_inv) { qc.n = 0; qc.e = 2.0 * kPi; }
      if (q_live && vq == 0) {
        wave_max_e = fmax(wave_max_e, qc.e);
        const double ind = (thres - qc.e) / thres + (double)qc.n;
        const int idx = H + lo1 + q;
        if (ind > best_ind || (ind == best_ind && idx < best_idx)) {
          best_ind = ind; best_idx = idx; best = qc; bX0 = Q0; bX1 = Q1; bX2 = Q2; b_inv = q_inv;
        }
      }
    }
    // ---- winner: first maximum of the indicator across lanes
    double wi = best_ind;
    int widx = best_idx;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double oi = __shfl_xor(wi, off, 64);
      const int ox = __shfl_xor(widx, off, 64);
      if (oi > wi || (oi == wi && ox < widx)) { wi = oi; widx = ox; }
    }
    const unsigned long long owner_mask = __ballot(best_idx == widx && best_ind == wi);
    const int owner = __ffsll((long long)owner_mask) - 1;
    const double W0 = __shfl(bX0, owner, 64), W1 = __shfl(bX1, owner, 64), W2 = __shfl(bX2, owner, 64);
    const int wn = __shfl(best.n, owner, 64);
    const bool w_inv = __shfl((int)b_inv, owner, 64) != 0;
    if (lane == 0) {
      out_pts[3 * (size_t)n] = W0; out_pts[3 * (size_t)n + 1] = W1; out_pts[3 * (size_t)n + 2] = W2;
      out_num[n] = wn;
    }
    // inlier mask of the winner: re-evaluate (lanes over views)
    for (int s = lane; s < S; s += 64) {
      const double* t = tab + s * kTab;
      bool isn;
      double depth;
      const double err = view_error(ext + 12 * s, t[0], t[1], t[2], W0, W1, W2, cos_gate, isn, depth);
      out_mask[(size_t)n * S + s] = (!w_inv && t[3] == 0.0 && !isn && err <= max_rad) ? 1 : 0;
    }
  }
  if (cur_chunk >= 0) {
    wave_max_e = wave_max(wave_max_e);
    if (lane == 0) atomicMax(gmax_all + cur_chunk, (unsigned long long)__double_as_longlong(wave_max_e));
  }
}

// Two-view DLT of every track between frame 0 and frame s (triangulate_by_pair, triangulation.py:45-135):
// one thread per (s, track).  The DLT matrix of a view is built with exactly the expressions of the per-view
// table of triangulate_kernel (same source expressions; golden tri_by_pair.npz is the check).
__device__ __forceinline__ void view_dlt_matrix(const double* __restrict__ P, double u, double v, Sym4& m) {
  const double nr = sqrt(u * u + v * v + 1.0);
  view_dlt_matrix_r(P, u / nr, v / nr, 1.0 / nr, m);
}

// camera centres -R^T t of all views (read by the angle tests through wave-uniform loads)
__global__ void view_centers_kernel(const double* __restrict__ ext, int S, double* __restrict__ centers) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const double* P = ext + 12 * s;
  centers[3 * s] = -(P[0] * P[3] + P[4] * P[7] + P[8] * P[11]);
  centers[3 * s + 1] = -(P[1] * P[3] + P[5] * P[7] + P[9] * P[11]);
  centers[3 * s + 2] = -(P[2] * P[3] + P[6] * P[7] + P[10] * P[11]);
}

__global__ __launch_bounds__(256) void triangulate_pairs_kernel(const double* __restrict__ ext,
                                                                const double* __restrict__ tn, int S, int N,
                                                                double* __restrict__ out) {
  const int s = blockIdx.y + 1;
  for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) {
    const double2 q0 = reinterpret_cast<const double2*>(tn)[n];
    const double2 qs = reinterpret_cast<const double2*>(tn)[(size_t)s * N + n];
    Sym4 m0, ms, m;
    view_dlt_matrix(ext, q0.x, q0.y, m0);
    view_dlt_matrix(ext + 12 * s, qs.x, qs.y, ms);
#pragma unroll
    for (int k = 0; k < 10; ++k) m.a[k] = m0.a[k] + ms.a[k];
    double v[4];
    smallest_eigvec4_fast(m, v, true);
    double* o = out + ((size_t)(s - 1) * N + n) * 3;
    o[0] = v[0] / v[3]; o[1] = v[1] / v[3]; o[2] = v[2] / v[3];
  }
}

}  // namespace vgg

using namespace vgg;

extern "C" {

static size_t tri_ws_bytes(int S, int num_chunks) {
  // per chunk: one 8-byte word (max of the mean inlier errors) + one threshold; then the camera centres [S][3]
  return 256 + 16 * (size_t)(num_chunks > 0 ? num_chunks : 1) + sizeof(double) * 3 * (size_t)(S > 0 ? S : 0);
}

size_t vgg_triangulate_workspace_bytes(int S, int N, int H, int lo_num) {
  (void)N; (void)H; (void)lo_num;
  return tri_ws_bytes(S, 1);
}

size_t vgg_triangulate_chunks_workspace_bytes(int S, int num_chunks) { return tri_ws_bytes(S, num_chunks); }

// All reference chunks of a call in ONE launch.  tracks_t (N,S,2) f64 track-major normalised rays;
// invalid_vis_conf_t (N,S) uint8; pairs (num_chunks,H,2) int32 -- the hypothesis pairs of every chunk, drawn by the
// caller in chunk order; chunk c covers tracks [c*chunk_size, min(N,(c+1)*chunk_size)).
// thresholds_io (host, [num_chunks]): in = residual-indicator threshold per chunk; out = max mean error + 1e-6
// measured per chunk (the caller re-runs when they differ).  Synchronises the stream once.
// Asynchronous core: enqueues the launch for `num_chunks` consecutive reference chunks whose first track is tracks_t[0] and
// returns -- nothing is copied back, nothing is waited for.  thresholds_dev [num_chunks] (device): residual-indicator
// threshold per chunk; gmax_dev [num_chunks] (device, zeroed by the caller): receives the bits of the largest mean inlier
// error per chunk (atomicMax on the non-negative doubles' bit patterns); centers_dev [S][3]: scratch for the camera centres.
// The host mirror draws the hypothesis pairs of a call chunk group by chunk group (torch.randperm on the host, ~0.2 ms per
// chunk at 200 views, 25 chunks) and enqueues every group as soon as its pairs are there: the draws of group g + 1 run
// while the GPU works on group g.
int vgg_triangulate_tracks_chunks_enqueue(const double* extrinsics, const double* tracks_t, const uint8_t* invalid_vis_conf_t,
                                          const int32_t* pairs, int S, int N, int H, int num_chunks, int chunk_size, int lo_num,
                                          double max_angular_error_deg, double min_tri_angle_deg, double* out_points,
                                          int64_t* out_inlier_num, uint8_t* out_inlier_mask, const double* thresholds_dev,
                                          unsigned long long* gmax_dev, double* centers_dev, void* stream) {
  if (S < 2 || N < 0 || H < 1 || H > 256 || lo_num < 1 || lo_num > 64 || !thresholds_dev || !gmax_dev || !centers_dev ||
      num_chunks < 1 || num_chunks > 4096 || chunk_size < 1 || (long)num_chunks * chunk_size < N)
    return VGG_ERR_INVALID_ARGUMENT;
  if (N == 0) return VGG_OK;
  hipStream_t st = (hipStream_t)stream;
  const int lo1 = lo_num < H ? lo_num : H;
  const int lo2 = lo1 < 10 ? lo1 : 10;                   // (<= 64 / kGV: six lanes per hypothesis in the second round)
  const double max_rad = max_angular_error_deg * (kPi / 180.0);
  const size_t lds = sizeof(double) * ((size_t)S * kTab + 64 * 4) + sizeof(int) * (((H + 63) / 64) * 64 + 64 + (((size_t)S + 1) & ~(size_t)1)) +
                     sizeof(double) * kGvMax * 10;
  if (lds > 160 * 1024) return VGG_ERR_UNSUPPORTED;
  view_centers_kernel<<<div_up(S, 64), 64, 0, st>>>(extrinsics, S, centers_dev);
  const int grid = N < 256 * 32 ? N : 256 * 32;
  void (*kern)(const double*, const double*, const uint8_t*, const int32_t*, int, int, int, int, int, double, double,
               const double*, int, double*, int64_t*, uint8_t*, unsigned long long*, const double*) =
      (H <= 64) ? triangulate_kernel<1> : (H <= 128) ? triangulate_kernel<2> : triangulate_kernel<4>;
  if (lds > 64 * 1024) VGG_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  kern<<<grid, 64, lds, st>>>(extrinsics, tracks_t, invalid_vis_conf_t, pairs, S, N, H, lo1, lo2, max_rad, min_tri_angle_deg,
                              thresholds_dev, chunk_size, out_points, out_inlier_num, out_inlier_mask, gmax_dev, centers_dev);
  VGG_LAUNCH_CHECK();
  return VGG_OK;
}

int vgg_triangulate_tracks_chunks(const double* extrinsics, const double* tracks_t, const uint8_t* invalid_vis_conf_t,
                                  const int32_t* pairs, int S, int N, int H, int num_chunks, int chunk_size, int lo_num,
                                  double max_angular_error_deg, double min_tri_angle_deg, double* out_points,
                                  int64_t* out_inlier_num, uint8_t* out_inlier_mask, double* thresholds_io,
                                  void* workspace, void* stream) {
  if (!thresholds_io || !workspace || num_chunks < 1 || num_chunks > 4096) return VGG_ERR_INVALID_ARGUMENT;
  if (N == 0) return VGG_OK;
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* gmax = (unsigned long long*)((char*)workspace + 256);
  double* thres = (double*)(gmax + num_chunks);
  double* centers = thres + num_chunks;
  VGG_HIP_CHECK(hipMemsetAsync(gmax, 0, sizeof(unsigned long long) * num_chunks, st));
  VGG_HIP_CHECK(hipMemcpyAsync(thres, thresholds_io, sizeof(double) * num_chunks, hipMemcpyHostToDevice, st));
  const int rc = vgg_triangulate_tracks_chunks_enqueue(extrinsics, tracks_t, invalid_vis_conf_t, pairs, S, N, H, num_chunks, chunk_size,
                                                       lo_num, max_angular_error_deg, min_tri_angle_deg, out_points, out_inlier_num,
                                                       out_inlier_mask, thres, gmax, centers, stream);
  if (rc != VGG_OK) return rc;
  unsigned long long bits[4096];
  VGG_HIP_CHECK(hipMemcpyAsync(bits, gmax, sizeof(unsigned long long) * num_chunks, hipMemcpyDeviceToHost, st));
  VGG_HIP_CHECK(hipStreamSynchronize(st));
  for (int c = 0; c < num_chunks; ++c) {
    double m;
    memcpy(&m, &bits[c], sizeof(double));
    thresholds_io[c] = m + 1e-6;
  }
  return VGG_OK;
}

// One chunk (the reference's triangulate_tracks_single_chunk): pairs (H,2), *threshold_io as above.
int vgg_triangulate_tracks(const double* extrinsics, const double* tracks_t, const uint8_t* invalid_vis_conf_t,
                           const int32_t* pairs, int S, int N, int H, int lo_num, double max_angular_error_deg,
                           double min_tri_angle_deg, double* out_points, int64_t* out_inlier_num,
                           uint8_t* out_inlier_mask, double* threshold_io, void* workspace, void* stream) {
  if (N < 0) return VGG_ERR_INVALID_ARGUMENT;
  return vgg_triangulate_tracks_chunks(extrinsics, tracks_t, invalid_vis_conf_t, pairs, S, N, H, 1, N > 0 ? N : 1, lo_num,
                                       max_angular_error_deg, min_tri_angle_deg, out_points, out_inlier_num,
                                       out_inlier_mask, threshold_io, workspace, stream);
}

// extrinsics (S,3,4) f64, tracks_normalized (S,N,2) f64 frame-major -> out_points (S-1,N,3): the two-view DLT
// point of every track between frame 0 and frame s.
int vgg_triangulate_by_pair(const double* extrinsics, const double* tracks_normalized, int S, int N, double* out_points,
                            void* stream) {
  if (S < 2 || N < 0 || !extrinsics || !tracks_normalized || !out_points) return VGG_ERR_INVALID_ARGUMENT;
  if (N == 0) return VGG_OK;
  const dim3 grid(min(div_up(N, 256), 1024), S - 1);
  triangulate_pairs_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(extrinsics, tracks_normalized, S, N, out_points);
  VGG_LAUNCH_CHECK();
  return VGG_OK;
}

}  // extern "C"
