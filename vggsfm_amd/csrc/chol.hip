// Dense fp64 Cholesky solve of the reduced camera system (6C+K unknowns) on gfx950.
//
// In the reference this is the DENSE_SCHUR / SPARSE_SCHUR factorisation inside Ceres, reached through
// pycolmap.bundle_adjustment (vggsfm/utils/triangulation.py:213,1050,1142).  Two paths:
//   * n >= 128 with the right-hand side stored behind the matrix (what bundle adjustment passes): ONE launch, a
//     workgroup per 64 x 64 tile, left-looking, tiles handed over through per-tile flags -- chol_dataflow_kernel and
//     chol_backward_dataflow_kernel in the second half of this file;
//   * otherwise (and with VGG_CHOL_LEGACY=1): the multi-launch form described next -- right-looking blocked
//     Cholesky, two launches per block column of NB columns:
//   panel  : every workgroup re-factors the NB x NB diagonal block in LDS (outer-product form, ONE barrier per
//            column, reciprocal instead of divide), then solves the panel rows below by forward
//            substitution, one row per lane with the row in NB registers; one extra workgroup pushes the
//            identity through the same substitution: T_k = L_kk^-T for the backward solve;
//   update : trailing SYRK on the matrix cores, one wavefront per 32x32 tile =
//            2x2 v_mfma_f64_16x16x4_f64 accumulators x NB/4 k-steps.
// The whole thing is a chain of dependent launches, each with a floor of ~4.5 us plus two memory round trips, and
// a ~250 ns pivot step per column.  The kernels are templates on NB; NB = 64 (half the launches) was measured
// and lost: its panel kernel takes 84 us against 2 x 18 us (the 64 x 64 factor and the 2016 broadcast LDS reads
// of the substitution are LDS-issue bound), so NB = 32 is what runs.
// The right-hand side is stored as row n of the (n+1) x n array, so the factorisation performs the
// forward substitution on the way; the backward substitution is a sequence of NB x NB mat-vecs with T_k.
// Only the lower triangle (row-major, ld = n) is read or written.
#include <cstdlib>

#include "common.hpp"

namespace vgg {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// 1/x from the hardware estimate (measured max relative error 4.6e-8, scripts/ubench/rcp_accuracy.hip) + Newton
// steps: one step leaves 2.2e-15, two 1.1e-16.  The pivot reciprocal sits on the per-column critical path of the
// factorisation, where one step (a few ulp on L) is enough; everything else uses two.
template <int STEPS = 2>
__device__ __forceinline__ double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
#pragma unroll
  for (int i = 0; i < STEPS; ++i) r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}

// The 256 threads of a workgroup factor the NB x NB block D (LDS, ld = NB + 1, rows >= nb padded with the
// identity) in place.  Outer-product form WITHOUT normalising inside the loop (a_ic -= a_ij a_cj / d_j), so
// a step needs a single barrier: the column it reads was finalised by the previous step, the elements it
// writes are disjoint from it.  Columns are scaled by 1/sqrt(d_j) at the end; rdiag[j] = 1 / L_jj.
template <int NB>
__device__ __forceinline__ void factor_diag_lds(double* D, double* rdiag, int32_t* fail_flag) {
  constexpr int LD = NB + 1, STRIDE = 256 / NB, CNT = NB / STRIDE;
  const int tid = threadIdx.x;
  const int c = tid % NB, i0 = tid / NB;        // thread owns elements (i0 + STRIDE m, c), m = 0..CNT-1
  bool bad = false;
  for (int j = 0; j < NB - 1; ++j) {
    __syncthreads();
    const double dj = D[j * LD + j];
    if (!(dj > 0.0) || !(dj < 1.7976931348623157e308)) bad = true;
    const double inv = fast_rcp<1>((dj > 0.0) ? dj : 1.0);
    if (c > j) {
      const double lcj = D[c * LD + j] * inv;
#pragma unroll
      for (int m = 0; m < CNT; ++m) {
        const int i = i0 + STRIDE * m;
        if (i >= c) D[i * LD + c] -= D[i * LD + j] * lcj;
      }
    }
  }
  __syncthreads();
  {
    const double dl = D[(NB - 1) * LD + NB - 1];
    if (!(dl > 0.0) || !(dl < 1.7976931348623157e308)) bad = true;
  }
  // scale column c by 1/sqrt(d_c)
  const double dc = D[c * LD + c];
  const double sd = sqrt((dc > 0.0) ? dc : 1.0);
  const double rs = 1.0 / sd;
  __syncthreads();
#pragma unroll
  for (int m = 0; m < CNT; ++m) {
    const int i = i0 + STRIDE * m;
    if (i > c) D[i * LD + c] *= rs;
    else if (i == c) { D[i * LD + c] = sd; rdiag[c] = rs; }
  }
  if (bad && tid == 0 && fail_flag) *fail_flag = 1;
  __syncthreads();
}

// The same factorisation FOUR pivot columns per synchronisation.  The pivot chain of the column-at-a-time version costs
// a barrier + an LDS round trip + a reciprocal per column (~280 ns); here every thread factors the 4 x 4 pivot block
// redundantly in registers (broadcast LDS reads, four dependent reciprocals), one thread per row pushes its 4 panel
// entries through it (u_r0..u_r3, the unnormalised multipliers, and their products with the pivot reciprocals) into a
// small LDS panel, and the trailing update of an element is one rank-4 step: three barriers per four columns.
// Arithmetic is that of factor_diag_lds, operation for operation (a_ic -= a_ij (a_cj / d_j), j ascending, columns scaled
// by 1 / sqrt(d_c) at the end): the factor is bit-identical.
// WITH_T: the rows of the identity in Tl (LDS, ld = NB + 1) receive the same column operations, so that Tl ends up as
// L^-T (what the panel product and the backward substitution need) without a second, dependent pass.
// scratch: 2 * (WITH_T ? 2 : 1) * NB * 4 doubles of LDS.
template <int NB, bool WITH_T, int LD = NB + 1>
__device__ __forceinline__ void factor_diag_lds4(double* D, double* Tl, double* rdiag, double* scratch, int32_t* fail_flag) {
  static_assert(NB % 4 == 0 && 256 % NB == 0, "block of 4-column steps, 256 threads");
  constexpr int STRIDE = 256 / NB, CNT = NB / STRIDE, ROWS = WITH_T ? 2 * NB : NB;
  double* U = scratch;                     // [ROWS][4] unnormalised panel entries u_rt
  double* V = scratch + ROWS * 4;          // [ROWS][4] u_rt / d_t
  const int tid = threadIdx.x;
  const int c = tid % NB, i0 = tid / NB;
  bool bad = false;
  if (WITH_T) {
    for (int e = tid; e < NB * NB; e += 256) Tl[(e / NB) * LD + (e % NB)] = (e / NB == e % NB) ? 1.0 : 0.0;
  }
  for (int j0 = 0; j0 < NB; j0 += 4) {
    __syncthreads();
    // 4 x 4 pivot block, every thread (same addresses: LDS broadcast)
    const double* P = D + j0 * LD + j0;
    const double d0 = P[0];
    const double u10 = P[LD], p11 = P[LD + 1];
    const double u20 = P[2 * LD], p21 = P[2 * LD + 1], p22 = P[2 * LD + 2];
    const double u30 = P[3 * LD], p31 = P[3 * LD + 1], p32 = P[3 * LD + 2], p33 = P[3 * LD + 3];
    const double r0 = fast_rcp<1>((d0 > 0.0) ? d0 : 1.0);
    const double m10 = u10 * r0, m20 = u20 * r0, m30 = u30 * r0;        // multipliers a_cj / d_j
    const double d1 = p11 - u10 * m10;
    const double u21 = p21 - u20 * m10, u31 = p31 - u30 * m10;
    const double r1 = fast_rcp<1>((d1 > 0.0) ? d1 : 1.0);
    const double m21 = u21 * r1, m31 = u31 * r1;
    const double d2 = (p22 - u20 * m20) - u21 * m21;
    const double u32 = (p32 - u30 * m20) - u31 * m21;
    const double r2 = fast_rcp<1>((d2 > 0.0) ? d2 : 1.0);
    const double m32 = u32 * r2;
    const double d3 = ((p33 - u30 * m30) - u31 * m31) - u32 * m32;
    const double r3 = fast_rcp<1>((d3 > 0.0) ? d3 : 1.0);
    constexpr double HUGE_ = 1.7976931348623157e308;
    if (!(d0 > 0.0) || !(d0 < HUGE_) || !(d1 > 0.0) || !(d1 < HUGE_) || !(d2 > 0.0) || !(d2 < HUGE_) || !(d3 > 0.0) || !(d3 < HUGE_))
      bad = true;
    // one thread per row: the row's four panel entries through the pivot block
    if (tid < ROWS) {
      const bool trow = WITH_T && tid >= NB;
      const int r = trow ? tid - NB : tid;
      const double* X = (trow ? Tl : D) + r * LD + j0;
      const bool live = trow ? (r < j0 + 4) : (r >= j0 + 4);      // rows that have entries in (or below) this panel
      double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0;
      if (live) {
        x0 = X[0];
        x1 = X[1] - x0 * m10;
        x2 = (X[2] - x0 * m20) - x1 * m21;
        x3 = ((X[3] - x0 * m30) - x1 * m31) - x2 * m32;
      }
      double* Ur = U + tid * 4;
      double* Vr = V + tid * 4;
      Ur[0] = x0; Ur[1] = x1; Ur[2] = x2; Ur[3] = x3;
      Vr[0] = x0 * r0; Vr[1] = x1 * r1; Vr[2] = x2 * r2; Vr[3] = x3 * r3;
    }
    __syncthreads();
    // rank-4 trailing update (columns beyond the panel): D lower triangle, and every live row of Tl
    if (c >= j0 + 4) {
      const double v0 = V[c * 4], v1 = V[c * 4 + 1], v2 = V[c * 4 + 2], v3 = V[c * 4 + 3];
#pragma unroll
      for (int m = 0; m < CNT; ++m) {
        const int i = i0 + STRIDE * m;
        if (i >= c) {
          const double* Ui = U + i * 4;
          double a = D[i * LD + c];
          a -= Ui[0] * v0; a -= Ui[1] * v1; a -= Ui[2] * v2; a -= Ui[3] * v3;
          D[i * LD + c] = a;
        }
        if (WITH_T && i < j0 + 4) {
          const double* Ui = U + (NB + i) * 4;
          double a = Tl[i * LD + c];
          a -= Ui[0] * v0; a -= Ui[1] * v1; a -= Ui[2] * v2; a -= Ui[3] * v3;
          Tl[i * LD + c] = a;
        }
      }
    }
    // the panel columns themselves: rows below the pivot block get u_rt; the pivot block its own entries
    if (tid < ROWS) {
      const bool trow = WITH_T && tid >= NB;
      const int r = trow ? tid - NB : tid;
      double* X = (trow ? Tl : D) + r * LD + j0;
      const double* Ur = U + tid * 4;
      if (trow ? (r < j0 + 4) : (r >= j0 + 4)) { X[0] = Ur[0]; X[1] = Ur[1]; X[2] = Ur[2]; X[3] = Ur[3]; }
    }
    if (tid == 0) {
      double* Pw = D + j0 * LD + j0;
      Pw[LD + 1] = d1; Pw[2 * LD + 1] = u21; Pw[2 * LD + 2] = d2; Pw[3 * LD + 1] = u31; Pw[3 * LD + 2] = u32; Pw[3 * LD + 3] = d3;
    }
  }
  __syncthreads();
  // scale column c by 1/sqrt(d_c)
  const double dc = D[c * LD + c];
  const double sd = sqrt((dc > 0.0) ? dc : 1.0);
  const double rs = 1.0 / sd;
  __syncthreads();
#pragma unroll
  for (int m = 0; m < CNT; ++m) {
    const int i = i0 + STRIDE * m;
    if (i > c) D[i * LD + c] *= rs;
    else if (i == c) { D[i * LD + c] = sd; rdiag[c] = rs; }
    if (WITH_T && i <= c) Tl[i * LD + c] *= rs;
  }
  if (bad && tid == 0 && fail_flag) *fail_flag = 1;
  __syncthreads();
}

// factor_diag_lds4 with LOOK-AHEAD: the pivot chain of a step (four dependent reciprocals, ~700 cycles that every thread
// used to repeat before anything else could start) is computed for step j + 1 by the LAST wavefront while the other three
// apply the trailing update of step j.  That wavefront first brings the next 4 x 4 pivot block up to date itself (the same
// rank-4 expressions, in the same order, that the trailing update would have applied -- the factor stays bit-identical),
// runs the chain and leaves the 13 numbers every thread needs (reciprocals, multipliers, eliminated block entries) in a
// small LDS record.  Per step: barrier, read the record, panel rows, barrier, {chain | update}.
// MEASURED SLOWER (opt-in, -DVGG_CHOL_LOOKAHEAD): c3 factorisation 0.505 -> 0.548 ms, c2 0.208 -> 0.221, a c4 shard 1.36 ->
// 1.46.  The trailing update it takes off the critical path is the cheap part of a step; the chain + the panel rows are
// the step, and the record adds LDS round trips (record write -> barrier -> read; U / V / D reads of the chain wavefront)
// to exactly that path.  Same factor (tests/test_gpu_ba.py::test_cholesky_* pass with it).
// scratch: 2 * ROWS * 4 doubles (U, V) + 16 (pivot record).
template <int NB, bool WITH_T, int LD = NB + 1>
__device__ __forceinline__ void factor_diag_lds4_la(double* D, double* Tl, double* rdiag, double* scratch, int32_t* fail_flag) {
  static_assert(NB % 4 == 0 && NB <= 64, "block of 4-column steps");
  constexpr int ROWS = WITH_T ? 2 * NB : NB;
  constexpr int UT = 192;                          // threads of the trailing update (wavefronts 0..2)
  constexpr int STRIDE = UT / NB, CNT = (NB + STRIDE - 1) / STRIDE;
  static_assert(UT % NB == 0 && ROWS <= UT, "update layout");
  double* U = scratch;                             // [ROWS][4] unnormalised panel entries u_rt
  double* V = scratch + ROWS * 4;                  // [ROWS][4] u_rt / d_t
  double* PR = scratch + 2 * ROWS * 4;             // pivot record: r0..r3, m10 m20 m30 m21 m31 m32, d1 u21 d2 u31 u32 d3 (16)
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = tid % NB, i0 = tid / NB;           // (update threads only: tid < UT)
  bool bad = false;
  constexpr double HUGE_ = 1.7976931348623157e308;
  if (WITH_T) {
    for (int e = tid; e < NB * NB; e += 256) Tl[(e / NB) * LD + (e % NB)] = (e / NB == e % NB) ? 1.0 : 0.0;
  }
  // pivot chain of the 4 x 4 block (values given) -> record; every lane of the calling wavefront computes the same
  auto chain = [&](double d0, double u10, double p11, double u20, double p21, double p22, double u30, double p31, double p32, double p33) {
    const double r0 = fast_rcp<1>((d0 > 0.0) ? d0 : 1.0);
    const double m10 = u10 * r0, m20 = u20 * r0, m30 = u30 * r0;
    const double d1 = p11 - u10 * m10;
    const double u21 = p21 - u20 * m10, u31 = p31 - u30 * m10;
    const double r1 = fast_rcp<1>((d1 > 0.0) ? d1 : 1.0);
    const double m21 = u21 * r1, m31 = u31 * r1;
    const double d2 = (p22 - u20 * m20) - u21 * m21;
    const double u32 = (p32 - u30 * m20) - u31 * m21;
    const double r2 = fast_rcp<1>((d2 > 0.0) ? d2 : 1.0);
    const double m32 = u32 * r2;
    const double d3 = ((p33 - u30 * m30) - u31 * m31) - u32 * m32;
    const double r3 = fast_rcp<1>((d3 > 0.0) ? d3 : 1.0);
    if (!(d0 > 0.0) || !(d0 < HUGE_) || !(d1 > 0.0) || !(d1 < HUGE_) || !(d2 > 0.0) || !(d2 < HUGE_) || !(d3 > 0.0) || !(d3 < HUGE_))
      bad = true;
    if ((tid & 63) == 0) {
      PR[0] = r0; PR[1] = r1; PR[2] = r2; PR[3] = r3;
      PR[4] = m10; PR[5] = m20; PR[6] = m30; PR[7] = m21; PR[8] = m31; PR[9] = m32;
      PR[10] = d1; PR[11] = u21; PR[12] = d2; PR[13] = u31; PR[14] = u32; PR[15] = d3;
    }
  };
  __syncthreads();
  if (wave == 3) {                                 // record of step 0: the block as it stands
    const double* P = D;
    chain(P[0], P[LD], P[LD + 1], P[2 * LD], P[2 * LD + 1], P[2 * LD + 2], P[3 * LD], P[3 * LD + 1], P[3 * LD + 2], P[3 * LD + 3]);
  }
  for (int j0 = 0; j0 < NB; j0 += 4) {
    __syncthreads();                               // record of this step written; all earlier updates applied
    const double r0 = PR[0], r1 = PR[1], r2 = PR[2], r3 = PR[3];
    const double m10 = PR[4], m20 = PR[5], m30 = PR[6], m21 = PR[7], m31 = PR[8], m32 = PR[9];
    const double d1 = PR[10], u21 = PR[11], d2 = PR[12], u31 = PR[13], u32 = PR[14], d3 = PR[15];
    // one thread per row: the row's four panel entries through the pivot block
    if (tid < ROWS) {
      const bool trow = WITH_T && tid >= NB;
      const int r = trow ? tid - NB : tid;
      const double* X = (trow ? Tl : D) + r * LD + j0;
      const bool live = trow ? (r < j0 + 4) : (r >= j0 + 4);
      double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0;
      if (live) {
        x0 = X[0];
        x1 = X[1] - x0 * m10;
        x2 = (X[2] - x0 * m20) - x1 * m21;
        x3 = ((X[3] - x0 * m30) - x1 * m31) - x2 * m32;
      }
      double* Ur = U + tid * 4;
      double* Vr = V + tid * 4;
      Ur[0] = x0; Ur[1] = x1; Ur[2] = x2; Ur[3] = x3;
      Vr[0] = x0 * r0; Vr[1] = x1 * r1; Vr[2] = x2 * r2; Vr[3] = x3 * r3;
    }
    __syncthreads();                               // panel in place; everybody has read the record
    const int jn = j0 + 4;                         // first column of the NEXT pivot block
    if (wave == 3) {
      if (jn < NB) {
        // next pivot block: apply this step's rank-4 update to its 10 entries (as the trailing update does), store, chain
        double q[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b <= a; ++b) {
            const double* Ua = U + (jn + a) * 4;
            const double* Vb = V + (jn + b) * 4;
            double v = D[(jn + a) * LD + jn + b];
            v -= Ua[0] * Vb[0]; v -= Ua[1] * Vb[1]; v -= Ua[2] * Vb[2]; v -= Ua[3] * Vb[3];
            q[a][b] = v;
          }
        if ((tid & 63) == 0) {
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) D[(jn + a) * LD + jn + b] = q[a][b];
        }
        chain(q[0][0], q[1][0], q[1][1], q[2][0], q[2][1], q[2][2], q[3][0], q[3][1], q[3][2], q[3][3]);
      }
    } else {
      // rank-4 trailing update (columns beyond the panel; the next pivot block belongs to the other wavefront)
      if (c >= jn) {
        const double v0 = V[c * 4], v1 = V[c * 4 + 1], v2 = V[c * 4 + 2], v3 = V[c * 4 + 3];
#pragma unroll
        for (int m = 0; m < CNT; ++m) {
          const int i = i0 + STRIDE * m;
          if (i < NB && i >= c && !(i < jn + 4 && c < jn + 4)) {
            const double* Ui = U + i * 4;
            double a = D[i * LD + c];
            a -= Ui[0] * v0; a -= Ui[1] * v1; a -= Ui[2] * v2; a -= Ui[3] * v3;
            D[i * LD + c] = a;
          }
          if (WITH_T && i < NB && i < jn) {
            const double* Ui = U + (NB + i) * 4;
            double a = Tl[i * LD + c];
            a -= Ui[0] * v0; a -= Ui[1] * v1; a -= Ui[2] * v2; a -= Ui[3] * v3;
            Tl[i * LD + c] = a;
          }
        }
      }
      // the panel columns themselves: rows below the pivot block get u_rt; the pivot block its own entries
      if (tid < ROWS) {
        const bool trow = WITH_T && tid >= NB;
        const int r = trow ? tid - NB : tid;
        double* X = (trow ? Tl : D) + r * LD + j0;
        const double* Ur = U + tid * 4;
        if (trow ? (r < j0 + 4) : (r >= j0 + 4)) { X[0] = Ur[0]; X[1] = Ur[1]; X[2] = Ur[2]; X[3] = Ur[3]; }
      }
      if (tid == 0) {
        double* Pw = D + j0 * LD + j0;
        Pw[LD + 1] = d1; Pw[2 * LD + 1] = u21; Pw[2 * LD + 2] = d2; Pw[3 * LD + 1] = u31; Pw[3 * LD + 2] = u32; Pw[3 * LD + 3] = d3;
      }
    }
  }
  __syncthreads();
  // scale column c by 1/sqrt(d_c): all 256 threads again
  constexpr int STRIDE2 = 256 / NB, CNT2 = NB / STRIDE2;
  const int c2 = tid % NB, i02 = tid / NB;
  const double dc = D[c2 * LD + c2];
  const double sd = sqrt((dc > 0.0) ? dc : 1.0);
  const double rs = 1.0 / sd;
  __syncthreads();
#pragma unroll
  for (int m = 0; m < CNT2; ++m) {
    const int i = i02 + STRIDE2 * m;
    if (i > c2) D[i * LD + c2] *= rs;
    else if (i == c2) { D[i * LD + c2] = sd; rdiag[c2] = rs; }
    if (WITH_T && i <= c2) Tl[i * LD + c2] *= rs;
  }
  if (bad && (tid & 63) == 0 && fail_flag) *fail_flag = 1;
  __syncthreads();
}

// One double of lane `src` (compile-time constant after unrolling) as a wave-uniform value: two v_readlane_b32.
__device__ __forceinline__ double readlane_f64(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// The same broadcast through the LDS crossbar (ds_bpermute_b32: no LDS memory involved): the value lands in a VECTOR register.
// v_readlane writes a scalar pair, and an fp64 instruction on gfx9 takes ONE scalar operand -- a step of factor16_mfma spent 14
// v_mov_b32 on carrying broadcast values back into vector registers (round 6; VGG_F16_BCAST).  `zero` = a vector register
// holding 0 (the lane address goes into the instruction's offset field).
__device__ __forceinline__ double bpermute_f64(double v, int src, int zero) {
  const int lo = __builtin_amdgcn_ds_bpermute(zero + 4 * src, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(zero + 4 * src, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

// The same factorisation by ONE wavefront with the block in registers: lane i < 32 holds row i of the (symmetric) block,
// no LDS round trip and no barrier per pivot column -- the multipliers of a column are broadcast with v_readlane.
// Per column j: d_j = lane j's x[j]; m = x[j] / d_j (every lane: lane c now holds the multiplier of column c);
// x[c] -= x[j] * m_c for c > j.  Arithmetic and order are those of factor_diag_lds (non-normalised outer product,
// one Newton step on the pivot reciprocal, columns scaled by 1/sqrt(d_c) at the end): the factor is bit-identical.
// (Opt-in experiment, see FACTOR_DIAG below: it measured no faster than the LDS version.)
// D: LDS block (ld = NB + 1), lower triangle valid on entry, factor on exit; rdiag[j] = 1 / L_jj.
// Must be called by all 64 lanes of one wavefront; the caller synchronises the workgroup afterwards.
template <int NB>
__device__ __forceinline__ void factor_diag_wave(double* D, double* rdiag, int32_t* fail_flag) {
  static_assert(NB == 32, "one row per lane, lanes 0..31");
  constexpr int LD = NB + 1;
  const int lane = threadIdx.x & 63;
  const int i = lane & (NB - 1);
  double x[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) x[c] = D[(c <= i ? i * LD + c : c * LD + i)];      // full symmetric row
  bool bad = false;
  double myd = 1.0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const double dj = readlane_f64(x[j], j);
    if (!(dj > 0.0) || !(dj < 1.7976931348623157e308)) bad = true;
    if (i == j) myd = x[j];
    if (j < NB - 1) {
      const double inv = fast_rcp<1>((dj > 0.0) ? dj : 1.0);
      const double m = x[j] * inv;
      // all multipliers of the column first (distinct scalar registers), then the FMAs: back-to-back v_readlane pipeline,
      // whereas readlane -> fma pairs through one scalar pair wait for the scalar write every time
      double sc[NB];
#pragma unroll
      for (int c = j + 1; c < NB; ++c) sc[c] = readlane_f64(m, c);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = j + 1; c < NB; ++c) x[c] -= x[j] * sc[c];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // scale column c by 1/sqrt(d_c): lane c knows d_c; broadcast the 32 scales through LDS (rdiag doubles as staging)
  const double sd = sqrt((myd > 0.0) ? myd : 1.0);
  const double rs = 1.0 / sd;
  if (lane < NB) rdiag[lane] = rs;
  __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the ds_write above is visible to this wave's reads below
  __builtin_amdgcn_wave_barrier();
  if (lane < NB) {
#pragma unroll
    for (int c = 0; c < NB; ++c) {
      if (c < i) D[i * LD + c] = x[c] * rdiag[c];
      else if (c == i) D[i * LD + c] = sd;
    }
  }
  if (bad && lane == 0 && fail_flag) *fail_flag = 1;
}

// x <- x L_kk^-T for one row held in registers.  Column-oriented substitution: once x_k is final it is
// eliminated from all later columns with independent FMAs, so the dependent chain is one multiply + one FMA
// per column (an fp64 FMA has a 32-cycle dependent latency on gfx950) instead of an NB(NB+1)/2-long chain.
template <int NB>
__device__ __forceinline__ void substitute_row(double (&x)[NB], const double* D, const double* rdiag) {
  constexpr int LD = NB + 1;
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    x[k] *= rdiag[k];
#pragma unroll
    for (int c = k + 1; c < NB; ++c) x[c] -= x[k] * D[c * LD + k];
  }
}

// Diagonal-block factorisation used by the panel kernels: the 256-thread LDS version.  -DVGG_CHOL_WAVE_FACTOR selects the
// single-wavefront register version (wavefront 2 factors, the workgroup waits) -- built and measured in round 2
// (scripts/ubench/chol_bench, n = 1202): 1.047 ms against 1.057 ms, i.e. no gain: ~1000 v_readlane_b32 per 32 x 32 block
// cost what the 32 barrier + LDS round trips cost.
#if defined(VGG_CHOL_COLUMN_FACTOR)
#define FACTOR_DIAG(Dp, rdp, failp) factor_diag_lds<NB>(Dp, rdp, failp)
#elif !defined(VGG_CHOL_WAVE_FACTOR)
#define FACTOR_DIAG(Dp, rdp, failp) factor_diag_lds4<NB, false>(Dp, nullptr, rdp, fscr, failp)
#else
#define FACTOR_DIAG(Dp, rdp, failp)                                          \
  do {                                                                       \
    __syncthreads();                                                         \
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 2) {             \
      if constexpr (NB == 32) factor_diag_wave<32>(Dp, rdp, failp);          \
    }                                                                        \
    if constexpr (NB != 32) factor_diag_lds<NB>(Dp, rdp, failp);             \
    __syncthreads();                                                         \
  } while (0)
#endif

template <int NB>
__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ A, int n, int nrows, int k0,
                                                         int32_t* fail, const int32_t* skip,
                                                         double* __restrict__ inv_blocks,
                                                         const double* __restrict__ S2) {
  constexpr int LD = NB + 1;
  __shared__ double D[NB * LD];
  __shared__ double rdiag[NB];
  __shared__ double fscr[8 * NB];
  if (skip && *skip) return;
  const int nb = min(NB, n - k0);
  const int tid = threadIdx.x;
  // panel row of this lane: issue its loads before the factorisation so that their latency overlaps it
  const bool extra_wg = inv_blocks && blockIdx.x == gridDim.x - 1;
  const int row = k0 + nb + blockIdx.x * 256 + tid;
  const bool has_row = !extra_wg && row < nrows && nb == NB;
  double x[NB];
  if (has_row) {
    const double* Arow = A + (size_t)row * n + k0;
#pragma unroll
    for (int c = 0; c < NB; ++c) x[c] = Arow[c];   // unconditional: all NB loads in flight at once
    if (S2 && row < n) {                           // lazily added overlapped-batch contributions (see panel2)
      const double* Srow = S2 + (size_t)row * n + k0;
#pragma unroll
      for (int c = 0; c < NB; ++c) x[c] += Srow[c];
    }
  } else {
#pragma unroll
    for (int c = 0; c < NB; ++c) x[c] = (extra_wg && c == tid) ? 1.0 : 0.0;
  }
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e / NB, c = e % NB;
    const size_t o = (size_t)(k0 + i) * n + k0 + c;
    D[i * LD + c] = (i < nb && c <= i) ? A[o] + (S2 ? S2[o] : 0.0) : ((i == c) ? 1.0 : 0.0);
  }
  FACTOR_DIAG(D, rdiag, (blockIdx.x == 0) ? fail : nullptr);
  if (blockIdx.x == 0) {
    for (int e = tid; e < NB * NB; e += 256) {
      const int i = e / NB, c = e % NB;
      if (i < nb && c <= i) A[(size_t)(k0 + i) * n + k0 + c] = D[i * LD + c];
    }
  }
  if (extra_wg) {
    // extra workgroup: T = L_kk^-T (rows of the identity pushed through the same substitution), used by the
    // backward solve as a plain NB x NB mat-vec instead of an NB-step dependent chain.  Row-major NB x NB.
    if (tid < NB) {
      substitute_row<NB>(x, D, rdiag);
      double* T = inv_blocks + (size_t)(k0 / NB) * NB * NB + tid * NB;
#pragma unroll
      for (int c = 0; c < NB; ++c) T[c] = x[c];
    }
    return;
  }
  if (row >= nrows) return;
  double* Arow = A + (size_t)row * n + k0;
  if (nb < NB) {                                // ragged last block: only the appended rhs row sits below it
    for (int c = 0; c < nb; ++c) {
      double sacc = Arow[c];
      for (int k = 0; k < c; ++k) sacc -= Arow[k] * D[c * LD + k];
      Arow[c] = sacc * rdiag[c];
    }
    return;
  }
  substitute_row<NB>(x, D, rdiag);
#pragma unroll
  for (int c = 0; c < NB; ++c) Arow[c] = x[c];
}

// Two consecutive block columns (64 columns) in ONE launch: identical arithmetic to two chol_panel_kernel<32> steps
// with the rank-32 update of the second block column folded in, so that a 64-column step costs one panel launch
// + one K = 64 trailing update instead of two of each (every launch has a ~4.5 us floor plus two dependent
// memory round trips).  Requires k0 + 64 <= n.  Workgroup = 256 threads for the two 32x32 factorisations;
// wavefront 0 owns 64 panel rows (one per lane, 64 registers), wavefront 1 solves the 32 rows of L21.
// blockIdx.y selects one of TWO independent panels (k0 or k0_second) of the same launch: when the leading part of the
// matrix is block diagonal (column sets A and B with A[B rows][A cols] = 0, see cholesky_solve_enqueue) the panels of
// the two blocks do not depend on each other and their ~40 us pivot chains run side by side.
__global__ __launch_bounds__(256) void chol_panel2_kernel(double* __restrict__ A, int n, int nrows, int k0_first,
                                                          int32_t* fail, const int32_t* skip,
                                                          double* __restrict__ inv_blocks,
                                                          const double* __restrict__ S2, int k0_second) {
  constexpr int NB = 32, LD = NB + 1;
  const int k0 = blockIdx.y ? k0_second : k0_first;
  __shared__ double D1[NB * LD], D2[NB * LD], L21[NB * LD];
  __shared__ double rd1[NB], rd2[NB];
  __shared__ double fscr[8 * NB];
  if (skip && *skip) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const bool extra_wg = blockIdx.x == gridDim.x - 1;
  const int row = k0 + 2 * NB + blockIdx.x * 64 + lane;
  const bool has_row = !extra_wg && wave == 0 && row < nrows;
  double x[2 * NB];                               // wave 0: the panel row; wave 1 (lanes < 32): row of B21 in x[0..31]
  // S2 (optional, n x n, same layout): contributions to these 64 columns that were computed while earlier columns
  // were being factored (overlapped Schur tile batches); they are added when the columns become the panel.  The
  // appended rhs row (row n) has no S2 row.
  if (has_row) {
    const double* Arow = A + (size_t)row * n + k0;
#pragma unroll
    for (int c = 0; c < 2 * NB; ++c) x[c] = Arow[c];
    if (S2 && row < n) {
      const double* Srow = S2 + (size_t)row * n + k0;
#pragma unroll
      for (int c = 0; c < 2 * NB; ++c) x[c] += Srow[c];
    }
  } else if (wave == 1 && lane < NB) {
    const double* Arow = A + (size_t)(k0 + NB + lane) * n + k0;
#pragma unroll
    for (int c = 0; c < NB; ++c) x[c] = Arow[c];
    if (S2) {
      const double* Srow = S2 + (size_t)(k0 + NB + lane) * n + k0;
#pragma unroll
      for (int c = 0; c < NB; ++c) x[c] += Srow[c];
    }
#pragma unroll
    for (int c = NB; c < 2 * NB; ++c) x[c] = 0.0;
  } else {
#pragma unroll
    for (int c = 0; c < 2 * NB; ++c) x[c] = 0.0;
  }
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e / NB, c = e % NB;
    const size_t o1 = (size_t)(k0 + i) * n + k0 + c, o2 = (size_t)(k0 + NB + i) * n + k0 + NB + c;
    D1[i * LD + c] = (c <= i) ? A[o1] + (S2 ? S2[o1] : 0.0) : 0.0;
    D2[i * LD + c] = (c <= i) ? A[o2] + (S2 ? S2[o2] : 0.0) : 0.0;
  }
  FACTOR_DIAG(D1, rd1, (blockIdx.x == 0) ? fail : nullptr);
  // first block column: panel rows (wave 0) and the 32 rows of L21 (wave 1) through L11
  if (wave <= 1) {
    double (&x1)[NB] = reinterpret_cast<double (&)[NB]>(x);
    substitute_row<NB>(x1, D1, rd1);
    if (wave == 1 && lane < NB) {
#pragma unroll
      for (int c = 0; c < NB; ++c) L21[lane * LD + c] = x1[c];
    }
  }
  __syncthreads();
  // D2 -= L21 L21^T (lower triangle), then factor it
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e / NB, c = e % NB;
    if (c <= i) {
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int k = 0; k < NB; k += 2) { s0 += L21[i * LD + k] * L21[c * LD + k]; s1 += L21[i * LD + k + 1] * L21[c * LD + k + 1]; }
      D2[i * LD + c] -= s0 + s1;
    }
  }
  FACTOR_DIAG(D2, rd2, (blockIdx.x == 0) ? fail : nullptr);
  if (blockIdx.x == 0) {
    for (int e = tid; e < NB * NB; e += 256) {
      const int i = e / NB, c = e % NB;
      if (c <= i) {
        A[(size_t)(k0 + i) * n + k0 + c] = D1[i * LD + c];
        A[(size_t)(k0 + NB + i) * n + k0 + NB + c] = D2[i * LD + c];
      }
      A[(size_t)(k0 + NB + i) * n + k0 + c] = L21[i * LD + c];
    }
  }
  if (extra_wg) {
    // T1 = L11^-T, T2 = L22^-T for the backward solve (rows of the identity through the two substitutions)
    if (wave <= 1 && lane < NB) {
      double t[NB];
#pragma unroll
      for (int c = 0; c < NB; ++c) t[c] = (c == lane) ? 1.0 : 0.0;
      substitute_row<NB>(t, wave == 0 ? D1 : D2, wave == 0 ? rd1 : rd2);
      double* T = inv_blocks + (size_t)(k0 / NB + wave) * NB * NB + lane * NB;
#pragma unroll
      for (int c = 0; c < NB; ++c) T[c] = t[c];
    }
    return;
  }
  if (!has_row) return;
  // second block column of the panel rows: a2 -= x1 L21^T, then through L22
  double x2[NB];
#pragma unroll
  for (int c = 0; c < NB; ++c) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int k = 0; k < NB; k += 2) { s0 += x[k] * L21[c * LD + k]; s1 += x[k + 1] * L21[c * LD + k + 1]; }
    x2[c] = x[NB + c] - (s0 + s1);
  }
  substitute_row<NB>(x2, D2, rd2);
  double* Aout = A + (size_t)row * n + k0;
#pragma unroll
  for (int c = 0; c < NB; ++c) { Aout[c] = x[c]; Aout[NB + c] = x2[c]; }
}

// trailing update A[i][j] -= sum_k L[i][k0+k] L[j][k0+k] for i >= j >= k0+NB, tiles of 32x32, k < NB
template <int NB>
__global__ __launch_bounds__(256) void chol_update_kernel(double* __restrict__ A, int n, int nrows, int k0,
                                                          int num_tiles, const int32_t* skip) {
  constexpr int KS = NB / 4;
  if (skip && *skip) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // scalar: MFMAs
  const int t = blockIdx.x * 4 + wave;                                                       // behind scalar branches
  if (t >= num_tiles) return;
  int bi = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  while (bi * (bi + 1) / 2 > t) --bi;
  const int bj = t - bi * (bi + 1) / 2;
  const int base = k0 + NB;
  const int r0 = base + bi * 32, c0 = base + bj * 32;
  const int li = lane & 15, lk = lane >> 4;
  // operands: a[m][kk] = L[r0 + 16 m + li][k0 + 4 kk + lk], b[m][kk] = L[c0 + 16 m + li][k0 + 4 kk + lk]
  double a[2][KS], b[2][KS];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int ra = r0 + 16 * m + li, rb = c0 + 16 * m + li;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      a[m][kk] = (ra < nrows) ? A[(size_t)ra * n + k0 + 4 * kk + lk] : 0.0;
      b[m][kk] = (rb < nrows) ? A[(size_t)rb * n + k0 + 4 * kk + lk] : 0.0;
    }
  }
  // the tile being updated is loaded up front, together with the operands (f64 C/D layout: col = lane & 15,
  // row = (lane >> 4) + 4 * reg): one memory round trip instead of two on a latency-bound launch
  f64x4 acc[2][2], old[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      acc[m][q] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int i = r0 + 16 * m + lk + 4 * reg, j = c0 + 16 * q + li;
        old[m][q][reg] = (i < nrows && j < n && j <= i) ? A[(size_t)i * n + j] : 0.0;
      }
    }
#pragma unroll
  for (int kk = 0; kk < KS; ++kk)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        acc[m][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][kk], b[q][kk], acc[m][q], 0, 0, 0);
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int i = r0 + 16 * m + lk + 4 * reg, j = c0 + 16 * q + li;
        if (i < nrows && j < n && j <= i) A[(size_t)i * n + j] = old[m][q][reg] - acc[m][q][reg];
      }
}

// Substitutions by one workgroup: the slow, size-unlimited path.  FORWARD (L z = b) is only needed when b
// is not stored as row n of the factored matrix; BACKWARD only when the system is too large for the LDS kernel.
// Each diagonal block is staged in LDS so that the dependent steps never wait on L2/HBM.
template <int NB, bool FORWARD>
__device__ __forceinline__ void tri_solve_block(const double* __restrict__ L, double* __restrict__ b, int n, int k0,
                                                double* Dl, double* rd, double* z) {
  constexpr int LD = NB + 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int nb = min(NB, n - k0);
  for (int e = tid; e < NB * NB; e += 256) {
    const int i = e / NB, c = e % NB;
    if (i < nb && c <= i) {
      const double v = L[(size_t)(k0 + i) * n + k0 + c];
      Dl[i * LD + c] = v;
      if (i == c) rd[i] = fast_rcp(v);
    }
  }
  __syncthreads();
  if (tid < 64) {
    double v = (lane < nb) ? b[k0 + lane] : 0.0;
    if (FORWARD) {
      for (int k = 0; k < nb; ++k) {
        const double zk = __shfl(v, k, 64) * rd[k];
        if (lane == k) v = zk;
        if (lane > k && lane < nb) v -= Dl[lane * LD + k] * zk;
      }
    } else {
      for (int k = nb - 1; k >= 0; --k) {
        const double yk = __shfl(v, k, 64) * rd[k];
        if (lane == k) v = yk;
        if (lane < k) v -= Dl[k * LD + lane] * yk;
      }
    }
    if (lane < nb) { z[lane] = v; b[k0 + lane] = v; }
  }
  __syncthreads();
}

template <int NB>
__global__ __launch_bounds__(256) void chol_solve_kernel(const double* __restrict__ L, double* __restrict__ b, int n,
                                                         int do_forward, int do_backward, const int32_t* skip) {
  __shared__ double Dl[NB * (NB + 1)];
  __shared__ double rd[NB];
  __shared__ double z[NB];
  if (skip && *skip) return;
  const int tid = threadIdx.x;
  const int nblk = (n + NB - 1) / NB;
  if (do_forward) {
    for (int blk = 0; blk < nblk; ++blk) {
      const int k0 = blk * NB, nb = min(NB, n - k0);
      tri_solve_block<NB, true>(L, b, n, k0, Dl, rd, z);
      for (int i = k0 + nb + tid; i < n; i += 256) {
        const double* Li = L + (size_t)i * n + k0;
        double s = 0.0;
        for (int k = 0; k < nb; ++k) s += Li[k] * z[k];
        b[i] -= s;
      }
      __syncthreads();
    }
  }
  if (!do_backward) return;
  for (int blk = nblk - 1; blk >= 0; --blk) {
    const int k0 = blk * NB, nb = min(NB, n - k0);
    tri_solve_block<NB, false>(L, b, n, k0, Dl, rd, z);
    for (int i = tid; i < k0; i += 256) {
      double s0 = b[i];
      for (int k = 0; k < nb; ++k) s0 -= L[(size_t)(k0 + k) * n + i] * z[k];
      b[i] = s0;
    }
    __syncthreads();
  }
}

// Backward substitution L^T x = y with the inverted diagonal blocks: one workgroup of 1024 threads, y in LDS.
// Per block row (last to first): x_j = T_j y_j (T_j = L_jj^-T, NB x NB mat-vec by NB lanes), then
// y_i -= sum_k L[k0+k][i] x_k for all i < k0, one column per thread, 32 rows at a time.  The first 32 loads of
// a column and the next T block are issued BEFORE the mat-vec, so their latency overlaps it.
constexpr int kBackThreads = 1024;
template <int NB>
__global__ __launch_bounds__(kBackThreads) void chol_backward_kernel(const double* __restrict__ L, double* __restrict__ b,
                                                                     int n, const double* __restrict__ inv_blocks,
                                                                     const int32_t* skip) {
  constexpr int LD = NB + 1, TPT = NB * NB / kBackThreads;     // T elements per thread (1 or 4)
  extern __shared__ double sh[];
  if (skip && *skip) return;
  const int tid = threadIdx.x;
  const int nblk = (n + NB - 1) / NB;
  double* y = sh;                                   // [nblk * NB]
  double* T = y + nblk * NB;                        // [NB][NB + 1]
  double* xj = T + NB * LD;                         // [NB]
  for (int i = tid; i < nblk * NB; i += kBackThreads) y[i] = (i < n) ? b[i] : 0.0;
#pragma unroll
  for (int q = 0; q < TPT; ++q) {
    const int e = tid + q * kBackThreads;
    T[(e / NB) * LD + (e % NB)] = inv_blocks[(size_t)(nblk - 1) * NB * NB + e];
  }
  __syncthreads();
  for (int blk = nblk - 1; blk >= 0; --blk) {
    const int k0 = blk * NB, nb = min(NB, n - k0);
    // issue the loads that do not depend on x_j
    double tnext[TPT];
#pragma unroll
    for (int q = 0; q < TPT; ++q)
      tnext[q] = (blk > 0) ? inv_blocks[(size_t)(blk - 1) * NB * NB + tid + q * kBackThreads] : 0.0;
    double l[32];
    const bool have = (tid < k0) && nb == NB;
#pragma unroll
    for (int k = 0; k < 32; ++k) l[k] = have ? L[(size_t)(k0 + k) * n + tid] : 0.0;
    if (tid < NB) {
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int c = 0; c < NB; c += 4) {
        s0 += T[tid * LD + c] * y[k0 + c];
        s1 += T[tid * LD + c + 1] * y[k0 + c + 1];
        s2 += T[tid * LD + c + 2] * y[k0 + c + 2];
        s3 += T[tid * LD + c + 3] * y[k0 + c + 3];
      }
      xj[tid] = (tid < nb) ? (s0 + s1) + (s2 + s3) : 0.0;
    }
    __syncthreads();
    if (tid < NB) y[k0 + tid] = xj[tid];
    if (nb == NB) {
      for (int i = tid; i < k0; i += kBackThreads) {
        double acc = 0.0;
#pragma unroll
        for (int h = 0; h < NB; h += 32) {
          if (h > 0 || i != tid) {                    // (the first 32 rows of the first column are already here)
#pragma unroll
            for (int k = 0; k < 32; ++k) l[k] = L[(size_t)(k0 + h + k) * n + i];
          }
          double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
          for (int k = 0; k < 32; k += 4) {
            s0 += l[k] * xj[h + k]; s1 += l[k + 1] * xj[h + k + 1]; s2 += l[k + 2] * xj[h + k + 2]; s3 += l[k + 3] * xj[h + k + 3];
          }
          acc += (s0 + s1) + (s2 + s3);
        }
        y[i] -= acc;
      }
    } else {                                        // ragged last block (processed first)
      for (int i = tid; i < k0; i += kBackThreads) {
        double s0 = 0.0;
        for (int k = 0; k < nb; ++k) s0 += L[(size_t)(k0 + k) * n + i] * xj[k];
        y[i] -= s0;
      }
    }
    __syncthreads();                                // all reads of T / xj are done
#pragma unroll
    for (int q = 0; q < TPT; ++q) {
      const int e = tid + q * kBackThreads;
      T[(e / NB) * LD + (e % NB)] = tnext[q];
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += kBackThreads) b[i] = y[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// Dataflow factorisation: ONE launch, one workgroup per 64 x 64 tile of the lower triangle (+ one tile row for the
// appended rhs), ordered by (block column, block row).  A tile's workgroup loads its block of A into MFMA accumulators,
// applies the updates of the block columns to its left AS THEIR FACTOR TILES BECOME AVAILABLE (left-looking, per-tile
// flags), then finishes: a diagonal tile factors itself (factor_diag_lds4 with the inverse T = L^-T as a by-product),
// an off-diagonal tile multiplies by T of its column.  Nothing but the true dependences orders the work: the trailing
// update of step k overlaps the pivot chain of steps k+1.., decoupled leading blocks (camera split, band structure:
// first_blk) factor concurrently without any special casing, and ~40 dependent launches become one.
// Progress: a workgroup waits only for tiles EARLIER in the launch order; workgroups are dispatched in order per XCD
// (MI355X_MICROARCH.md, "Workgroup dispatch"), so the earliest unfinished tile is always resident and never waits on
// an undispatched one.  Every spin is bounded all the same (fail flag 2 instead of a hang).
// Hand-offs follow the {8-byte agent-scope atomics on both sides} form of the guide: published tiles are written
// with relaxed agent-scope stores (write-through), drained (s_waitcnt vmcnt(0)), then the flag is stored; readers poll
// the flag with relaxed agent-scope loads and read the payload with agent-scope loads (L1 bypassed) -- no fences.
constexpr int DFB = 64;                       // tile edge
constexpr int kSpinLimit = 1 << 22;

__device__ __forceinline__ void st_agent(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Round 6, backward substitution: x travels between workgroups WITHOUT a flag.  The 64 values of a block are published into a
// buffer of their own (`xpub`, behind the launch-order map) that the forward launch has filled with a sentinel -- a NaN pattern no
// solve produces -- and every consumer lane polls ITS element until it is no longer the sentinel: one memory round trip per
// hand-off instead of two (flag observed, then the values fetched), and no store drain + flag on the producer's side.
#ifndef VGG_BW_SENTINEL
#define VGG_BW_SENTINEL 1
#endif
constexpr unsigned long long kXSentinel = 0x7FF8C0DEC0DE0001ull;
__host__ __device__ inline double* df_xpub(int32_t* flags, int nbk) {
  const size_t ints = (size_t)(nbk + 1) * nbk + 3 * (size_t)nbk + 1 + ((size_t)nbk * (nbk + 1) / 2 + nbk);   // flags | map
  return reinterpret_cast<double*>(flags + ((ints + 1) & ~(size_t)1));                                        // (8-byte aligned: flags is)
}
// all threads: wait until *flag >= want (raised by another workgroup of this launch; flags only grow)
__device__ __forceinline__ void df_wait_ge(const int32_t* flag, int32_t want, int32_t* fail) {
  if (threadIdx.x == 0) {
    int spins = 0;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      __builtin_amdgcn_s_sleep(4);
      ++spins;
      // (a launch that has already failed is not waited out flag by flag)
      const bool lost = fail && (spins & 1023) == 0 && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2;
      if (spins > kSpinLimit || lost) { if (fail) __hip_atomic_store(fail, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
  }
  __syncthreads();
}
__device__ __forceinline__ void df_wait(const int32_t* flag, int32_t* fail) { df_wait_ge(flag, 1, fail); }
// workgroup barrier for data that went through LDS only: __syncthreads() is a workgroup-scope release, which on gfx9 waits
// for every outstanding global STORE of the wavefront as well (s_waitcnt vmcnt(0)) -- ~0.7 us when agent-scope stores of a
// tile are in flight, and the flag that publishes them has its own drain anyway
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// all threads: every store of this workgroup has left the CU, then raise the flag
__device__ __forceinline__ void df_publish(int32_t* flag, int32_t value = 1) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifdef VGG_CHOL_TRACE
__device__ unsigned long long* g_chol_trace = nullptr;      // [tiles][8] wall-clock stamps (100 MHz), scripts/ubench/chol_bench
#define DF_STAMP_AT(tile, slot) do { if (g_chol_trace && threadIdx.x == 0) g_chol_trace[(size_t)(tile) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define DF_STAMP_AT(tile, slot) do { } while (0)
#endif
#define DF_STAMP(slot) DF_STAMP_AT(nat_tile, slot)

// 32 x 32 x 32 product on the matrix cores by the four wavefronts of a workgroup, operands read from LDS through
// accessors: out(i, j) <- sum_k opA(i, k) * opB(j, k); wavefront w owns the 16 x 16 tile (w >> 1, w & 1).
template <class FA, class FB, class FO>
__device__ __forceinline__ void mm32_lds(FA opA, FB opB, FO out) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ti = wave >> 1, tj = wave & 1;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < 8; ++s)
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(opA(16 * ti + li, 4 * s + lk), opB(16 * tj + li, 4 * s + lk), acc, 0, 0, 0);
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) out(16 * ti + lk + 4 * reg, 16 * tj + li, acc[reg]);
}

// 64 x 64 diagonal block (LDS, ld = 65, lower triangle valid) -> its factor in place and T = L^-T (upper triangular) in
// Tl, as two 32 x 32 factorisations (factor_diag_lds4) glued by four small matrix-core products:
//   L21 = D21 T11,  D22 -= L21 L21^T,  T12 = -T11 (L21^T T22).
// (One 64-wide factor_diag_lds4 was measured at 40 us: its rank-4 trailing updates are LDS-issue bound.)
#ifdef VGG_CHOL_LOOKAHEAD                          // opt-in experiment (measured slower, see factor_diag_lds4_la)
#define FACTOR32(Dp, Tp, rdp, scrp, failp) factor_diag_lds4_la<H, true, LD>(Dp, Tp, rdp, scrp, failp)
#else
#define FACTOR32(Dp, Tp, rdp, scrp, failp) factor_diag_lds4<H, true, LD>(Dp, Tp, rdp, scrp, failp)
#endif
__device__ __forceinline__ void factor64(double* D, double* Tl, double* rd, double* scr, int32_t* fail) {
  constexpr int LD = DFB + 1, H = 32;
  const int tid = threadIdx.x;
  FACTOR32(D, Tl, rd, scr, fail);
  double v[4];
  int vi[4], vj[4], cnt = 0;
  mm32_lds([&](int i, int k) { return D[(H + i) * LD + k]; }, [&](int j, int k) { return Tl[k * LD + j]; },
           [&](int i, int j, double x) { v[cnt] = x; vi[cnt] = i; vj[cnt] = j; ++cnt; });
  __syncthreads();                                       // every wavefront has read D21 before anyone overwrites it
#pragma unroll
  for (int q = 0; q < 4; ++q) D[(H + vi[q]) * LD + vj[q]] = v[q];
  for (int e = tid; e < H * H; e += 256) Tl[(H + e / H) * LD + e % H] = 0.0;           // T21 = 0
  __syncthreads();
  mm32_lds([&](int i, int k) { return D[(H + i) * LD + k]; }, [&](int j, int k) { return D[(H + j) * LD + k]; },
           [&](int i, int j, double x) { if (j <= i) D[(H + i) * LD + H + j] -= x; });
  FACTOR32(D + H * LD + H, Tl + H * LD + H, rd + H, scr, fail);
  double* W = scr;                                       // 32 x 32
  mm32_lds([&](int i, int k) { return D[(H + k) * LD + i]; }, [&](int j, int k) { return Tl[(H + k) * LD + H + j]; },
           [&](int i, int j, double x) { W[i * H + j] = x; });
  __syncthreads();
  mm32_lds([&](int i, int k) { return Tl[i * LD + k]; }, [&](int j, int k) { return W[k * H + j]; },
           [&](int i, int j, double x) { Tl[i * LD + H + j] = -x; });
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------------------------
// Round 3: the 64 x 64 diagonal block as FOUR 16 x 16 blocks (blocked right-looking), instead of two 32 x 32 ones.
//   * A 16 x 16 diagonal block is factored by ONE wavefront with the block in a matrix-core accumulator (factor16_mfma
//     below), its inverse transpose coming out of the same pass.
//   * Everything else is 16 x 16 x 16 products on the matrix cores (mm16): the panel (block column kb times T_kk), the
//     trailing update, and the same column operations on the rows of the identity, which leave T = L^-T in Tl.
//   * Only the panel block under the diagonal + the 16 x 16 factorisation sit on the critical path (wavefront 0, the next
//     diagonal block carried in its registers); the other trailing products of step kb - 1 are done by wavefronts 1..3
//     meanwhile; two barriers per 16 columns.
// (The first form of the 16 x 16 routine -- lane = (row, quarter of the columns), panel and multipliers exchanged through
//  LDS -- and its compile-time ablations are in the history: commit 9f5c616)
template <class FA, class FB, class FO>
__device__ __forceinline__ void mm16(FA opA, FB opB, FO out) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  f64x4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(opA(li, 4 * s + lk), opB(li, 4 * s + lk), acc, 0, 0, 0);
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) out(lk + 4 * reg, li, acc[reg]);
}

// rsqrt / sqrt of a positive double from the hardware estimate + two Newton steps and one correction of the root (the
// library sqrt and the division behind it are ~70 dependent operations, four times per 64 x 64 block)
__device__ __forceinline__ void fast_rsqrt_sqrt(double x, double& rs, double& sd) {
  double r = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  r = r * __builtin_fma(-h * r, r, 1.5);
  r = r * __builtin_fma(-h * r, r, 1.5);
  double s = x * r;
  s = __builtin_fma(0.5 * r, __builtin_fma(-s, s, x), s);
  rs = r; sd = s;
}

// The 16 x 16 block at (16 kb, 16 kb): L_kk into D, T_kk = L_kk^-T (upper triangular) into the same block of Tl, by ONE
// wavefront with the block in ONE matrix-core accumulator and no LDS traffic inside the loop.
//   * The (symmetric, fully kept) block sits in the C layout of v_mfma_f64_16x16x4: lane (li, lk) holds the entries
//     (row lk + 4 reg, column li) -- by symmetry also row li, columns lk + 4 reg.  So the four panel columns 4 qq .. 4 qq + 3
//     of ALL rows are register acc[qq], and they are laid out exactly as a matrix-core operand (row li, k = lk).
//   * A step of four pivots is then four matrix instructions on registers the lanes already hold:
//       X^T = W^T A_panel^T      the elimination inside the panel (W: 4 x 4 unit upper triangular, from the multipliers)
//       A   -= X (X R)^T         the rank-4 update (R = diag 1/d_t), rows / columns of finished pivots masked to zero
//       Y^T = W^T E_panel^T,  E^T -= (X R) Y^T     the same column operations on the rows of the identity (T = L^-T)
//     The only cross-lane traffic is the 4 x 4 pivot block, broadcast with v_readlane (ten doubles); its elimination
//     (reciprocals, multipliers) is computed redundantly by every lane as before.
//   * The column scales 1 / sqrt(d) are taken once at the end (each lane needs the four of its own columns).
// Measured (scripts/ubench/factor64_bench): see DESIGN.md section 6.
// the block (kb, kb) of D (lower triangle valid) in that layout, filled symmetrically
template <int LD>
__device__ __forceinline__ f64x4 load_block16_sym(const double* D, int kb) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  const double* Db = D + (16 * kb) * LD + 16 * kb;
  f64x4 acc;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = lk + 4 * reg, hi = max(row, li), lo = min(row, li);
    acc[reg] = Db[hi * LD + lo];
  }
  return acc;
}

// Round 6, three formulations of the step MEASURED AND NOT ADOPTED (compile-time switches, defaults = the round-5 step;
// scripts/ubench/factor64_bench, profiles/r06_ab_factor16_variants.jsonl -- one 64 x 64 block, same box):
//   VGG_F16_LA = 1 / 2 / 4 -- look-ahead: the first LA rows of the NEXT 4 x 4 pivot block are brought up to date by every lane
//     redundantly, in scalar arithmetic, from values broadcast at the start of the step, so that the next reciprocal chain
//     starts one multiply-add behind r3 instead of behind two matrix instructions and ten broadcasts: 9.63 -> 10.3 / 11.2 /
//     13.2 us.  What it takes off the dependent chain (~100 cycles for LA = 1: the first reciprocal only -- the other nine
//     values still arrive through the update) it puts back as ~30 more instructions per step in a wavefront that issues in
//     order.
//   VGG_F16_SQ = 1 -- d_{t+1} takes its last term as (u^2) r_t, one operation behind r_t instead of two: inside the noise.
//   VGG_F16_BCAST = 1 / 2 -- the pivot block broadcast through the LDS crossbar (ds_bpermute_b32, result in a VECTOR register:
//     an fp64 instruction takes one scalar operand, and a step spends 14 v_mov_b32 on carrying v_readlane results back):
//     123 -> 95 instructions per step and 9.53 -> 10.1 / 10.3 us -- the crossbar's round trip in front of every pivot chain
//     costs more than the moves.
//   -mllvm -amdgpu-mfma-vgpr-form (no AGPR copies around the matrix instructions, 123 -> 105 instructions): 9.68 -> 9.53 us.
// Reading: a step is ~850 cycles of dependent fp64 latency (four reciprocal chains, two matrix instructions, the broadcast)
// plus the issue slots of whatever else is in the stream; removing cheap instructions buys little, and nothing found shortens
// the dependent part without adding more than it removes.  The 64 x 64 block stays at 9.6 us, the block column at 14 us.
#ifndef VGG_F16_LA
#define VGG_F16_LA 0
#endif
#ifndef VGG_F16_SQ
#define VGG_F16_SQ 0
#endif
#ifndef VGG_F16_BCAST
#define VGG_F16_BCAST 0
#endif
template <int LD, int LA = VGG_F16_LA, bool SQ = (VGG_F16_SQ != 0)>
__device__ __forceinline__ bool factor16_mfma(f64x4 acc, double* D, double* Tl, int kb) {
  static_assert(LA >= 0 && LA <= 4, "rows of the next pivot block taken ahead");
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  double* Db = D + (16 * kb) * LD + 16 * kb;
  double* Tb = Tl + (16 * kb) * LD + 16 * kb;
  f64x4 accE;
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) accE[reg] = (li == lk + 4 * reg) ? 1.0 : 0.0;
  double xfin[4], yfin[4], rs[4], sd[4];
  double dlast = 1.0;
  const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
  // Round 4: the step is bound by its INSTRUCTION COUNT (one wavefront, ~135 instructions per four pivots), so the lane
  // selections are multiply-adds with per-lane 0 / 1 constants instead of nested selects (exact: one term per lane is
  // non-zero), and a non-positive or non-finite pivot is no longer caught pivot by pivot (eight compares, eight selects
  // per step) but where it ends up anyway: the scale sqrt(d) of the lane's own column is NaN (rsq of d <= 0, 0 * inf,
  // or a NaN handed down from an earlier pivot), tested once per 16 columns.
  const double k01 = (li == 1 && lk == 0) ? 1.0 : 0.0, k02 = (li == 2 && lk == 0) ? 1.0 : 0.0, k03 = (li == 3 && lk == 0) ? 1.0 : 0.0;
  const double k12 = (li == 2 && lk == 1) ? 1.0 : 0.0, k13 = (li == 3 && lk == 1) ? 1.0 : 0.0, k23 = (li == 3 && lk == 2) ? 1.0 : 0.0;
  const double kone = (li < 4 && lk == li) ? 1.0 : 0.0;
  const double q0 = (lk == 0) ? 1.0 : 0.0, q1 = (lk == 1) ? 1.0 : 0.0, q2 = (lk == 2) ? 1.0 : 0.0, q3 = (lk == 3) ? 1.0 : 0.0;
  double pl[4][4] = {};                              // rows < LA of the next pivot block, taken ahead (lower triangle)
  int vzero = 0;
  asm volatile("" : "+v"(vzero));                    // (a vector register the compiler cannot fold: bpermute_f64)
  (void)vzero;
#pragma unroll
  for (int j0 = 0; j0 < 16; j0 += 4) {
    const int qq = j0 / 4;
    const double pa = acc[qq], pe = accE[qq];
    // p_ab = A[j0 + a][j0 + b] is register acc[qq] of lane (li = j0 + a, lk = b) -- or was taken ahead by the previous step
#if VGG_F16_BCAST == 0
#define VGG_BC(v, l) readlane_f64(v, l)
#elif VGG_F16_BCAST == 1
#define VGG_BC(v, l) bpermute_f64(v, l, vzero)
#else                                                // 2: the first pivot by v_readlane (its reciprocal starts the chain), the rest through LDS
#define VGG_BC(v, l) (((l) == j0) ? readlane_f64(v, l) : bpermute_f64(v, l, vzero))
#endif
#define VGG_P(a, b) ((j0 > 0 && (a) < LA) ? pl[a][b] : VGG_BC(pa, j0 + (a) + 16 * (b)))
    const double d0 = VGG_P(0, 0);
    const double r0 = fast_rcp<1>(d0);
    const double u10 = VGG_P(1, 0), p11 = VGG_P(1, 1);
    const double u20 = VGG_P(2, 0), p21 = VGG_P(2, 1), p22 = VGG_P(2, 2);
    const double u30 = VGG_P(3, 0), p31 = VGG_P(3, 1), p32 = VGG_P(3, 2), p33 = VGG_P(3, 3);
#undef VGG_P
#undef VGG_BC
    // inputs of the look-ahead: row j0 + 4 + a of this panel (c) and of the next pivot block (e), as they stand
    constexpr int LAR = LA > 0 ? LA : 1;
    double cx[LAR][4], ce[LAR][LAR];
    const bool ahead = LA > 0 && j0 < 12;
    if (ahead) {
      const double pn = acc[(qq + 1) & 3];
#pragma unroll
      for (int a = 0; a < LA; ++a) {
#pragma unroll
        for (int b = 0; b < 4; ++b) cx[a][b] = readlane_f64(pa, j0 + 4 + a + 16 * b);
#pragma unroll
        for (int b = 0; b <= a; ++b) ce[a][b] = readlane_f64(pn, j0 + 4 + a + 16 * b);
      }
    }
    // the scale of this lane's column of the PREVIOUS step: one short chain per lane in the shadow of the matrix instructions
    if (j0 > 0) fast_rsqrt_sqrt(dlast, rs[qq - 1], sd[qq - 1]);
    const double m10 = u10 * r0, m20 = u20 * r0, m30 = u30 * r0;        // multipliers a_cj / d_j
    const double d1 = SQ ? __builtin_fma(-(u10 * u10), r0, p11) : p11 - u10 * m10;
    const double u21 = p21 - u20 * m10, u31 = p31 - u30 * m10;
    const double r1 = fast_rcp<1>(d1);
    const double m21 = u21 * r1, m31 = u31 * r1;
    const double d2 = SQ ? __builtin_fma(-(u21 * u21), r1, p22 - u20 * m20) : (p22 - u20 * m20) - u21 * m21;
    const double u32 = (p32 - u30 * m20) - u31 * m21;
    const double r2 = fast_rcp<1>(d2);
    const double m32 = u32 * r2;
    const double d3 = SQ ? __builtin_fma(-(u32 * u32), r2, (p33 - u30 * m30) - u31 * m31) : ((p33 - u30 * m30) - u31 * m31) - u32 * m32;
    const double r3 = fast_rcp<1>(d3);
    if (ahead) {
      double xs[LAR][4];
#pragma unroll
      for (int a = 0; a < LA; ++a) {
        xs[a][0] = cx[a][0];
        xs[a][1] = cx[a][1] - xs[a][0] * m10;
        xs[a][2] = (cx[a][2] - xs[a][0] * m20) - xs[a][1] * m21;
        xs[a][3] = ((cx[a][3] - xs[a][0] * m30) - xs[a][1] * m31) - xs[a][2] * m32;
      }
#pragma unroll
      for (int a = 0; a < LA; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
          double v = __builtin_fma(-(xs[a][0] * xs[b][0]), r0, ce[a][b]);
          v = __builtin_fma(-(xs[a][1] * xs[b][1]), r1, v);
          v = __builtin_fma(-(xs[a][2] * xs[b][2]), r2, v);
          pl[a][b] = __builtin_fma(-(xs[a][3] * xs[b][3]), r3, v);
        }
    }
    // x_t = sum_k a_k W[k][t]:  W[k][t] = -sum_{k <= s < t} W[k][s] m_ts,  W[k][k] = 1
    const double w02 = __builtin_fma(m10, m21, -m20), w13 = __builtin_fma(m21, m32, -m31);
    const double w03 = __builtin_fma(-w02, m32, __builtin_fma(m10, m31, -m30));
    // operand lane (li = n, lk = k) carries W[k][n] (n < 4, k <= n), zero elsewhere  (w01 = -m10, w12 = -m21, w23 = -m32)
    double wsel = __builtin_fma(-k01, m10, kone);
    wsel = __builtin_fma(-k12, m21, wsel);
    wsel = __builtin_fma(k02, w02, wsel);
    wsel = __builtin_fma(-k23, m32, wsel);
    wsel = __builtin_fma(k13, w13, wsel);
    wsel = __builtin_fma(k03, w03, wsel);
    const f64x4 xt = __builtin_amdgcn_mfma_f64_16x16x4f64(wsel, pa, zero4, 0, 0, 0);
    const f64x4 yt = __builtin_amdgcn_mfma_f64_16x16x4f64(wsel, pe, zero4, 0, 0, 0);
    const double x = xt[0], y = yt[0];               // X[li][lk], Y[li][lk]
    const double rsel = __builtin_fma(q3, r3, __builtin_fma(q2, r2, __builtin_fma(q1, r1, q0 * r0)));
    dlast = __builtin_fma(q3, d3, __builtin_fma(q2, d2, __builtin_fma(q1, d1, q0 * d0)));
    xfin[qq] = x; yfin[qq] = y;
    if (j0 < 12) {
      const double xm = (li >= j0 + 4) ? x : 0.0;
      const double v = xm * rsel;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xm, v, acc, 0, 0, 0);
      accE = __builtin_amdgcn_mfma_f64_16x16x4f64(-v, y, accE, 0, 0, 0);
    }
  }
  fast_rsqrt_sqrt(dlast, rs[3], sd[3]);
  bool badlane = false;
  constexpr double HUGE_ = 1.7976931348623157e308;
  // (the upper triangle of the block in D is never read: it takes whatever the product left there)
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const int c = 4 * qq + lk;
    badlane = badlane || !(sd[qq] > 0.0) || !(sd[qq] < HUGE_);
    Db[li * LD + c] = (li == c) ? sd[qq] : xfin[qq] * rs[qq];
    Tb[li * LD + c] = (li <= c) ? yfin[qq] * rs[qq] : 0.0;
  }
  return __any(badlane);
}

// slab(kb): called by wavefront 1 alone (all 64 lanes) once the 16 columns 16 kb .. 16 kb + 15 of T are final (kb = 0, 1, 2;
// the last 16 columns are final on return): the caller hands them on while the factorisation goes on.
#ifndef VGG_F64_ABL
#define VGG_F64_ABL 0      // scripts/ubench/factor64_bench ablations: 1 no 16 x 16 factorisation, 2 no trailing updates, 4 no
#endif                     // side panels, 8 no critical panel
template <class FS>
__device__ __forceinline__ void factor64_blocked(double* D, double* Tl, double* scr, int32_t* fail, FS slab) {
  constexpr int LD = DFB + 1, B = 16;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (Round 4: no zero fill of Tl and no barrier here -- 0.6 us on the pivot chain per 64 columns.  A block of T above the
  //  diagonal is STORED by its first update (step k = its block row, where the identity's block is still zero); the blocks
  //  below the diagonal are never written and never read from LDS: whoever hands T on writes zeros for them.  The caller's
  //  barrier after filling D is the one the first loads below need.)
  bool bad = false;
  // D[ib][jb] -= L[ib][k] L[jb][k]^T   /   E[ib][jb] -= E[ib][k] L[jb][k]^T   (16 x 16 blocks; E = the rows of the identity in Tl)
  auto trail_A = [&](int ib, int jb, int k) __attribute__((always_inline)) {
    mm16([&](int i, int kk) { return D[(B * ib + i) * LD + B * k + kk]; }, [&](int j, int kk) { return D[(B * jb + j) * LD + B * k + kk]; },
         [&](int i, int j, double x) { D[(B * ib + i) * LD + B * jb + j] -= x; });
  };
  auto trail_E = [&](int ib, int jb, int k) __attribute__((always_inline)) {
    if (k == ib)
      mm16([&](int i, int kk) { return Tl[(B * ib + i) * LD + B * k + kk]; }, [&](int j, int kk) { return D[(B * jb + j) * LD + B * k + kk]; },
           [&](int i, int j, double x) { Tl[(B * ib + i) * LD + B * jb + j] = -x; });
    else
      mm16([&](int i, int kk) { return Tl[(B * ib + i) * LD + B * k + kk]; }, [&](int j, int kk) { return D[(B * jb + j) * LD + B * k + kk]; },
           [&](int i, int j, double x) { Tl[(B * ib + i) * LD + B * jb + j] -= x; });
  };
  f64x4 dacc = {0.0, 0.0, 0.0, 0.0};               // wavefront 0: the diagonal block it is about to factor
  if (wave == 0) dacc = load_block16_sym<LD>(D, 0);
#pragma unroll 1
  for (int kb = 0; kb < 4; ++kb) {
    if (wave == 0) {
      if (!(VGG_F64_ABL & 1)) bad = factor16_mfma<LD>(dacc, D, Tl, kb) || bad;
    } else if (kb > 0 && !(VGG_F64_ABL & 2)) {
      // the rest of the trailing update of step k = kb - 1, dealt to wavefronts 2, 3, 1, 2, 3, 1 ...; wavefront 1 first hands
      // on the columns of T that step k completed (they are not touched again)
      const int k = kb - 1;
      if (wave == 1) slab(k);
      int t = 1;
      for (int jb = k + 1; jb < 4; ++jb)
        for (int ib = jb; ib < 4; ++ib) {
          if (ib == kb && jb == kb) continue;              // wavefront 0 did it with the panel (critical path)
          if (1 + (t % 3) == wave) trail_A(ib, jb, k);
          ++t;
        }
      for (int ib = 0; ib <= k; ++ib)
        for (int jb = k + 1; jb < 4; ++jb) {
          if (1 + (t % 3) == wave) trail_E(ib, jb, k);
          ++t;
        }
    }
    // (kb == 0: every global store this wavefront issued before the factorisation has left the CU -- free here, 2 us after
    //  the last of them; the caller's `slab(0)` may then publish what they wrote: VGG_DF_LATE_XFLAG)
    if (kb == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                 // L_kk, T_kk and every update of step kb - 1 are in LDS
    // panels: L[ib][kb] = D[ib][kb] T_kk (ib > kb), E[ib][kb] = E[ib][kb] T_kk (ib < kb): three products.
    // Wavefront 0 has the block under the diagonal, the one the next diagonal block waits for: it forms the product
    // TRANSPOSED, so that its accumulator registers are the matrix-core operands of X X^T as they stand (k taken in the
    // order lk + 4 reg on both sides), and takes the next diagonal block (every earlier update is in LDS by now) minus
    // X X^T straight into the registers it factors from -- no pass through LDS, no barrier on that path.
    if (wave == 0 && kb < 3) {
      if (VGG_F64_ABL & 8) { __syncthreads(); continue; }
      const int lane = tid & 63, li = lane & 15, lk = lane >> 4, ib = kb + 1;
      dacc = load_block16_sym<LD>(D, ib);
      f64x4 xa = {0.0, 0.0, 0.0, 0.0}, xb = {0.0, 0.0, 0.0, 0.0};
      xa = __builtin_amdgcn_mfma_f64_16x16x4f64(Tl[(B * kb + lk) * LD + B * kb + li], D[(B * ib + li) * LD + B * kb + lk], xa, 0, 0, 0);
      xb = __builtin_amdgcn_mfma_f64_16x16x4f64(Tl[(B * kb + 4 + lk) * LD + B * kb + li], D[(B * ib + li) * LD + B * kb + 4 + lk], xb, 0, 0, 0);
      xa = __builtin_amdgcn_mfma_f64_16x16x4f64(Tl[(B * kb + 8 + lk) * LD + B * kb + li], D[(B * ib + li) * LD + B * kb + 8 + lk], xa, 0, 0, 0);
      xb = __builtin_amdgcn_mfma_f64_16x16x4f64(Tl[(B * kb + 12 + lk) * LD + B * kb + li], D[(B * ib + li) * LD + B * kb + 12 + lk], xb, 0, 0, 0);
      const f64x4 xt = xa + xb;
      f64x4 xx = {0.0, 0.0, 0.0, 0.0}, xy = {0.0, 0.0, 0.0, 0.0};
      xx = __builtin_amdgcn_mfma_f64_16x16x4f64(xt[0], xt[0], xx, 0, 0, 0);
      xy = __builtin_amdgcn_mfma_f64_16x16x4f64(xt[1], xt[1], xy, 0, 0, 0);
      xx = __builtin_amdgcn_mfma_f64_16x16x4f64(xt[2], xt[2], xx, 0, 0, 0);
      xy = __builtin_amdgcn_mfma_f64_16x16x4f64(xt[3], xt[3], xy, 0, 0, 0);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) D[(B * ib + li) * LD + B * kb + lk + 4 * reg] = xt[reg];
      dacc = dacc - (xx + xy);
    } else if (wave < 3 && !(VGG_F64_ABL & 4)) {
      const int ib = (wave < 3 - kb) ? kb + 1 + wave : wave - (3 - kb);      // kb+1 .. 3, then 0 .. kb-1
      double* X = (wave < 3 - kb) ? D : Tl;
      mm16([&](int i, int kk) { return X[(B * ib + i) * LD + B * kb + kk]; }, [&](int j, int kk) { return Tl[(B * kb + kk) * LD + B * kb + j]; },
           [&](int i, int j, double x) { X[(B * ib + i) * LD + B * kb + j] = x; });
    }
    __syncthreads();
  }
  if (bad && (tid & 63) == 0) *fail = 1;
  __syncthreads();
}

constexpr int kDfOrderMax = 64;               // block columns up to which the chain and the update order below are used
// Round 6: the diagonal tile's EARLIER updates off the merged workgroup.  The workgroup of tile (c+1, c) that finishes the diagonal
// tile (c+1, c+1) used to carry that tile's whole update queue beside its own -- two 64 x 64 x 64 products per column, 6.2 us --
// and a tile is dispatched only ~256 / (tiles per column) block columns ahead of the pivot column: at n = 3200 the queue of the
// merged tile (c updates) is longer than its lead over the chain for c = 15 .. 35, and the block column there takes 16-28 us
// instead of 14 (trace: scripts/ubench/chol_bench built -DVGG_CHOL_TRACE, n = 3200).  Now the diagonal tile's OWN workgroup --
// which used to return at once -- applies the updates k <= c - 1 of tile (c+1, c+1), leaves the partial tile in place and raises
// dready[c+1]; the merged workgroup adds only X X^T (k = c) and factors.  The sums are associated differently (the partial
// tile first, X X^T last): results move in the last bits against round 5; a solve stays bit-reproducible.
#ifndef VGG_DF_SPLIT_DIAG
#define VGG_DF_SPLIT_DIAG 1
#endif

struct DfShared {
  double D[DFB * (DFB + 1)];
  double T[DFB * (DFB + 1)];
  double scr[2 * 2 * DFB * 4];
  double rd[DFB];
  int32_t depth[kDfOrderMax], order[kDfOrderMax];
};

// flags: ready[(nbk + 1) * nbk] (tile (r, c) final), then tready[nbk]; all zero on entry.
//
// CHAIN (chain != 0): the pivot chain  factor(c) -> L[c+1][c] = tile T_c -> last update of the diagonal tile (c+1,c+1) ->
// factor(c+1)  used to cross two workgroup hand-offs per block column: T_c to the workgroup of tile (c+1,c), and its
// product on to the workgroup of the diagonal tile (store, drain, flag, poll, reload: ~4 us of a 30 us step at n = 1202).
// Where column c+1 couples to column c (first_of(c+1) <= c) the workgroup of tile (c+1,c) now finishes the diagonal tile
// (c+1,c+1) as well: while it waits for T_c it accumulates the updates k < c of BOTH tiles (the diagonal's operand
// L[c+1][k] is the one it stages anyway), and once X = L[c+1][c] exists in its registers it goes straight on to X X^T, the
// subtraction from A and the factorisation -- per element the operations the two workgroups performed before, in the
// same order.  The diagonal tile's own workgroup returns at once.  A column that starts a decoupled block (camera split, envelope) keeps
// its own workgroup, so the side-by-side chains remain.
// Workgroups per CU the kernel is compiled for (round 6: with the tile's own values loaded late both forms fit 256 registers;
// -DVGG_DF_OCC=2 / 1 force both, profiles/r06_ab_factor16_variants.jsonl + r06_ab_chol_occupancy_c5.jsonl).  ONE, for both forms:
//   * the CHAINED form (up to kDfOrderMax block columns) is a latency chain: a second workgroup on the CU slows the chain's
//     wavefront down -- its matrix instructions share the fp64 pipe (chain wavefronts at priority 3 over 1 all the same):
//     n = 1202 0.282 -> 0.310 ms, n = 3200 0.92 -> 1.00;
//   * the plain form gains where the launch is bound by how many tiles are resident ahead of the pivot column -- a DENSE
//     n = 6002: 2.85 -> 2.44 ms -- but the large systems of the product are the k-way ordered joint problems of a long video
//     (row envelope, depth-ordered launch), and there two per CU lose 1.5 % (configs[4] final joint problem, n = 6002, same box,
//     interleaved: 1.038 / 1.026 against 1.014 / 1.011 ms).
#ifndef VGG_DF_OCC
#define VGG_DF_OCC 1
#endif
constexpr int df_occupancy(bool chain) { (void)chain; return VGG_DF_OCC; }
template <bool OVERLAP, bool CHAIN>
__global__ __launch_bounds__(256, df_occupancy(CHAIN)) void chol_dataflow_kernel(double* __restrict__ A, int n, int nbk, double* __restrict__ Tinv,
                                                            int32_t* __restrict__ flags, int32_t* fail, const int32_t* skip,
                                                            int split_a, int split_b, const int32_t* __restrict__ first_blk,
                                                            DfOverlap ov, const int32_t* __restrict__ tile_map) {
  static_assert(!(OVERLAP && CHAIN), "the chained form does not take the overlap flags");
  extern __shared__ double df_smem[];
  DfShared& sh = *reinterpret_cast<DfShared*>(df_smem);
  constexpr int LD = DFB + 1;
  if (skip && *skip) return;
  // This launch is a latency chain with one wavefront per SIMD: whatever else is resident on the CU (a tile batch in
  // overlap mode; DESIGN.md section 6 on the boxes where something outside the process is) must not take its issue slots
  __builtin_amdgcn_s_setprio(df_occupancy(CHAIN) > 1 ? 1 : 3);
  // tile of this workgroup: column c holds rows c .. nbk-1 and the rhs row block nbk; launch order = (column, row), or --
  // with a row envelope -- the order of df_tile_map_kernel: columns by dependency depth, tiles outside the envelope left out
  int c = 0, t = blockIdx.x;
  if (tile_map) {
    if (t >= tile_map[0]) return;
    const int32_t e = tile_map[1 + t];
    c = e & 0xffff; t = (e >> 16) - c;
  } else {
    while (t >= nbk - c + 1) { t -= nbk - c + 1; ++c; }
  }
  const int r = c + t;                                  // r == nbk: the appended right-hand side (one row)
  const int nat_tile = c * (nbk + 1) - c * (c - 1) / 2 + t;        // position in the (column, row) order: trace slot
  (void)nat_tile;
  auto first_of = [&](int br) {
    if (first_blk) return (br < nbk) ? first_blk[br] : 0;                  // row envelope given by the caller
    return (split_b > 0 && br < nbk && DFB * br >= split_a && DFB * (br + 1) <= split_a + split_b) ? split_a / DFB : 0;
  };
  if (c < first_of(r)) return;                          // structurally zero tile (block-diagonal leading part)
  if (VGG_BW_SENTINEL && r == nbk && threadIdx.x < DFB)    // (the backward launch's hand-off buffer: see kXSentinel)
    reinterpret_cast<unsigned long long*>(df_xpub(flags, nbk))[DFB * c + threadIdx.x] = kXSentinel;
  const bool diag = (r == c);
  // the diagonal tile of column x is finished by the workgroup of tile (x, x - 1)
  auto chained = [&](int x) { return CHAIN && x >= 1 && x < nbk && first_of(x) <= x - 1; };
  const bool prep = VGG_DF_SPLIT_DIAG && diag && chained(c);      // diagonal tile finished elsewhere: this workgroup applies its early updates
  if (!VGG_DF_SPLIT_DIAG && diag && chained(c)) return;
  const bool merged = CHAIN && !diag && r == c + 1 && chained(r);
  // (two workgroups per CU: the ones on the pivot chain -- diagonal tiles and the merged first sub-diagonal ones -- go first)
  if (df_occupancy(CHAIN) > 1 && (diag || merged)) __builtin_amdgcn_s_setprio(3);
  const int kfirst = max(first_of(r), first_of(c));
  const int kstart = (merged && !VGG_DF_SPLIT_DIAG) ? first_of(r) : kfirst;     // (the diagonal tile (r,r) starts at the row's own envelope)
  const int kend = prep ? c - 1 : c;                    // (a prepared diagonal tile leaves update k = c - 1 = X X^T to the merged workgroup)
  int32_t* ready = flags;
  int32_t* tready = flags + (size_t)(nbk + 1) * nbk;
  int32_t* dready = tready + 2 * (size_t)nbk;           // [nbk] behind tready and xready: partial diagonal tile c is in place
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wy = wave >> 1, wx = wave & 1, li = lane & 15, lk = lane >> 4;
  const int r0 = (r == nbk) ? n : DFB * r, c0 = DFB * c;
  const int vr = (r == nbk) ? 1 : min(DFB, n - r0), vc = min(DFB, n - c0);
  const int dtile = r * (nbk + 1) - r * (r - 1) / 2;    // launch position of diagonal tile r (trace of a merged workgroup)
  (void)dtile;
  DF_STAMP(0);

  // Overlap with the Schur tile batches (ba.hip, options.overlap_factorization): the contributions to the columns >=
  // ov.first_col are still being summed into S2 by another stream when this launch starts; flag k says that everything
  // for the columns >= ov.wait_col[k] has landed.  A tile waits for the last flag its columns need and adds S2.
  const double* S2 = nullptr;
  if (OVERLAP && ov.S2 && r < nbk && c0 + vc > ov.first_col) {
    int need = -1;
    for (int k = 0; k < ov.num_waits; ++k)
      if (ov.wait_col[k] < c0 + vc) need = k;
    if (need >= 0) df_wait(&ov.flags[need], fail);
    S2 = ov.S2;
  }
  // the tile of A (C layout: row = 32 wy + 16 m + lk + 4 reg, col = 32 wx + 16 q + li) and the product accumulators;
  // a merged workgroup carries the diagonal tile (r,r) along (a0d, accd).
  // Round 6: the tile's own values are NOT fetched up front any more (64 registers held through the whole update loop for a
  // subtraction at its end): a0 is requested in front of the LAST update's matrix instructions (its round trip hides behind
  // them), a0d in front of the wait for the last slab of T.  The kernel then fits 256 registers: two workgroups per CU
  // (VGG_DF_OCC), and the matrix instructions of the 16 x 16 factorisation can stay in the VGPR file (no AGPR copies).
  f64x4 acc[2][2], a0[2][2], accd[2][2], a0d[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      acc[m][q] = (f64x4){0.0, 0.0, 0.0, 0.0};
      accd[m][q] = (f64x4){0.0, 0.0, 0.0, 0.0};
    }
  auto load_a0 = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int i = 32 * wy + 16 * m + lk + 4 * reg, j = 32 * wx + 16 * q + li;
          const bool in = i < vr && j < vc && (!diag || j <= i);
          a0[m][q][reg] = in ? A[(size_t)(r0 + i) * n + c0 + j] : 0.0;
          if (OVERLAP && S2 && in) a0[m][q][reg] += ld_agent(&S2[(size_t)(r0 + i) * n + c0 + j]);
        }
  };
  auto load_a0d = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int i = 32 * wy + 16 * m + lk + 4 * reg, j = 32 * wx + 16 * q + li;
          if (VGG_DF_SPLIT_DIAG)                        // (written by the diagonal tile's workgroup during this launch)
            a0d[m][q][reg] = (merged && i < vr && j < vr && j <= i) ? ld_agent(&A[(size_t)(r0 + i) * n + r0 + j]) : 0.0;
          else
            a0d[m][q][reg] = (merged && i < vr && j < vr && j <= i) ? A[(size_t)(r0 + i) * n + r0 + j] : 0.0;
        }
  };

  // sum += (rows of bufA) (rows of bufB)^T over the 64 columns staged in LDS.  k-index permutation of the MFMA steps: lane
  // group lk supplies the 16 consecutive columns 16 lk .. 16 lk + 15 of the operand row (same permutation for both operands)
  auto multiply_staged = [&](const double* bufA, const double* bufB, f64x4 (&sum)[2][2]) __attribute__((always_inline)) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      double a[2][8], b[2][8];
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) {
        const int kk = 4 * (8 * half + s8) + lk;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          a[m][s8] = bufA[(32 * wy + 16 * m + li) * LD + kk];
          b[m][s8] = bufB[(32 * wx + 16 * m + li) * LD + kk];
        }
      }
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            sum[m][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][s8], b[q][s8], sum[m][q], 0, 0, 0);
    }
  };

  // ORDER of the left-looking updates.  Where decoupled blocks meet (camera split [A, B, rest], k-way envelope) a tile of
  // the rows behind them has update columns from several pivot chains that finish at different times; taken in ascending
  // order, the columns of the chain that started later (B: available long ago) queue behind the last column of the longer
  // one (A), and that queue sits on the critical path (c3: four updates, ~20 us, after A's last column).  So the columns
  // are taken by DEPTH in the dependence graph -- depth(k) = 1 + max depth over the block columns row k couples to, a
  // stand-in for "when column k completes" -- then by index: a fixed order (a function of the envelope alone: results are
  // reproducible run to run), equal to the ascending one for a dense matrix.  Every workgroup derives it for itself
  // (~3 us at 50 block columns, hidden behind its first wait); past kDfOrderMax block columns the launch is bound by the
  // number of resident workgroups and those microseconds would add up (c5, 94 block columns: +6 %), so the order stays
  // ascending there.
  const int nupd = max(kend - kstart, 0);
  const bool by_depth = CHAIN && nupd >= 2 && nbk <= kDfOrderMax;
  if (by_depth) {
    if (wave == 0) {
      volatile int32_t* dep = sh.depth;              // (lane 0 writes what the other lanes read one step later)
      for (int k = 0; k < c; ++k) {
        const int f = first_of(k);
        int m = 0;
        for (int j = f + lane; j < k; j += 64) m = max(m, dep[j]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
        if (lane == 0) dep[k] = m + 1;
      }
      for (int idx = lane; idx < nupd; idx += 64) {
        const int k = kstart + idx, dk = dep[k];
        int rank = 0;
        for (int k2 = kstart; k2 < kend; ++k2) { const int d2 = dep[k2]; rank += (d2 < dk) || (d2 == dk && k2 < k); }
        sh.order[rank] = k;
      }
    }
    __syncthreads();
  }
  if (nupd <= 0) load_a0();
  // left-looking updates: acc += L[r][k] L[c][k]^T  (merged: and accd += L[r][k] L[r][k]^T)
  // -DVGG_DF_PREFETCH=1 (round 6, MEASURED, off): the operand tiles of update t + 1 requested BEFORE the matrix instructions of
  // update t whenever their flags are already up (a relaxed look, no wait; every wavefront decides for itself -- it stages its own
  // rows -- and waits alone where it has to), written to LDS behind them.  The idea: at n = 3200 a block column takes 18-21 us
  // against the 14 us of the pivot chain, and a tile works through ~40 updates of 3.2 us (round trip + matrix instructions) with
  // ~5 block columns resident ahead.  scripts/ubench/chol_bench, same box, two rounds: n = 3200 0.926 -> 0.913 ms, with the
  // configs[3] camera split 0.826 -> 0.818, n = 1202 unchanged, n = 6002 dense 2.41 -> 2.63 (the extra loads in flight cost more
  // than they hide there): the update queues are not what sets the step.  Same sums in the same order either way.
#ifndef VGG_DF_PREFETCH
#define VGG_DF_PREFETCH 0
#endif
  {
    double* bufA = sh.D;
    double* bufB = diag ? sh.D : sh.T;
    double va[16], vb[16];
    auto upd_k = [&](int t) { return by_depth ? sh.order[t] : kstart + t; };
    auto request = [&](int k) __attribute__((always_inline)) {
      const int k0 = DFB * k;
      const bool do_tile = k >= kfirst;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (tid >> 6) + 4 * q, col = lane;
        va[q] = (row < vr) ? ld_agent(&A[(size_t)(r0 + row) * n + k0 + col]) : 0.0;
        if (!diag) vb[q] = (do_tile && row < vc) ? ld_agent(&A[(size_t)(c0 + row) * n + k0 + col]) : 0.0;
      }
    };
    // are the operand tiles of update column k final?  (wave-uniform: lane 0 looks, everybody gets its answer)
    auto flags_up = [&](int k) -> bool {
      int up = 0;
      if (lane == 0) {
        up = __hip_atomic_load(&ready[(size_t)r * nbk + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 1;
        if (up && !diag && k >= kfirst) up = __hip_atomic_load(&ready[(size_t)c * nbk + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 1;
      }
      return __builtin_amdgcn_readfirstlane(up) != 0;
    };
    // wait of ONE wavefront (lane 0 polls; bounded like df_wait_ge)
    auto wave_wait = [&](const int32_t* flag) {
      if (lane == 0) {
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 1) {
          __builtin_amdgcn_s_sleep(4);
          ++spins;
          const bool lost = fail && (spins & 1023) == 0 && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2;
          if (spins > kSpinLimit || lost) { if (fail) __hip_atomic_store(fail, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
      }
    };
    bool have = false;                                   // this wavefront's registers hold the operands of the next update
    for (int t = 0; t < nupd; ++t) {
      const int k = upd_k(t);
      const bool do_tile = k >= kfirst;
      if (!have) {                                       // (not requested ahead: wait for the flags, this wavefront alone -- it
        wave_wait(&ready[(size_t)r * nbk + k]);          //  stages its own rows, and the barrier behind the LDS writes is the one
        if (!diag && do_tile) wave_wait(&ready[(size_t)c * nbk + k]);   // all four meet at)
        request(k);
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = (tid >> 6) + 4 * q, col = lane;
        bufA[row * LD + col] = va[q];
        if (!diag) bufB[row * LD + col] = vb[q];
      }
      __syncthreads();
      have = false;
      if (VGG_DF_PREFETCH && t + 1 < nupd) {
        const int kn = upd_k(t + 1);
        if (flags_up(kn)) { request(kn); have = true; }
      }
      if (t == nupd - 1) load_a0();                      // (in flight behind the matrix instructions below)
      if (do_tile) multiply_staged(bufA, bufB, acc);
      if (merged && !VGG_DF_SPLIT_DIAG) multiply_staged(bufA, bufA, accd);
      __syncthreads();                                   // operands consumed: the buffers may be refilled
    }
  }

  DF_STAMP(1);                                           // all updates applied
  if (prep) {
    // partial diagonal tile (updates k <= c - 2) back in place, then the flag the merged workgroup of tile (c, c - 1) waits for
    if (nupd > 0) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int i = 32 * wy + 16 * m + lk + 4 * reg, j = 32 * wx + 16 * q + li;
            if (i < vr && j < vc && j <= i) st_agent(&A[(size_t)(r0 + i) * n + c0 + j], a0[m][q][reg] - acc[m][q][reg]);
          }
    }
    df_publish(&dready[c]);
    return;
  }
  // tile value = A - sum: into LDS, row-major (the diagonal tile factors there; the others need it as an MFMA operand)
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int i = 32 * wy + 16 * m + lk + 4 * reg, j = 32 * wx + 16 * q + li;
        double v = a0[m][q][reg] - acc[m][q][reg];
        if (diag && (i >= vc || j >= vc)) v = (i == j) ? 1.0 : 0.0;       // ragged last block: identity padding
        sh.D[i * LD + j] = v;
      }
  __syncthreads();

  // the factorisation of the diagonal block in sh.D, then T first -- the tiles of the column wait for it -- and L_bb
  // (which nobody waits for: it only matters through T_b and as output)
  // T is handed on in four slabs of 16 columns, tready[bc] = number of slabs out: the tiles of the column multiply by the
  // first three while the factorisation is still running (columns 16 kb .. of T are final after step kb)
  auto factor_and_publish = [&](int bc, int b0, int vb, int trace_tile, int32_t* late_flag) __attribute__((always_inline)) {
    (void)trace_tile;
    double* Tg = Tinv + (size_t)bc * DFB * DFB;
#ifdef VGG_CHOL_PAIRS                              // A/B: the round-2 form, two 32 x 32 blocks, T in one piece
    factor64(sh.D, sh.T, sh.rd, sh.scr, fail);
    DF_STAMP_AT(trace_tile, 2);                          // factored
    for (int e = tid; e < DFB * DFB; e += 256) {
      const int i = e / DFB, j = e % DFB;
      st_agent(&Tg[e], (j >= i) ? sh.T[i * LD + j] : 0.0);
    }
#else
    factor64_blocked(sh.D, sh.T, sh.scr, fail, [&](int kb) {
      // (the flag of the merged workgroup's sub-diagonal tile: its stores were drained in front of the first barrier inside)
      if (kb == 0 && late_flag && lane == 0) __hip_atomic_store(late_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      // one wavefront: 64 rows x 16 columns (rows below the diagonal block are zero), its own drain, then the count
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int i = 4 * q + (lane >> 4), j = 16 * kb + (lane & 15);
        st_agent(&Tg[i * DFB + j], (i < 16 * (kb + 1)) ? sh.T[i * LD + j] : 0.0);      // (below the diagonal block: zero)
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(&tready[bc], kb + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    });
    DF_STAMP_AT(trace_tile, 2);                          // factored
    for (int e = tid; e < DFB * 16; e += 256) {
      const int i = e / 16, j = 48 + e % 16;
      st_agent(&Tg[i * DFB + j], sh.T[i * LD + j]);
    }
#endif
    df_publish(&tready[bc], 4);
    DF_STAMP_AT(trace_tile, 3);                          // published
    for (int e = tid; e < DFB * DFB; e += 256) {
      const int i = e / DFB, j = e % DFB;
#ifdef VGG_DF_L_AGENT
      if (j <= i && i < vb) st_agent(&A[(size_t)(b0 + i) * n + b0 + j], sh.D[i * LD + j]);
#else
      if (j <= i && i < vb) A[(size_t)(b0 + i) * n + b0 + j] = sh.D[i * LD + j];
#endif
    }
    // (ready[b][b] is never waited for: L_bb only matters through T_b)
  };
  if (diag) {
    factor_and_publish(c, c0, vc, nat_tile, nullptr);
    return;
  }

  // off-diagonal (and rhs) tile: X = tile * T_c, T_c = L_cc^-T;  X[i][j] = sum_k tile[i][k] T[k][j], by slabs of 16 columns
  // as the column's workgroup hands them on (T is upper triangular: slab kb needs the rows k < 16 (kb + 1) only, in the
  // plain k order -- the sums are those of the one-piece product).  Wavefront w owns the rows 16 w .. 16 w + 15 of X.  A
  // merged workgroup adds each slab's share of X X^T to the diagonal tile's sum at once (X through sh.T): when the last
  // slab arrives, a quarter of the product and a quarter of X X^T are left to do.
  const double* Tc = Tinv + (size_t)c * DFB * DFB;
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    if (kb == 3 && merged) {                             // (the diagonal tile's values: needed behind the last slab)
      if (VGG_DF_SPLIT_DIAG) df_wait(&dready[r], fail);    // (raised long ago: its last operand is a tile of column c - 1)
      load_a0d();
    }
    df_wait_ge(&tready[c], kb + 1, fail);
    if (kb == 3) DF_STAMP(2);                            // the last columns of T arrived
    const int ns = 4 * (kb + 1);
    double a[16], b[16];
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4)
      if (s4 < ns) {
        a[s4] = sh.D[(16 * wave + li) * LD + 4 * s4 + lk];
        b[s4] = ld_agent(Tc + (size_t)(4 * s4 + lk) * DFB + 16 * kb + li);
      }
    f64x4 xa = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4)
      if (s4 < ns) xa = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s4], b[s4], xa, 0, 0, 0);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int i = 16 * wave + lk + 4 * reg, j = 16 * kb + li;
      if (i < vr && j < vc) st_agent(&A[(size_t)(r0 + i) * n + c0 + j], xa[reg]);
      if (merged) sh.T[i * LD + j] = xa[reg];            // (rows past the end of the matrix are zero)
    }
    if (merged) {
      lds_barrier();                                     // (X went to sh.T; its stores to A drain behind the products)
      double a2[2][4], b2[2][4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          a2[m][s4] = sh.T[(32 * wy + 16 * m + li) * LD + 16 * kb + 4 * s4 + lk];
          b2[m][s4] = sh.T[(32 * wx + 16 * m + li) * LD + 16 * kb + 4 * s4 + lk];
        }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int q = 0; q < 2; ++q)
            accd[m][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[m][s4], b2[q][s4], accd[m][q], 0, 0, 0);
    }
  }
  DF_STAMP(4);                                           // product done, stores issued
  if (!merged) {
    df_publish(&ready[(size_t)r * nbk + c]);
    DF_STAMP(3);
    return;
  }
  // merged: the diagonal tile (r,r) has its last update; its value into sh.D (every wavefront is past its reads of the
  // tile there: the barrier of the last slab), then its factorisation
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int i = 32 * wy + 16 * m + lk + 4 * reg, j = 32 * wx + 16 * q + li;
        double v = a0d[m][q][reg] - accd[m][q][reg];
        if (i >= vr || j >= vr) v = (i == j) ? 1.0 : 0.0;
        sh.D[i * LD + j] = v;
      }
  // -DVGG_DF_LATE_XFLAG=1 (measured twice, off): the flag of tile (r, c) = X raised from INSIDE the factorisation, behind its
  // first 16 x 16 step, where the drain of the X stores is free -- 0.7 us off the factor path, but the flag is ~2.4 us late for
  // tile (r+1, r), whose wait for it + last update + first three slab products are a second chain per block column.  Round 4:
  // 0.273 -> 0.294 ms at n = 1202.  Round 6, with that tile's workgroup relieved of the diagonal tile's update queue
  // (VGG_DF_SPLIT_DIAG: its chain ~11.5 us against the 14.3 of the factor path): 0.260 -> 0.286 ms at n = 1202, 0.797 -> 0.868
  // at n = 3200 -- the block column goes to 16 us: 11.5 + 2.4 + the slab it then waits for is past the factor path again.
#ifndef VGG_DF_LATE_XFLAG
#define VGG_DF_LATE_XFLAG 0
#endif
  if (VGG_DF_LATE_XFLAG) lds_barrier();                   // (the reads of X in sh.T are through; the diagonal tile's value is in sh.D)
  else df_publish(&ready[(size_t)r * nbk + c]);
  DF_STAMP(3);
  DF_STAMP_AT(dtile, 1);                                 // diagonal tile r: updates applied
  factor_and_publish(r, r0, vr, dtile, VGG_DF_LATE_XFLAG ? &ready[(size_t)r * nbk + c] : nullptr);
}

// Backward substitution L^T x = z in dataflow form: one workgroup per 64-column block c (launched last block first, so
// that a workgroup only waits for EARLIER ones).  Workgroup c accumulates sum_r L[r][c]^T x_r over the block rows r > c
// inside the envelope as their x_r are published (fixed order r = last .. c + 1: deterministic), then x_c = T_c (z_c - sum)
// with T_c = L_cc^-T from the factorisation.  The tile L[r][c] is fetched BEFORE the wait for x_r, so the step on the
// critical path is one 64 x 64 mat-vec + the T mat-vec + a 512-byte hand-off.  (The single-workgroup kernel above reads
// the whole factor through one CU: 0.14 ms at n = 1202, ~3.5 ms at n = 6002.)
__global__ __launch_bounds__(256) void chol_backward_dataflow_kernel(const double* __restrict__ L, double* __restrict__ b, int n,
                                                                     int nbk, const double* __restrict__ Tinv,
                                                                     int32_t* __restrict__ xready, int32_t* fail,
                                                                     const int32_t* skip, int split_a, int split_b,
                                                                     const int32_t* __restrict__ first_blk) {
  constexpr int LD = DFB + 1;
  __shared__ double Ts[DFB * LD];
  __shared__ double xs[DFB], part[4][DFB], w[DFB];
  if (skip && *skip) return;
  __builtin_amdgcn_s_setprio(3);                   // (as in chol_dataflow_kernel)
  const int c = nbk - 1 - (int)blockIdx.x, c0 = DFB * c, vc = min(DFB, n - c0);
  const int tid = threadIdx.x, j = tid & 63, q = tid >> 6;          // thread: column j of the tile, rows 16 q .. 16 q + 15
  auto first_of = [&](int br) {
    if (first_blk) return first_blk[br];
    return (split_b > 0 && DFB * br >= split_a && DFB * (br + 1) <= split_a + split_b) ? split_a / DFB : 0;
  };
  double* xpub = df_xpub(xready - ((size_t)(nbk + 1) * nbk + nbk), nbk);      // (xready sits at that offset of the flags)
  for (int e = tid; e < DFB * DFB; e += 256) Ts[(e / DFB) * LD + e % DFB] = Tinv[(size_t)c * DFB * DFB + e];
  double acc = 0.0;
  auto load_tile = [&](int r, double (&t)[16]) {
    const int r0 = DFB * r, vr = min(DFB, n - r0);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = 16 * q + i;
      t[i] = (row < vr && j < vc) ? L[(size_t)(r0 + row) * n + c0 + j] : 0.0;
    }
  };
  int r = nbk - 1;
  while (r > c && first_of(r) > c) --r;
  double cur[16], nxt[16];
  if (r > c) load_tile(r, cur);
  while (r > c) {
    int rn = r - 1;
    while (rn > c && first_of(rn) > c) --rn;
    if (rn > c) load_tile(rn, nxt);                       // in flight while this workgroup waits for x_r
    if (VGG_BW_SENTINEL) {
      if (tid < DFB) {
        double v = 0.0;
        if (DFB * r + tid < n) {
          int spins = 0;
          for (;;) {
            v = ld_agent(&xpub[DFB * r + tid]);
            if ((unsigned long long)__double_as_longlong(v) != kXSentinel) break;
            __builtin_amdgcn_s_sleep(1);
            ++spins;
            const bool lost = fail && (spins & 1023) == 0 && __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2;
            if (spins > kSpinLimit || lost) { if (fail) __hip_atomic_store(fail, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v = 0.0; break; }
          }
        }
        xs[tid] = v;
      }
    } else {
      df_wait(&xready[r], fail);
      if (tid < DFB) xs[tid] = (DFB * r + tid < n) ? ld_agent(&b[DFB * r + tid]) : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += cur[i] * xs[16 * q + i];
    __syncthreads();                                      // xs consumed
#pragma unroll
    for (int i = 0; i < 16; ++i) cur[i] = nxt[i];
    r = rn;
  }
  part[q][j] = acc;
  __syncthreads();
  if (tid < DFB) w[tid] = ((tid < vc) ? b[c0 + tid] : 0.0) - (((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid]);
  __syncthreads();
  // x_c[i] = sum_{j >= i} T[i][j] w[j]: thread (i = tid & 63, quarter q of the columns)
  {
    const int i = tid & 63;
    double s0 = 0.0;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) s0 += Ts[i * LD + 16 * q + jj] * w[16 * q + jj];
    part[q][i] = s0;
  }
  __syncthreads();
  if (tid < vc) {
    const double xv = ((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid];
    if (VGG_BW_SENTINEL) { st_agent(&xpub[c0 + tid], xv); b[c0 + tid] = xv; }
    else st_agent(&b[c0 + tid], xv);
  }
  if (!VGG_BW_SENTINEL) df_publish(&xready[c]);
}

// LAUNCH ORDER with a row envelope.  In (column, row) order the tiles of a pivot chain that starts late in the matrix
// (k-way camera order: the second, third ... interior run) are dispatched behind every tile of the columns before it, and
// those wait, resident, for their own chain: at c5 (94 block columns, three side-by-side chains + separators) the 512
// workgroup slots of the chip were taken by the first chain's columns and the chains ran one after the other (1.56 ms = 94
// block columns back to back).  Here the columns are ordered by DEPTH in the dependence graph -- depth(c) = 1 + max depth
// of the columns a tile of column c waits for -- then by index, tiles outside the envelope and the diagonal tiles that a
// merged workgroup finishes are left out.  Still a topological order: a workgroup only ever waits for earlier ones.
// map[0] = number of tiles, map[1 + i] = c | r << 16.  One workgroup; the depth recursion by its first wavefront.
constexpr int kDfMapCols = 1024;
__global__ __launch_bounds__(256) void df_tile_map_kernel(const int32_t* __restrict__ first_blk, int nbk, int chain, int32_t* __restrict__ map) {
  __shared__ int32_t fo[kDfMapCols + 1], dep[kDfMapCols], cnt[kDfMapCols], off[kDfMapCols];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int r = tid; r <= nbk; r += 256) fo[r] = (r < nbk) ? first_blk[r] : 0;
  __syncthreads();
  auto chained = [&](int x) { return chain && x >= 1 && x < nbk && fo[x] <= x - 1; };
  if (tid < 64) {
    volatile int32_t* d = dep;
    for (int c = 0; c < nbk; ++c) {
      int lo = fo[c];
      if (chained(c + 1)) lo = min(lo, fo[c + 1]);           // (the merged workgroup of tile (c + 1, c) also takes row c + 1's updates)
      int m = 0;
      for (int k = lo + lane; k < c; k += 64) m = max(m, d[k]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
      if (lane == 0) d[c] = m + 1;
    }
  }
  __syncthreads();
  for (int c = tid; c < nbk; c += 256) {
    int n = 0;
    for (int r = c; r <= nbk; ++r) n += (fo[r] <= c) && (VGG_DF_SPLIT_DIAG || !(r == c && chained(c)));
    cnt[c] = n;
  }
  __syncthreads();
  for (int c = tid; c < nbk; c += 256) {
    int o = 0;
    const int dc = dep[c];
    for (int k = 0; k < nbk; ++k) { const int dk = dep[k]; if (dk < dc || (dk == dc && k < c)) o += cnt[k]; }
    off[c] = o;
  }
  __syncthreads();
  for (int c = tid; c < nbk; c += 256) {
    int o = off[c];
    for (int r = c; r <= nbk; ++r)
      if ((fo[r] <= c) && (VGG_DF_SPLIT_DIAG || !(r == c && chained(c)))) map[1 + o++] = c | (r << 16);
  }
  if (tid == 0) { int tot = 0; for (int c = 0; c < nbk; ++c) tot += cnt[c]; map[0] = tot; }
}

// LDS request of the dataflow kernel: what it uses, or -- compiled for ONE workgroup per CU -- more than half a CU's 160 KB, so
// that the hardware cannot place a second one beside a pivot chain whatever the register count of the day allows (round 6: the
// plain form fits 256 registers since the tile values are loaded late, and two per CU measured slower on the product's systems)
static size_t df_lds_bytes() {
  const size_t need = sizeof(DfShared), half = 82 * 1024;
  return (VGG_DF_OCC == 1 && need < half) ? half : need;
}
static size_t dataflow_flag_count(int n) {
  const int nbk = div_up(n, DFB);
  return (size_t)(nbk + 1) * nbk + 3 * (size_t)nbk;        // ready[(nbk + 1) nbk], tready[nbk], xready[nbk], dready[nbk]
}
static size_t dataflow_tile_count(int n) {
  const size_t nbk = div_up(n, DFB);
  return nbk * (nbk + 1) / 2 + nbk;
}
static size_t dataflow_workspace_bytes(int n) {     // T blocks | flags | launch-order map (1 + tiles)
  return (size_t)div_up(n, DFB) * DFB * DFB * sizeof(double) + (dataflow_flag_count(n) + 1 + dataflow_tile_count(n) + 2) * sizeof(int32_t) +
         (size_t)div_up(n, DFB) * DFB * sizeof(double) + 256;      // ... | xpub (df_xpub)
}

static inline int block_size_for(int n) { (void)n; return 32; }

// workspace = the inverted diagonal blocks T_j = L_jj^-T, NB x NB doubles each
size_t cholesky_workspace_bytes(int n) {
  const int nb = block_size_for(n);
  const size_t legacy = (size_t)div_up(n, nb) * nb * nb * sizeof(double) + 256;
  const size_t df = dataflow_workspace_bytes(n);
  return legacy > df ? legacy : df;
}

// VGG_CHOL_LEGACY=1 in the environment selects the multi-launch path (A/B measurements)
static bool use_dataflow(int n) {
  static const bool legacy = [] { const char* e = getenv("VGG_CHOL_LEGACY"); return e && e[0] == '1'; }();
  // (round 6: every size -- rounds 3-5 kept the multi-launch path below two 64-blocks, where it is 1.2-1.9 x slower: 76 -> 39 us
  //  at the n = 102 of a 17-frame video window, 44 -> 25 at n = 54; VGG_DF_MIN_N restores a threshold for A/B measurements)
  static const int min_n = [] { const char* e = getenv("VGG_DF_MIN_N"); return e ? atoi(e) : 1; }();
  return !legacy && n >= min_n;
}

// raises one overlap flag (a launch of its own behind a tile batch: the batch's stores are then visible device-wide)
__global__ void df_signal_kernel(int32_t* flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
void dataflow_signal(int32_t* flag, hipStream_t st) { df_signal_kernel<<<1, 1, 0, st>>>(flag); }

// The flags of the dataflow factorisation inside `ws` (nullptr when a system of n unknowns takes the multi-launch path).  A
// caller that zeroes them itself -- *count int32, any time between the previous solve in this workspace and the next one
// (bundle adjustment: in a kernel it launches anyway) -- passes flags_cleared = true and saves the fill launch.
int32_t* cholesky_dataflow_flags(double* ws, int n, size_t* count) {
  if (!ws || !use_dataflow(n)) return nullptr;
  if (count) *count = dataflow_flag_count(n);
  return reinterpret_cast<int32_t*>(ws + (size_t)div_up(n, DFB) * DFB * DFB);
}

static int enqueue_dataflow(double* A, double* b, int n, double* ws, int32_t* device_fail, const int32_t* skip, hipStream_t st,
                            int split_a, int split_b, const int32_t* first_blk, const CholOverlap* overlap, bool flags_cleared) {
  const int nbk = div_up(n, DFB);
  double* Tinv = ws;
  int32_t* flags = reinterpret_cast<int32_t*>(ws + (size_t)nbk * DFB * DFB);
  // VGG_CHOL_CHAIN=0 / 1 in the environment: every diagonal tile in a workgroup of its own / chained (A/B measurements)
  // (default: chained up to 64 block columns -- beyond, the dense left-looking update queue of a late column's workgroup
  // is what the chain waits for, and a merged workgroup carries two of them; measured in DESIGN.md section 6)
  static const int chain_env = [] { const char* e = getenv("VGG_CHOL_CHAIN"); return !e ? -1 : (e[0] == '0' ? 0 : 1); }();
  const int chain = chain_env >= 0 ? chain_env : (nbk <= kDfOrderMax ? 1 : 0);
  if (!flags_cleared && hipMemsetAsync(flags, 0, dataflow_flag_count(n) * sizeof(int32_t), st) != hipSuccess) return VGG_ERR_HIP;
  static bool attr_set = false;
  if (!attr_set) {
    const void* kernels[3] = {reinterpret_cast<const void*>(&chol_dataflow_kernel<false, false>),
                              reinterpret_cast<const void*>(&chol_dataflow_kernel<false, true>),
                              reinterpret_cast<const void*>(&chol_dataflow_kernel<true, false>)};
    for (const void* k : kernels)
      if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)df_lds_bytes()) != hipSuccess) return VGG_ERR_HIP;
    attr_set = true;
  }
  if (!(split_a >= DFB && split_b >= DFB && split_a % DFB == 0 && split_a + split_b <= n)) split_a = split_b = 0;
  const int tiles = nbk * (nbk + 1) / 2 + nbk;
  // launch order by dependency depth when the caller gave a row envelope (VGG_CHOL_TILE_MAP=0: the plain order, A/B)
  static const bool map_on = [] { const char* e = getenv("VGG_CHOL_TILE_MAP"); return !(e && e[0] == '0'); }();
  int32_t* tile_map = nullptr;
  if (first_blk && map_on && nbk <= kDfMapCols && !(overlap && overlap->dev_flags)) {
    tile_map = flags + dataflow_flag_count(n);
    df_tile_map_kernel<<<1, 256, 0, st>>>(first_blk, nbk, chain, tile_map);
  }
  DfOverlap ov = {};
  if (overlap && overlap->dev_flags) {
    ov.S2 = overlap->S2; ov.flags = overlap->dev_flags; ov.first_col = overlap->first_col; ov.num_waits = overlap->num_waits;
    for (int k = 0; k < overlap->num_waits && k < 8; ++k) ov.wait_col[k] = overlap->wait_col[k];
  }
  if (ov.S2) chol_dataflow_kernel<true, false><<<tiles, 256, df_lds_bytes(), st>>>(A, n, nbk, Tinv, flags, device_fail, skip, split_a, split_b, first_blk, ov, tile_map);
  else if (chain) chol_dataflow_kernel<false, true><<<tiles, 256, df_lds_bytes(), st>>>(A, n, nbk, Tinv, flags, device_fail, skip, split_a, split_b, first_blk, ov, tile_map);
  else chol_dataflow_kernel<false, false><<<tiles, 256, df_lds_bytes(), st>>>(A, n, nbk, Tinv, flags, device_fail, skip, split_a, split_b, first_blk, ov, tile_map);
  int32_t* xready = flags + (size_t)(nbk + 1) * nbk + nbk;
  chol_backward_dataflow_kernel<<<nbk, 256, 0, st>>>(A, b, n, nbk, Tinv, xready, device_fail, skip, split_a, split_b, first_blk);
  if (hipGetLastError() != hipSuccess) return VGG_ERR_HIP;
  return VGG_OK;
}

// If b is stored directly behind A (b == A + n*n, i.e. "row n" of an (n+1) x n matrix) the right-hand side
// rides through the factorisation as one more panel row: the panel solve and the trailing update then
// perform the forward substitution for free and only L^T y = z is left.
template <int NB>
static int enqueue(double* A, double* b, int n, double* inv_blocks, int32_t* device_fail, const int32_t* skip,
                   hipStream_t st, const CholOverlap* ov, int split_a, int split_b) {
  int next_wait = 0;
  // before the panel over columns [k0, k1): wait for the producers of those columns; S2 only where it can be non-zero
  auto panel_s2 = [&](int k1) -> const double* {
    if (!ov) return nullptr;
    while (next_wait < ov->num_waits && ov->wait_col[next_wait] < k1) (void)hipStreamWaitEvent(st, ov->wait_ev[next_wait++], 0);
    return (k1 > ov->first_col) ? ov->S2 : nullptr;
  };
  const bool fused_rhs = (b == A + (size_t)n * n);
  const int nrows = fused_rhs ? n + 1 : n;
  // fast backward solve: y and one T block in LDS (up to 150 KB); larger systems use the single-workgroup kernel
  const size_t back_lds = ((size_t)div_up(n, NB) * NB + NB * (NB + 1) + NB) * sizeof(double);
  bool lds_backward = back_lds <= 150 * 1024;
  if (lds_backward && back_lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&chol_backward_kernel<NB>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)back_lds) != hipSuccess)
      lds_backward = false;
  }
  // one fused double step: 64 columns = one panel launch + one K = 64 trailing update
  auto fused_step = [&](int k0, int k0_second) {
    const int rows_panel = nrows - k0 - 64;
    const double* S2 = panel_s2((k0_second >= 0 ? k0_second : k0) + 64);
    const dim3 grid((rows_panel > 0 ? div_up(rows_panel, 64) : 1) + 1, k0_second >= 0 ? 2 : 1);
    chol_panel2_kernel<<<grid, 256, 0, st>>>(A, n, nrows, k0, device_fail, skip, inv_blocks, S2, k0_second);
    for (int which = 0; which < (k0_second >= 0 ? 2 : 1); ++which) {
      const int k = which ? k0_second : k0;
      const int rows_below = nrows - k - 64;
      if (rows_below > 0 && k + 64 < n) {
        const int T = div_up(rows_below, 32);
        const int tiles = T * (T + 1) / 2;
        chol_update_kernel<64><<<div_up(tiles, 4), 256, 0, st>>>(A, n, nrows, k, tiles, skip);
      }
    }
  };
  int k0 = 0;
  if (NB == 32) {
    // Block-diagonal leading part (split_a, split_b > 0): the column sets A = [0, split_a) and B = [split_a, split_a +
    // split_b) do not couple (A[B rows][A cols] = 0 -- the caller ordered the unknowns that way, ba.py), so panel s of A
    // and panel s of B are independent: they share one launch and their pivot chains run side by side.  The trailing
    // updates stay separate launches (an A panel has zeros in the rows of B, so its update leaves B's columns
    // bit-for-bit untouched).  What is left of A, then of B, then everything else follows in the usual order.
    int lock = 0;
    if (!ov && split_a >= 64 && split_b >= 64 && split_a % 64 == 0 && split_a + split_b <= n) lock = min(split_a, split_b) / 64;
    for (int s2 = 0; s2 < lock; ++s2) fused_step(64 * s2, split_a + 64 * s2);
    if (lock > 0) {
      for (k0 = 64 * lock; k0 + 64 <= split_a; k0 += 64) fused_step(k0, -1);     // the rest of A (a multiple of 64)
      k0 = split_a + 64 * lock;                                                  // the rest of B and everything behind it
    }
    // fused double steps while at least 64 columns remain
    for (; k0 + 64 <= n; k0 += 64) fused_step(k0, -1);
  }
  for (; k0 < n; k0 += NB) {
    const int nb = (n - k0 < NB) ? n - k0 : NB;
    const int rows_panel = nrows - k0 - nb;
    const int grid = (rows_panel > 0 ? div_up(rows_panel, 256) : 1) + 1;
    const double* S2 = panel_s2(k0 + nb);
    chol_panel_kernel<NB><<<grid, 256, 0, st>>>(A, n, nrows, k0, device_fail, skip, inv_blocks, S2);
    const int rows_below = nrows - k0 - NB;
    if (rows_below > 0 && k0 + NB < n) {
      const int T = div_up(rows_below, 32);
      const int tiles = T * (T + 1) / 2;
      chol_update_kernel<NB><<<div_up(tiles, 4), 256, 0, st>>>(A, n, nrows, k0, tiles, skip);
    }
  }
  (void)panel_s2(n + 1);                        // (n == 0 cannot happen; every producer is waited for by now)
  if (!fused_rhs) chol_solve_kernel<NB><<<1, 256, 0, st>>>(A, b, n, 1, lds_backward ? 0 : 1, skip);
  else if (!lds_backward) chol_solve_kernel<NB><<<1, 256, 0, st>>>(A, b, n, 0, 1, skip);
  if (lds_backward) chol_backward_kernel<NB><<<1, kBackThreads, back_lds, st>>>(A, b, n, inv_blocks, skip);
  if (hipGetLastError() != hipSuccess) return VGG_ERR_HIP;
  return VGG_OK;
}

int cholesky_solve_enqueue(double* A, double* b, int n, double* inv_blocks, int32_t* device_fail, const int32_t* skip,
                           hipStream_t st, const CholOverlap* overlap, int split_a, int split_b, const int32_t* first_blk,
                           bool flags_cleared) {
  if (!inv_blocks) return VGG_ERR_INVALID_ARGUMENT;
  if ((!overlap || overlap->dev_flags) && b == A + (size_t)n * n && use_dataflow(n))
    return enqueue_dataflow(A, b, n, inv_blocks, device_fail, skip, st, split_a, split_b, first_blk, overlap, flags_cleared);
  return enqueue<32>(A, b, n, inv_blocks, device_fail, skip, st, overlap, split_a, split_b);
}

}  // namespace vgg

extern "C" {
size_t vgg_cholesky_workspace_bytes(int n) { return n > 0 ? vgg::cholesky_workspace_bytes(n) : 0; }

int vgg_cholesky_solve(double* A, double* b, int n, void* workspace, int32_t* device_fail, void* stream) {
  if (n <= 0 || !A || !b || !workspace) return VGG_ERR_INVALID_ARGUMENT;
  return vgg::cholesky_solve_enqueue(A, b, n, (double*)workspace, device_fail, nullptr, (hipStream_t)stream, nullptr, 0, 0, nullptr, false);
}

int vgg_cholesky_solve_envelope(double* A, double* b, int n, const int32_t* first_blk, void* workspace, int32_t* device_fail,
                                void* stream) {
  if (n <= 0 || !A || !b || !workspace || !first_blk) return VGG_ERR_INVALID_ARGUMENT;
  return vgg::cholesky_solve_enqueue(A, b, n, (double*)workspace, device_fail, nullptr, (hipStream_t)stream, nullptr, 0, 0, first_blk, false);
}

int vgg_cholesky_solve_split(double* A, double* b, int n, int split_a, int split_b, void* workspace, int32_t* device_fail,
                             void* stream) {
  if (n <= 0 || !A || !b || !workspace || split_a < 0 || split_b < 0 || split_a + split_b > n) return VGG_ERR_INVALID_ARGUMENT;
  return vgg::cholesky_solve_enqueue(A, b, n, (double*)workspace, device_fail, nullptr, (hipStream_t)stream, nullptr, split_a,
                                     split_b, nullptr, false);
}
}
