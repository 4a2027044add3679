// Dense fp64 Cholesky solve of the reduced camera system (6C+K unknowns) on gfx950.
//
// In the reference this is the DENSE_SCHUR / SPARSE_SCHUR factorisation inside Ceres, reached through
// pycolmap.bundle_adjustment (vggsfm/utils/triangulation.py:213,1050,1142).  Here: right-looking blocked
// Cholesky, NB = 32, two launches per block column:
//   panel  : every workgroup re-factors the 32x32 diagonal block in LDS (outer-product form, ONE barrier per
//            column, reciprocal instead of divide), then solves the panel rows below by forward
//            substitution, one row per lane with the row in 32 registers;
//   update : trailing SYRK on the matrix cores, one wavefront per 32x32 tile =
//            2x2 v_mfma_f64_16x16x4_f64 accumulators x 8 k-steps.
// The right-hand side is stored as row n of the (n+1) x n array, so the factorisation performs the
// forward substitution on the way; the backward substitution stages each diagonal block in LDS.
// Only the lower triangle (row-major, ld = n) is read or written.
#include "common.hpp"

namespace vgg {

constexpr int kNB = 32;
constexpr int kLD = kNB + 1;
typedef double f64x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double fast_rcp(double x) {      // 1/x: hardware estimate + 2 Newton steps
  double r = __builtin_amdgcn_rcp(x);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-x, r, 1.0), r, r);
  return r;
}

// The 256 threads of a workgroup factor the 32x32 block D (LDS, ld = kLD, rows >= nb padded with the
// identity) in place.  Outer-product form WITHOUT normalising inside the loop (a_ic -= a_ij a_cj / d_j), so
// a step needs a single barrier: the column it reads was finalised by the previous step, the elements it
// writes are disjoint from it.  Columns are scaled by 1/sqrt(d_j) at the end; rdiag[j] = 1 / L_jj.
__device__ __forceinline__ void factor_diag_lds(double* D, double* rdiag, int32_t* fail_flag) {
  const int tid = threadIdx.x;
  const int c = tid & 31, i0 = tid >> 5;        // thread owns elements (i0 + 8 m, c), m = 0..3
  bool bad = false;
  for (int j = 0; j < kNB - 1; ++j) {
    __syncthreads();
    const double dj = D[j * kLD + j];
    if (!(dj > 0.0) || !(dj < 1.7976931348623157e308)) bad = true;
    const double inv = fast_rcp((dj > 0.0) ? dj : 1.0);
    if (c > j) {
      const double lcj = D[c * kLD + j] * inv;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int i = i0 + 8 * m;
        if (i >= c) D[i * kLD + c] -= D[i * kLD + j] * lcj;
      }
    }
  }
  __syncthreads();
  {
    const double dl = D[(kNB - 1) * kLD + kNB - 1];
    if (!(dl > 0.0) || !(dl < 1.7976931348623157e308)) bad = true;
  }
  // scale column c by 1/sqrt(d_c)
  const double dc = D[c * kLD + c];
  const double sd = sqrt((dc > 0.0) ? dc : 1.0);
  const double rs = 1.0 / sd;
  __syncthreads();
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int i = i0 + 8 * m;
    if (i > c) D[i * kLD + c] *= rs;
    else if (i == c) { D[i * kLD + c] = sd; rdiag[c] = rs; }
  }
  if (bad && tid == 0 && fail_flag) *fail_flag = 1;
  __syncthreads();
}

__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ A, int n, int nrows, int k0,
                                                         int32_t* fail, const int32_t* skip,
                                                         double* __restrict__ inv_blocks) {
  __shared__ double D[kNB * kLD];
  __shared__ double rdiag[kNB];
  if (skip && *skip) return;
  const int nb = min(kNB, n - k0);
  const int tid = threadIdx.x;
  for (int e = tid; e < kNB * kNB; e += 256) {
    const int i = e >> 5, c = e & 31;
    D[i * kLD + c] = (i < nb && c <= i) ? A[(size_t)(k0 + i) * n + k0 + c] : ((i == c) ? 1.0 : 0.0);
  }
  factor_diag_lds(D, rdiag, (blockIdx.x == 0) ? fail : nullptr);
  if (blockIdx.x == 0) {
    for (int e = tid; e < kNB * kNB; e += 256) {
      const int i = e >> 5, c = e & 31;
      if (i < nb && c <= i) A[(size_t)(k0 + i) * n + k0 + c] = D[i * kLD + c];
    }
  }
  // panel rows below the diagonal block (and the appended rhs row): x L_kk^T = a, one row per lane.
  // Column-oriented substitution: once x_k is final it is eliminated from all later columns with
  // independent FMAs, so the dependent chain is one multiply + one FMA per column (an fp64 FMA has a
  // 32-cycle dependent latency on gfx950) instead of a 528-long chain.
  if (inv_blocks && blockIdx.x == gridDim.x - 1) {
    // extra workgroup: T = L_kk^-T (rows of the identity pushed through the same substitution), used by the
    // backward solve as a plain 32x32 mat-vec instead of a 32-step dependent chain.  Row-major 32x32.
    if (tid < kNB) {
      double x[kNB];
#pragma unroll
      for (int c = 0; c < kNB; ++c) x[c] = (c == tid) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < kNB; ++k) {
        x[k] *= rdiag[k];
#pragma unroll
        for (int c = k + 1; c < kNB; ++c) x[c] -= x[k] * D[c * kLD + k];
      }
      double* T = inv_blocks + (size_t)(k0 / kNB) * kNB * kNB + tid * kNB;
#pragma unroll
      for (int c = 0; c < kNB; ++c) T[c] = x[c];
    }
    return;
  }
  const int row = k0 + nb + blockIdx.x * 256 + tid;
  if (row >= nrows) return;
  double* Arow = A + (size_t)row * n + k0;
  if (nb < kNB) {                               // ragged last block: only the appended rhs row sits below it
    for (int c = 0; c < nb; ++c) {
      double sacc = Arow[c];
      for (int k = 0; k < c; ++k) sacc -= Arow[k] * D[c * kLD + k];
      Arow[c] = sacc * rdiag[c];
    }
    return;
  }
  double x[kNB];
#pragma unroll
  for (int c = 0; c < kNB; ++c) x[c] = Arow[c];   // unconditional: all 32 loads in flight at once
#pragma unroll
  for (int k = 0; k < kNB; ++k) {
    x[k] *= rdiag[k];
#pragma unroll
    for (int c = k + 1; c < kNB; ++c) x[c] -= x[k] * D[c * kLD + k];
  }
#pragma unroll
  for (int c = 0; c < kNB; ++c) Arow[c] = x[c];
}

// trailing update A[i][j] -= sum_k L[i][k0+k] L[j][k0+k] for i >= j >= k0+32, tiles of 32x32
__global__ __launch_bounds__(256) void chol_update_kernel(double* __restrict__ A, int n, int nrows, int k0,
                                                          int num_tiles, const int32_t* skip) {
  if (skip && *skip) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;   // scalar: MFMAs
  const int t = blockIdx.x * 4 + wave;                                                       // behind scalar branches
  if (t >= num_tiles) return;
  int bi = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
  while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
  while (bi * (bi + 1) / 2 > t) --bi;
  const int bj = t - bi * (bi + 1) / 2;
  const int base = k0 + kNB;
  const int r0 = base + bi * 32, c0 = base + bj * 32;
  const int li = lane & 15, lk = lane >> 4;
  // operands: a[m][kk] = L[r0 + 16 m + li][k0 + 4 kk + lk], b[m][kk] = L[c0 + 16 m + li][k0 + 4 kk + lk]
  double a[2][8], b[2][8];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int ra = r0 + 16 * m + li, rb = c0 + 16 * m + li;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      a[m][kk] = (ra < nrows) ? A[(size_t)ra * n + k0 + 4 * kk + lk] : 0.0;
      b[m][kk] = (rb < nrows) ? A[(size_t)rb * n + k0 + 4 * kk + lk] : 0.0;
    }
  }
  f64x4 acc[2][2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q) acc[m][q] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < 8; ++kk)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        acc[m][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m][kk], b[q][kk], acc[m][q], 0, 0, 0);
  // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int i = r0 + 16 * m + lk + 4 * reg, j = c0 + 16 * q + li;
        if (i < nrows && j < n && j <= i) A[(size_t)i * n + j] -= acc[m][q][reg];
      }
}

// Substitutions, one workgroup.  Each 32x32 diagonal block is staged in LDS (with the reciprocals of its
// diagonal) so that the 32 dependent steps never wait on L2/HBM.  FORWARD (L z = b) is only needed when b
// is not stored as row n of the factored matrix (cholesky_solve_enqueue).
template <bool FORWARD>
__device__ __forceinline__ void tri_solve_block(const double* __restrict__ L, double* __restrict__ b, int n, int k0,
                                                double* Dl, double* rd, double* z) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int nb = min(kNB, n - k0);
  for (int e = tid; e < kNB * kNB; e += 256) {
    const int i = e >> 5, c = e & 31;
    if (i < nb && c <= i) {
      const double v = L[(size_t)(k0 + i) * n + k0 + c];
      Dl[i * kLD + c] = v;
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
        if (lane > k && lane < nb) v -= Dl[lane * kLD + k] * zk;
      }
    } else {
      for (int k = nb - 1; k >= 0; --k) {
        const double yk = __shfl(v, k, 64) * rd[k];
        if (lane == k) v = yk;
        if (lane < k) v -= Dl[k * kLD + lane] * yk;
      }
    }
    if (lane < nb) { z[lane] = v; b[k0 + lane] = v; }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void chol_solve_kernel(const double* __restrict__ L, double* __restrict__ b, int n,
                                                         int do_forward, int do_backward, const int32_t* skip) {
  __shared__ double Dl[kNB * kLD];
  __shared__ double rd[kNB];
  __shared__ double z[kNB];
  if (skip && *skip) return;
  const int tid = threadIdx.x;
  const int nblk = (n + kNB - 1) / kNB;
  if (do_forward) {
    for (int blk = 0; blk < nblk; ++blk) {
      const int k0 = blk * kNB, nb = min(kNB, n - k0);
      tri_solve_block<true>(L, b, n, k0, Dl, rd, z);
      const int sub = tid & 7, rloc = tid >> 3;       // 8 lanes per row, 4 consecutive columns each
      for (int i0 = k0 + nb; i0 < n; i0 += 32) {
        const int i = i0 + rloc;
        double s = 0.0;
        if (i < n) {
          const double* Li = L + (size_t)i * n + k0;
          for (int k = sub * 4; k < min(sub * 4 + 4, nb); ++k) s += Li[k] * z[k];
        }
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if (i < n && sub == 0) b[i] -= s;
      }
      __syncthreads();
    }
  }
  if (!do_backward) return;
  for (int blk = nblk - 1; blk >= 0; --blk) {
    const int k0 = blk * kNB, nb = min(kNB, n - k0);
    tri_solve_block<false>(L, b, n, k0, Dl, rd, z);
    // earlier rows: b_i -= sum_k L[k0+k][i] y_k.  A single workgroup is latency bound on these reads, so
    // every thread keeps all 32 loads of a column in flight and uses four partial sums.
    if (nb == kNB) {
      for (int i = tid; i < k0; i += 256) {
        double l[kNB];
#pragma unroll
        for (int k = 0; k < kNB; ++k) l[k] = L[(size_t)(k0 + k) * n + i];
        double s0 = b[i], s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int k = 0; k < kNB; k += 4) { s0 -= l[k] * z[k]; s1 -= l[k + 1] * z[k + 1]; s2 -= l[k + 2] * z[k + 2]; s3 -= l[k + 3] * z[k + 3]; }
        b[i] = (s0 + s1) + (s2 + s3);
      }
    } else {
      for (int i = tid; i < k0; i += 256) {
        double s0 = b[i];
        for (int k = 0; k < nb; ++k) s0 -= L[(size_t)(k0 + k) * n + i] * z[k];
        b[i] = s0;
      }
    }
    __syncthreads();
  }
}

// Backward substitution L^T x = y with the inverted diagonal blocks: one workgroup of 1024 threads, y in LDS.
// Per block row (last to first): x_j = T_j y_j (T_j = L_jj^-T, 32x32 mat-vec by 32 lanes), then
// y_i -= sum_k L[k0+k][i] x_k for all i < k0, one column per thread.  The 32 loads of a column and the next
// T block are issued BEFORE the mat-vec, so their latency overlaps it; nothing in the loop waits on a
// dependent global load.
constexpr int kBackThreads = 1024;
__global__ __launch_bounds__(kBackThreads) void chol_backward_kernel(const double* __restrict__ L, double* __restrict__ b,
                                                                     int n, const double* __restrict__ inv_blocks,
                                                                     const int32_t* skip) {
  extern __shared__ double sh[];
  if (skip && *skip) return;
  const int tid = threadIdx.x;
  const int nblk = (n + kNB - 1) / kNB;
  double* y = sh;                                   // [nblk * 32]
  double* T = y + nblk * kNB;                       // [32][33]
  double* xj = T + kNB * kLD;                       // [32]
  for (int i = tid; i < nblk * kNB; i += kBackThreads) y[i] = (i < n) ? b[i] : 0.0;
  {
    const double t = inv_blocks[(size_t)(nblk - 1) * kNB * kNB + tid];
    T[(tid >> 5) * kLD + (tid & 31)] = t;
  }
  __syncthreads();
  for (int blk = nblk - 1; blk >= 0; --blk) {
    const int k0 = blk * kNB, nb = min(kNB, n - k0);
    // issue the loads that do not depend on x_j
    const double tnext = (blk > 0) ? inv_blocks[(size_t)(blk - 1) * kNB * kNB + tid] : 0.0;
    double l[kNB];
    const bool have = (tid < k0);
    if (nb == kNB) {
#pragma unroll
      for (int k = 0; k < kNB; ++k) l[k] = have ? L[(size_t)(k0 + k) * n + tid] : 0.0;
    }
    if (tid < kNB) {
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int c = 0; c < kNB; c += 4) {
        s0 += T[tid * kLD + c] * y[k0 + c];
        s1 += T[tid * kLD + c + 1] * y[k0 + c + 1];
        s2 += T[tid * kLD + c + 2] * y[k0 + c + 2];
        s3 += T[tid * kLD + c + 3] * y[k0 + c + 3];
      }
      xj[tid] = (tid < nb) ? (s0 + s1) + (s2 + s3) : 0.0;
    }
    __syncthreads();
    if (tid < kNB) y[k0 + tid] = xj[tid];
    if (nb == kNB) {
      if (have) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int k = 0; k < kNB; k += 4) { s0 += l[k] * xj[k]; s1 += l[k + 1] * xj[k + 1]; s2 += l[k + 2] * xj[k + 2]; s3 += l[k + 3] * xj[k + 3]; }
        y[tid] -= (s0 + s1) + (s2 + s3);
      }
      for (int i = tid + kBackThreads; i < k0; i += kBackThreads) {     // columns beyond the first 1024
#pragma unroll
        for (int k = 0; k < kNB; ++k) l[k] = L[(size_t)(k0 + k) * n + i];
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int k = 0; k < kNB; k += 4) { s0 += l[k] * xj[k]; s1 += l[k + 1] * xj[k + 1]; s2 += l[k + 2] * xj[k + 2]; s3 += l[k + 3] * xj[k + 3]; }
        y[i] -= (s0 + s1) + (s2 + s3);
      }
    } else {                                        // ragged last block (processed first)
      for (int i = tid; i < k0; i += kBackThreads) {
        double s0 = 0.0;
        for (int k = 0; k < nb; ++k) s0 += L[(size_t)(k0 + k) * n + i] * xj[k];
        y[i] -= s0;
      }
    }
    __syncthreads();                                // all reads of T / xj are done
    T[(tid >> 5) * kLD + (tid & 31)] = tnext;
    __syncthreads();
  }
  for (int i = tid; i < n; i += kBackThreads) b[i] = y[i];
}

// workspace = the inverted diagonal blocks T_j = L_jj^-T, 32x32 doubles each
size_t cholesky_workspace_bytes(int n) { return (size_t)div_up(n, kNB) * kNB * kNB * sizeof(double) + 256; }

// If b is stored directly behind A (b == A + n*n, i.e. "row n" of an (n+1) x n matrix) the right-hand side
// rides through the factorisation as one more panel row: the panel solve and the trailing update then
// perform the forward substitution for free and only L^T y = z is left.
int cholesky_solve_enqueue(double* A, double* b, int n, double* inv_blocks, int32_t* device_fail, const int32_t* skip,
                           hipStream_t st) {
  const bool fused_rhs = (b == A + (size_t)n * n);
  const int nrows = fused_rhs ? n + 1 : n;
  // fast backward solve: y and one T block in LDS (64 KB dynamic LDS => n <= ~7000); larger systems use the
  // single-workgroup substitution kernel
  const size_t back_lds = ((size_t)div_up(n, kNB) * kNB + kNB * kLD + kNB) * sizeof(double);
  const bool use_inv = (inv_blocks != nullptr) && back_lds <= 64 * 1024;
  for (int k0 = 0; k0 < n; k0 += kNB) {
    const int nb = (n - k0 < kNB) ? n - k0 : kNB;
    const int rows_panel = nrows - k0 - nb;
    const int grid = (rows_panel > 0 ? div_up(rows_panel, 256) : 1) + (use_inv ? 1 : 0);
    chol_panel_kernel<<<grid, 256, 0, st>>>(A, n, nrows, k0, device_fail, skip, use_inv ? inv_blocks : nullptr);
    const int rows_below = nrows - k0 - kNB;
    if (rows_below > 0 && k0 + kNB < n) {
      const int T = div_up(rows_below, 32);
      const int tiles = T * (T + 1) / 2;
      chol_update_kernel<<<div_up(tiles, 4), 256, 0, st>>>(A, n, nrows, k0, tiles, skip);
    }
  }
  if (use_inv) {
    if (!fused_rhs) chol_solve_kernel<<<1, 256, 0, st>>>(A, b, n, 1, 0, skip);
    chol_backward_kernel<<<1, kBackThreads, back_lds, st>>>(A, b, n, inv_blocks, skip);
  } else {
    chol_solve_kernel<<<1, 256, 0, st>>>(A, b, n, fused_rhs ? 0 : 1, 1, skip);
  }
  if (hipGetLastError() != hipSuccess) return VGG_ERR_HIP;
  return VGG_OK;
}

}  // namespace vgg

extern "C" {
size_t vgg_cholesky_workspace_bytes(int n) { return n > 0 ? vgg::cholesky_workspace_bytes(n) : 0; }

int vgg_cholesky_solve(double* A, double* b, int n, void* workspace, int32_t* device_fail, void* stream) {
  if (n <= 0 || !A || !b || !workspace) return VGG_ERR_INVALID_ARGUMENT;
  return vgg::cholesky_solve_enqueue(A, b, n, (double*)workspace, device_fail, nullptr, (hipStream_t)stream);
}
}
