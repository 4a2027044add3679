// Dense fp64 Cholesky solve of the reduced camera system (6C+K unknowns) on gfx950.
//
// In the reference this is the DENSE_SCHUR / SPARSE_SCHUR factorisation inside Ceres, reached through
// pycolmap.bundle_adjustment (vggsfm/utils/triangulation.py:213,1050,1142).  Here: right-looking blocked
// Cholesky, NB = 32.  Per block column two launches:
//   panel  : every workgroup re-factors the 32x32 diagonal block in LDS (cheaper than a third launch),
//            then solves 64 panel rows (one row per lane, 32 accumulators in registers);
//   update : trailing SYRK on the matrix cores, one wavefront per 32x32 tile =
//            2x2 v_mfma_f64_16x16x4_f64 accumulators x 8 k-steps.
// Only the lower triangle (row-major, ld = n) is read or written.
#include "common.hpp"

namespace vgg {

constexpr int kNB = 32;
typedef double f64x4 __attribute__((ext_vector_type(4)));

// factor the nb x nb diagonal block held in LDS (ld = kNB+1) with a 256-thread workgroup
__device__ __forceinline__ void factor_diag_lds(double* D, int nb, int* fail_flag) {
  const int tid = threadIdx.x;
  for (int j = 0; j < nb; ++j) {
    __syncthreads();
    const double d = D[j * (kNB + 1) + j];
    __syncthreads();
    if (!(d > 0.0) || !(d < 1.7976931348623157e308)) {
      if (tid == 0 && fail_flag) *fail_flag = 1;
      // keep going with a harmless pivot so every lane stays in lock step
    }
    const double sd = (d > 0.0) ? sqrt(d) : 1.0;
    if (tid == 0) D[j * (kNB + 1) + j] = sd;
    if (tid > j && tid < nb) D[tid * (kNB + 1) + j] /= sd;
    __syncthreads();
    // trailing update of the block: element (i,c), j < c <= i < nb
    for (int e = tid; e < nb * nb; e += blockDim.x) {
      const int i = e / nb, c = e - i * nb;
      if (c > j && c <= i) D[i * (kNB + 1) + c] -= D[i * (kNB + 1) + j] * D[c * (kNB + 1) + j];
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ A, int n, int k0, int32_t* fail, const int32_t* skip) {
  __shared__ double D[kNB * (kNB + 1)];
  if (skip && *skip) return;
  const int nb = min(kNB, n - k0);
  const int tid = threadIdx.x;
  for (int e = tid; e < nb * nb; e += 256) {
    const int i = e / nb, c = e - i * nb;
    D[i * (kNB + 1) + c] = (c <= i) ? A[(size_t)(k0 + i) * n + k0 + c] : 0.0;
  }
  factor_diag_lds(D, nb, (blockIdx.x == 0) ? fail : nullptr);
  if (blockIdx.x == 0) {
    for (int e = tid; e < nb * nb; e += 256) {
      const int i = e / nb, c = e - i * nb;
      if (c <= i) A[(size_t)(k0 + i) * n + k0 + c] = D[i * (kNB + 1) + c];
    }
  }
  // panel rows below the diagonal block: row = k0 + nb + blockIdx.x*256 + tid
  const int row = k0 + nb + blockIdx.x * 256 + tid;
  if (row >= n || nb < kNB) return;   // a ragged last block has no rows below it
  double x[kNB];
  double* Arow = A + (size_t)row * n + k0;
#pragma unroll
  for (int c = 0; c < kNB; ++c) x[c] = Arow[c];
#pragma unroll
  for (int c = 0; c < kNB; ++c) {
    double s = x[c];
#pragma unroll
    for (int k = 0; k < c; ++k) s -= x[k] * D[c * (kNB + 1) + k];
    x[c] = s / D[c * (kNB + 1) + c];
  }
#pragma unroll
  for (int c = 0; c < kNB; ++c) Arow[c] = x[c];
}

// trailing update A[i][j] -= sum_k L[i][k0+k] L[j][k0+k] for i >= j >= k0+32, tiles of 32x32
__global__ __launch_bounds__(256) void chol_update_kernel(double* __restrict__ A, int n, int k0, int num_tiles, const int32_t* skip) {
  if (skip && *skip) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + wave;
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
      a[m][kk] = (ra < n) ? A[(size_t)ra * n + k0 + 4 * kk + lk] : 0.0;
      b[m][kk] = (rb < n) ? A[(size_t)rb * n + k0 + 4 * kk + lk] : 0.0;
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
        if (i < n && j < n && j <= i) A[(size_t)i * n + j] -= acc[m][q][reg];
      }
}

// forward (L z = b) then backward (L^T y = z) substitution, one workgroup
__global__ __launch_bounds__(256) void chol_solve_kernel(const double* __restrict__ L, double* __restrict__ b, int n, const int32_t* skip) {
  __shared__ double z[kNB];
  if (skip && *skip) return;
  const int tid = threadIdx.x, lane = tid & 63;
  for (int k0 = 0; k0 < n; k0 += kNB) {
    const int nb = min(kNB, n - k0);
    if (tid < 64) {
      double v = (lane < nb) ? b[k0 + lane] : 0.0;
      for (int k = 0; k < nb; ++k) {
        const double piv = L[(size_t)(k0 + k) * n + k0 + k];
        const double zk = __shfl(v, k, 64) / piv;
        if (lane == k) v = zk;
        if (lane > k && lane < nb) v -= L[(size_t)(k0 + lane) * n + k0 + k] * zk;
      }
      if (lane < nb) { z[lane] = v; b[k0 + lane] = v; }
    }
    __syncthreads();
    for (int i = k0 + nb + tid; i < n; i += 256) {
      const double* Li = L + (size_t)i * n + k0;
      double s = b[i];
      for (int k = 0; k < nb; ++k) s -= Li[k] * z[k];
      b[i] = s;
    }
    __syncthreads();
  }
  const int nblk = (n + kNB - 1) / kNB;
  for (int blk = nblk - 1; blk >= 0; --blk) {
    const int k0 = blk * kNB;
    const int nb = min(kNB, n - k0);
    if (tid < 64) {
      double v = (lane < nb) ? b[k0 + lane] : 0.0;
      for (int k = nb - 1; k >= 0; --k) {
        const double piv = L[(size_t)(k0 + k) * n + k0 + k];
        const double yk = __shfl(v, k, 64) / piv;
        if (lane == k) v = yk;
        if (lane < k) v -= L[(size_t)(k0 + k) * n + k0 + lane] * yk;
      }
      if (lane < nb) { z[lane] = v; b[k0 + lane] = v; }
    }
    __syncthreads();
    for (int i = tid; i < k0; i += 256) {
      double s = b[i];
      for (int k = 0; k < nb; ++k) s -= L[(size_t)(k0 + k) * n + i] * z[k];
      b[i] = s;
    }
    __syncthreads();
  }
}

int cholesky_solve_enqueue(double* A, double* b, int n, int32_t* device_fail, const int32_t* skip, hipStream_t st) {
  for (int k0 = 0; k0 < n; k0 += kNB) {
    const int rows_below = n - k0 - kNB;
    const int grid = rows_below > 0 ? div_up(rows_below, 256) : 1;
    chol_panel_kernel<<<grid, 256, 0, st>>>(A, n, k0, device_fail, skip);
    if (rows_below > 0) {
      const int T = div_up(rows_below, 32);
      const int tiles = T * (T + 1) / 2;
      chol_update_kernel<<<div_up(tiles, 4), 256, 0, st>>>(A, n, k0, tiles, skip);
    }
  }
  chol_solve_kernel<<<1, 256, 0, st>>>(A, b, n, skip);
  if (hipGetLastError() != hipSuccess) return VGG_ERR_HIP;
  return VGG_OK;
}

}  // namespace vgg

extern "C" int vgg_cholesky_solve(double* A, double* b, int n, int32_t* device_fail, void* stream) {
  if (n <= 0 || !A || !b) return VGG_ERR_INVALID_ARGUMENT;
  return vgg::cholesky_solve_enqueue(A, b, n, device_fail, nullptr, (hipStream_t)stream);
}
