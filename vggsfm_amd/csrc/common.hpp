// Shared device/host helpers for the gfx950 (CDNA4, wave64) kernels of the geometry hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#define VGG_OK 0
#define VGG_ERR_INVALID_ARGUMENT (-1)
#define VGG_ERR_HIP (-2)
#define VGG_ERR_WORKSPACE (-3)
#define VGG_ERR_UNSUPPORTED (-4)

#define VGG_HIP_CHECK(expr)                                    \
  do {                                                         \
    hipError_t e_ = (expr);                                    \
    if (e_ != hipSuccess) return VGG_ERR_HIP;                  \
  } while (0)

#define VGG_LAUNCH_CHECK()                                     \
  do {                                                         \
    if (hipGetLastError() != hipSuccess) return VGG_ERR_HIP;   \
  } while (0)

namespace vgg {

constexpr int kWave = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// full-wave (64 lanes) butterfly reductions; every lane ends with the total
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// sum over the aligned group of W lanes (W = 16, 32, 64) this lane belongs to; every lane of the group gets the total
template <int W>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
  for (int off = W / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ double wave_bcast(double v, int src) { return __shfl(v, src, 64); }

template <typename T>
__device__ __forceinline__ double load_f64(const T* p) { return (double)(*p); }

inline int div_up(long a, long b) { return (int)((a + b - 1) / b); }

// Overlap of the factorisation with the computation of the reduced system (ba.hip <-> chol.hip): contributions to
// columns >= first_col arrive in a second matrix S2 (same n x n layout) from work running on another stream; event k
// signals that every contribution to columns >= wait_col[k] (ascending) is in place.  The factorisation waits for
// event k before the first panel that touches such a column and adds S2 to the panel when it loads it.
struct CholOverlap {
  const double* S2;
  int first_col;
  int num_waits;
  int wait_col[8];
  hipEvent_t wait_ev[8];
  const int32_t* dev_flags = nullptr;   // device [num_waits]: flag k != 0 <=> event k has happened (dataflow factorisation:
                                        // one launch, so it waits on the device; set by a kernel behind each tile batch)
};
// by-value kernel argument of the dataflow factorisation
struct DfOverlap {
  const double* S2;
  const int32_t* flags;
  int first_col, num_waits;
  int wait_col[8];
};

}  // namespace vgg
