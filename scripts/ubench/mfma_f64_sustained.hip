// micro-benchmark (not product code): SUSTAINED rate of v_mfma_f64_16x16x4_f64 -- does the FP64 matrix peak hold
// for kernels that run for about a millisecond (the Schur tile kernel), or does the chip throttle?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ void mfma_kernel(double* out, int iters) {
  f64x4 acc[9];
  for (int i = 0; i < 9; ++i) acc[i] = (f64x4){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 9; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  double* out; hipMalloc(&out, 8 * 1024 * 1024);
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  const int blocks = 1024, wg = 256;                 // 4 workgroups per CU, 4 wavefronts per SIMD
  for (int iters : {200, 2000, 20000, 200000}) {     // ~0.1 ms ... ~100 ms per launch
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(s); mfma_kernel<<<blocks, wg>>>(out, iters); hipEventRecord(e); hipEventSynchronize(e);
      float ms; hipEventElapsedTime(&ms, s, e);
      const double n = (double)blocks * (wg / 64) * iters * 9;
      printf("iters %6d rep %d: %8.3f ms  %.1f TFLOP/s\n", iters, rep, ms, n * 2048 / ms / 1e9);
    }
  }
  // back-to-back short launches for 0.5 s (the bench's duty cycle), rate of the last ones
  for (int i = 0; i < 400; ++i) mfma_kernel<<<blocks, wg>>>(out, 2000);
  hipEventRecord(s);
  for (int i = 0; i < 100; ++i) mfma_kernel<<<blocks, wg>>>(out, 2000);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  printf("after 400 warm launches: %.1f TFLOP/s\n", 100.0 * blocks * 4 * 2000 * 9 * 2048 / ms / 1e9);
  return 0;
}
