// micro-benchmark (not product code): issue rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void mfma_kernel(double* out, int iters) {
  f64x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f64x4){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void fma_kernel(double* out, int iters) {
  double acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = i;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(a, acc[i], b);
  }
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F> float timeit(F f) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  f(); hipDeviceSynchronize();
  hipEventRecord(s); f(); hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e); return ms;
}
int main() {
  double* out; hipMalloc(&out, 8 * 1024 * 1024);
  const int iters = 2000;
  for (int wg : {256, 512, 1024}) {
    const int blocks = 256 * (1024 / 256);   // 4 blocks per CU
    float ms = timeit([&] { mfma_kernel<4><<<blocks, wg, 0, 0>>>(out, iters); });
    double n = (double)blocks * (wg / 64) * iters * 4;
    printf("mfma f64 16x16x4 NACC=4 wg=%d: %.3f ms  %.1f TFLOP/s  (%.1f cycles/MFMA/SIMD @2.4GHz if all SIMDs busy)\n", wg, ms,
           n * 2048 / ms / 1e9, ms * 1e-3 * 2.4e9 / (n / 1024));
    ms = timeit([&] { mfma_kernel<9><<<blocks, wg, 0, 0>>>(out, iters); });
    n = (double)blocks * (wg / 64) * iters * 9;
    printf("mfma f64 16x16x4 NACC=9 wg=%d: %.3f ms  %.1f TFLOP/s\n", wg, ms, n * 2048 / ms / 1e9);
    ms = timeit([&] { fma_kernel<<<blocks, wg, 0, 0>>>(out, iters); });
    n = (double)blocks * wg * iters * 16;
    printf("v_fma_f64 wg=%d: %.3f ms  %.1f TFLOP/s\n", wg, ms, n * 2 / ms / 1e9);
  }
  return 0;
}
