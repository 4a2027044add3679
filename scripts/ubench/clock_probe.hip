// shader clock under a tiny grid vs a full grid (not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(long long* out, int iters) {
  long long c0 = clock64(), w0 = wall_clock64();
  double a = threadIdx.x * 1e-9 + 1.0, b = 1.0000001;
  for (int i = 0; i < iters; ++i) a = __builtin_fma(a, b, 1e-9);
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)(a * 1e3); }
}
int main() {
  long long* d; hipMalloc(&d, 64); long long h[3];
  int wrate; hipDeviceGetAttribute(&wrate, hipDeviceAttributeWallClockRate, 0);
  for (int grid : {1, 4, 256, 2048}) {
    for (int rep = 0; rep < 3; ++rep) {
      probe<<<grid, 256>>>(d, 200000);
      hipDeviceSynchronize();
      hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      double sec = (double)h[1] / (wrate * 1e3);
      printf("grid=%d: shader cycles %lld wall ticks %lld (wall rate %d kHz) -> %.0f MHz, %.2f cycles per dependent fma, %.3f ms\n", grid, h[0], h[1], wrate,
             h[0] / sec / 1e6, (double)h[0] / 200000, sec * 1e3);
    }
  }
  return 0;
}
