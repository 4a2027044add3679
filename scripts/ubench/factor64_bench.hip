// factor64_blocked (csrc/chol.hip) alone: one workgroup factors the same 64 x 64 block ITERS times out of LDS.
// -DNOFACTOR: load / copy only (the harness overhead, 1.8 us).  The ablation switches of the first, LDS-based 16 x 16
// routine (-DVGG_F16_ABL=<mask>: no fences / no identity rows / no update / no pivot chain / empty loop) went with it:
// commit 9f5c616 has both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../../vggsfm_amd/csrc/chol.hip"
__global__ __launch_bounds__(256) void f64_loop(const double* __restrict__ A0, double* __restrict__ out, int iters, int32_t* fail) {
  extern __shared__ double smem[];
  vgg::DfShared& sh = *reinterpret_cast<vgg::DfShared*>(smem);
  constexpr int LD = vgg::DFB + 1;
  __builtin_amdgcn_s_setprio(3);
  for (int it = 0; it < iters; ++it) {
    for (int e = threadIdx.x; e < 64 * 64; e += 256) sh.D[(e / 64) * LD + e % 64] = A0[e];
    __syncthreads();
#ifndef NOFACTOR
    vgg::factor64_blocked(sh.D, sh.T, sh.scr, fail, [](int) {});
#endif
    __syncthreads();
  }
  for (int e = threadIdx.x; e < 64 * 64; e += 256) { out[e] = sh.D[(e / 64) * LD + e % 64]; out[4096 + e] = sh.T[(e / 64) * LD + e % 64]; }
}
int main(int argc, char** argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 1, iters = 200;
  std::vector<double> A(4096), O(8192);
  srand(3);
  for (int i = 0; i < 64; ++i) for (int j = 0; j <= i; ++j) { double v = (rand() / (double)RAND_MAX - 0.5) * 0.1; A[i * 64 + j] = A[j * 64 + i] = v; }
  for (int i = 0; i < 64; ++i) A[i * 64 + i] = 2.0;
  double *dA, *dO; int32_t* fail;
  hipMalloc(&dA, 4096 * 8); hipMalloc(&dO, 8192 * 8 * wgs); hipMalloc(&fail, 4); hipMemset(fail, 0, 4);
  hipMemcpy(dA, A.data(), 4096 * 8, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)f64_loop, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(vgg::DfShared));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    f64_loop<<<wgs, 256, sizeof(vgg::DfShared)>>>(dA, dO, iters, fail);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("wgs=%d: %.3f us per 64 x 64 block\n", wgs, ms * 1e3 / iters);
  }
  hipMemcpy(O.data(), dO, 8192 * 8, hipMemcpyDeviceToHost);
  // residual |L L^T - A| and |L^T T^T ... | : T = L^-T  =>  L^T T = I
  double r1 = 0, r2 = 0;
  for (int i = 0; i < 64; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = 0; k <= j; ++k) s += O[i * 64 + k] * O[j * 64 + k]; r1 = fmax(r1, fabs(s - A[i * 64 + j])); }
  for (int i = 0; i < 64; ++i) for (int j = 0; j < 64; ++j) { double s = 0; for (int k = 0; k < 64; ++k) s += (k >= i ? O[k * 64 + i] : 0.0) * (j >= k ? O[4096 + k * 64 + j] : 0.0); r2 = fmax(r2, fabs(s - (i == j))); }
  int hf; hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
  printf("max |L L^T - A| %.2e   max |L^T T - I| %.2e  fail=%d\n", r1, r2, hf);
  return 0;
}
