// Accuracy of the hardware reciprocal / reciprocal-square-root estimates on gfx950 (how many Newton steps the
// factorisation kernels need).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = 1e-6 * pow(1e12, (double)i / n) * (1.0 + 0.37 * (i % 97) / 97.0);
  double r0 = __builtin_amdgcn_rcp(x);
  double r1 = __builtin_fma(__builtin_fma(-x, r0, 1.0), r0, r0);
  double r2 = __builtin_fma(__builtin_fma(-x, r1, 1.0), r1, r1);
  double q0 = __builtin_amdgcn_rsq(x);
  out[4 * i] = fabs(r0 * x - 1.0); out[4 * i + 1] = fabs(r1 * x - 1.0); out[4 * i + 2] = fabs(r2 * x - 1.0);
  out[4 * i + 3] = fabs(q0 * q0 * x - 1.0);
}
int main() {
  const int n = 1 << 20; double* d; hipMalloc(&d, 32ull * n);
  k<<<n / 256, 256>>>(d, n); double* h = new double[4 * n]; hipMemcpy(h, d, 32ull * n, hipMemcpyDeviceToHost);
  double m[4] = {0, 0, 0, 0};
  for (int i = 0; i < n; ++i) for (int j = 0; j < 4; ++j) m[j] = fmax(m[j], h[4 * i + j]);
  printf("max |x*rcp-1|: raw %.3e, 1 NR %.3e, 2 NR %.3e ; rsq raw (x*q*q-1) %.3e\n", m[0], m[1], m[2], m[3]);
  return 0;
}
