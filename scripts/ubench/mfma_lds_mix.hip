// micro-benchmark (not product code): FP64 MFMA rate when every group of MFMAs is fed by ds_read_b64 operand loads
// and separated by workgroup barriers, as in the Schur tile kernel (6 loads : 9 MFMAs, 3 groups per barrier).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int LOADS, bool BARRIER, bool WRITES>
__global__ __launch_bounds__(256, 4) void mix_kernel(double* out, int iters) {
  __shared__ double lds[2][2304];
  for (int i = threadIdx.x; i < 2 * 2304; i += 256) (&lds[0][0])[i] = 1e-3 * i;
  __syncthreads();
  f64x4 acc[9];
  for (int i = 0; i < 9; ++i) acc[i] = (f64x4){0, 0, 0, 0};
  const int lane = threadIdx.x & 63;
  const int base = (lane & 15) * 3 + (lane >> 4) * 288;
  double a[3] = {1.0, 2.0, 3.0}, b[3] = {1.5, 2.5, 3.5};
  for (int it = 0; it < iters; ++it) {
    const double* L = lds[it & 1];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) {
      if (LOADS >= 6) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { a[i] = L[base + 48 * i + ks * 96]; b[i] = L[1152 + base + 48 * i + ks * 96]; }
      }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[3 * i + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[3 * i + j], 0, 0, 0);
    }
    if (WRITES) {
      double2* dst = reinterpret_cast<double2*>(lds[(it & 1) ^ 1]);
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[threadIdx.x + 256 * i] = make_double2(a[0] + it, b[0]);
    }
    if (BARRIER) __syncthreads();
  }
  double s = 0;
  for (int i = 0; i < 9; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K> void run(const char* name, K kern, double* out) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  const int iters = 4000;
  kern<<<1024, 256>>>(out, iters); hipDeviceSynchronize();
  hipEventRecord(s); kern<<<1024, 256>>>(out, iters); hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  printf("%-44s %7.3f ms  %.1f TFLOP/s\n", name, ms, 1024.0 * 4 * iters * 27 * 2048 / ms / 1e9);
}
int main() {
  double* out; hipMalloc(&out, 8 << 20);
  run("MFMA only", mix_kernel<0, false, false>, out);
  run("MFMA + barrier", mix_kernel<0, true, false>, out);
  run("MFMA + 18 ds_read_b64", mix_kernel<6, false, false>, out);
  run("MFMA + 18 ds_read_b64 + barrier", mix_kernel<6, true, false>, out);
  run("MFMA + reads + 4 ds_write_b128 + barrier", mix_kernel<6, true, true>, out);
  return 0;
}
