// Where do the bytes of a global_load_lds_dwordx4 land?  (round-6 LDS-DMA experiment: csrc/ba.hip, schur_tile_dma_kernel)
// Every lane reads the two doubles (1000 lane + 0, 1000 lane + 1) -- with lanes XOR 8 swapped on the source side in the second
// test, and only lanes < 16 active in the third -- into one LDS buffer; the host prints which double sits at which LDS index.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* __restrict__ src, double* __restrict__ out, int mode) {
  __shared__ __attribute__((aligned(16))) double buf[3 * 128];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 3 * 128; i += 64) buf[i] = -1.0;
  __syncthreads();
  const int sl = (mode == 1) ? (lane ^ 8) : lane;
  const char* g = reinterpret_cast<const char*>(src) + sl * 16;
  if (mode < 2 || lane < 16)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)(buf + 128), 16, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * 128; i += 64) out[i] = buf[i];
}
int main() {
  std::vector<double> h(128), o(384);
  for (int l = 0; l < 64; ++l) { h[2 * l] = 1000.0 * l; h[2 * l + 1] = 1000.0 * l + 1; }
  double *d, *dout; hipMalloc(&d, 128 * 8); hipMalloc(&dout, 384 * 8);
  hipMemcpy(d, h.data(), 128 * 8, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 3; ++mode) {
    k<<<1, 64>>>(d, dout, mode);
    hipMemcpy(o.data(), dout, 384 * 8, hipMemcpyDeviceToHost);
    int bad = 0, outside = 0;
    for (int i = 0; i < 384; ++i) {
      const bool in = i >= 128 && i < 256;
      if (!in) { outside += o[i] != -1.0; continue; }
      const int l = (i - 128) / 2, sl = (mode == 1) ? (l ^ 8) : l;
      const double want = (mode == 2 && l >= 16) ? -1.0 : 1000.0 * sl + (i & 1);
      bad += o[i] != want;
    }
    printf("mode %d: %d of 128 doubles not where lane * 16 bytes puts them, %d written outside; first: %.0f %.0f %.0f %.0f | lane 8: %.0f %.0f\n", mode, bad, outside,
           o[128], o[129], o[130], o[131], o[128 + 16], o[128 + 17]);
  }
  return 0;
}
