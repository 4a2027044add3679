// Calibrates rocprofv3 FETCH_SIZE on gfx950 for the access widths the BA kernels use:
// coalesced streaming reads of a 1 GiB buffer (4x the Infinity Cache) at 4, 8 and 16 bytes per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T>
__global__ __launch_bounds__(256) void stream_read(const T* __restrict__ src, size_t n, double* sink) {
  double acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    T v = src[i];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc += (double)w[k];
  }
  if (acc == 1.2345) *sink = acc;
}

struct alignas(16) V16 { uint32_t a, b, c, d; };

int main() {
  const size_t bytes = 1ull << 30;
  void* buf; double* sink;
  hipMalloc(&buf, bytes); hipMalloc(&sink, 8);
  hipMemset(buf, 1, bytes);
  for (int rep = 0; rep < 3; ++rep) {
    stream_read<uint32_t><<<4096, 256>>>((const uint32_t*)buf, bytes / 4, sink);
    stream_read<double><<<4096, 256>>>((const double*)buf, bytes / 8, sink);
    stream_read<V16><<<4096, 256>>>((const V16*)buf, bytes / 16, sink);
  }
  hipDeviceSynchronize();
  printf("read %zu bytes per launch, 3 kernels x 3 reps\n", bytes);
  return 0;
}
