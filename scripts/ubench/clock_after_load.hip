// Shader clock seen by a one-workgroup latency-bound loop that runs right behind a chip-wide fp64 MFMA burst on the same
// stream -- the situation of the factorisation behind the Schur tile launches (not product code).
//   clock_after_load [burst_ms ~ 1] [probe_iters 40000 ~ 0.5 ms]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void burst(double* out, int iters) {
  f64x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
  const double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;
}
__global__ void probe(long long* out, int iters, int slot) {
  long long c0 = clock64(), w0 = wall_clock64();
  double a = threadIdx.x * 1e-9 + 1.0, b = 1.0000001;
  for (int i = 0; i < iters; ++i) a = __builtin_fma(a, b, 1e-9);
  long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[3 * slot] = c1 - c0; out[3 * slot + 1] = w1 - w0; out[3 * slot + 2] = (long long)(a * 1e3); }
}
int main(int argc, char** argv) {
  const double burst_ms = argc > 1 ? atof(argv[1]) : 1.0;
  const int probe_iters = argc > 2 ? atoi(argv[2]) : 40000;
  long long* d; hipMalloc(&d, 3 * 8 * 64); double* o; hipMalloc(&o, 64);
  int wrate; hipDeviceGetAttribute(&wrate, hipDeviceAttributeWallClockRate, 0);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // calibrate the burst length
  int iters = 2000;
  burst<<<1024, 256, 0, st>>>(o, iters); hipStreamSynchronize(st);
  hipEventRecord(e0, st); burst<<<1024, 256, 0, st>>>(o, iters); hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  iters = (int)(iters * burst_ms / ms);
  printf("burst: %d iterations of 8 MFMA per wavefront, 1024 workgroups ~ %.2f ms\n", iters, burst_ms);
  long long h[3 * 8];
  for (int mode = 0; mode < 2; ++mode) {            // 0: probe alone; 1: 30 x (burst, probe) back to back, as in an LM loop
    const int reps = mode ? 30 : 3;
    double mhz_min = 1e9, mhz_max = 0, mhz_sum = 0;
    for (int r = 0; r < reps; ++r) {
      if (mode) burst<<<1024, 256, 0, st>>>(o, iters);
      probe<<<1, 256, 0, st>>>(d, probe_iters, 0);
      if (!mode || r == reps - 1 || true) {
        hipStreamSynchronize(st);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        const double sec = (double)h[1] / (wrate * 1e3), mhz = h[0] / sec / 1e6;
        if (r >= (mode ? 5 : 0)) { mhz_min = mhz < mhz_min ? mhz : mhz_min; mhz_max = mhz > mhz_max ? mhz : mhz_max; mhz_sum += mhz; }
      }
    }
    const int cnt = reps - (mode ? 5 : 0);
    printf("%s: shader clock of the probe %.0f MHz mean (min %.0f, max %.0f)\n", mode ? "behind a burst" : "alone", mhz_sum / cnt, mhz_min, mhz_max);
  }
  // the same without host synchronisation between the pairs (a queue of 30 pairs, every probe in its own slot)
  for (int r = 0; r < 8; ++r) { burst<<<1024, 256, 0, st>>>(o, iters); probe<<<1, 256, 0, st>>>(d, probe_iters, r); }
  hipStreamSynchronize(st);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("queued pairs:");
  for (int r = 0; r < 8; ++r) { const double sec = (double)h[3 * r + 1] / (wrate * 1e3); printf(" %.0f", h[3 * r] / sec / 1e6); }
  printf(" MHz\n");
  return 0;
}
