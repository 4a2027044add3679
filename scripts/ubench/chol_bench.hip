// standalone timing harness for csrc/chol.hip (not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../../vggsfm_amd/csrc/chol.hip"
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1202;
  // optional block-diagonal leading part (columns of A, columns of B): chol_bench 1202 384 288
  const int split_a = argc > 3 ? atoi(argv[2]) : 0, split_b = argc > 3 ? atoi(argv[3]) : 0;
  const int reps = 20;
  std::vector<double> A((size_t)n * n + n), M((size_t)n * 8);
  srand(1);
  // SPD: banded-ish random + diagonal dominance
  for (size_t i = 0; i < A.size(); ++i) A[i] = 0;
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double v = (rand() / (double)RAND_MAX - 0.5) * 0.01; A[(size_t)i * n + j] = v; }
  for (int i = split_a; i < split_a + split_b && i < n; ++i) for (int j = 0; j < split_a; ++j) A[(size_t)i * n + j] = 0.0;
  for (int i = 0; i < n; ++i) A[(size_t)i * n + i] = 1.0 + n * 0.01;
  for (int i = 0; i < n; ++i) A[(size_t)n * n + i] = rand() / (double)RAND_MAX;
  double *dA, *d0, *ws; int* fail;
  hipMalloc(&dA, A.size() * 8); hipMalloc(&d0, A.size() * 8); const size_t wsb = vgg::cholesky_workspace_bytes(n); hipMalloc(&ws, wsb); hipMalloc(&fail, 4);
  hipMemcpy(d0, A.data(), A.size() * 8, hipMemcpyHostToDevice);
  hipMemset(fail, 0, 4);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; ++mode) {
    float tot = 0;
    for (int r = 0; r < reps + 2; ++r) {
      hipMemcpyAsync(dA, d0, A.size() * 8, hipMemcpyDeviceToDevice, st);
      if (mode == 1) {  // keep the GPU busy before: a big memset pair
        for (int k = 0; k < 4; ++k) hipMemsetAsync(d0 + 0, 0, 0, st);
      }
      hipEventRecord(e0, st);
      vgg::cholesky_solve_enqueue(dA, dA + (size_t)n * n, n, ws, fail, nullptr, st, nullptr, split_a, split_b, nullptr, false);
      hipEventRecord(e1, st);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (r >= 2) tot += ms;
    }
    int hf; hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
    printf("n=%d mode=%d: %.3f ms per solve (fail=%d)\n", n, mode, tot / reps, hf);
  }
#ifdef VGG_CHOL_TRACE
  {
    // one traced solve: per-tile stamps -> the critical path along the diagonal
    const int nbk = (n + 63) / 64, tiles = nbk * (nbk + 1) / 2 + nbk;
    unsigned long long* tr; hipMalloc(&tr, (size_t)tiles * 8 * 8); hipMemset(tr, 0, (size_t)tiles * 8 * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(vgg::g_chol_trace), &tr, sizeof(tr));
    hipMemcpyAsync(dA, d0, A.size() * 8, hipMemcpyDeviceToDevice, st);
    vgg::cholesky_solve_enqueue(dA, dA + (size_t)n * n, n, ws, fail, nullptr, st, nullptr, split_a, split_b, nullptr, false);
    hipStreamSynchronize(st);
    std::vector<unsigned long long> h((size_t)tiles * 8);
    hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long t0 = ~0ull;
    for (int t = 0; t < tiles; ++t) if (h[(size_t)t * 8] && h[(size_t)t * 8] < t0) t0 = h[(size_t)t * 8];
    auto us = [&](unsigned long long v) { return v ? (double)(v - t0) * 0.01 : -1.0; };
    int t = 0;
    printf("col: diag[start updates_done factored published] | sub-diagonal tile [start updates_done T_arrived product published] (us)\n");
    for (int c = 0; c < nbk; ++c) {
      const unsigned long long* d = &h[(size_t)t * 8];
      const unsigned long long* p = (c + 1 < nbk) ? &h[(size_t)(t + 1) * 8] : nullptr;
      printf("%2d: %7.2f %7.2f %7.2f %7.2f", c, us(d[0]), us(d[1]), us(d[2]), us(d[3]));
      if (p) printf(" | %7.2f %7.2f %7.2f %7.2f %7.2f", us(p[0]), us(p[1]), us(p[2]), us(p[4]), us(p[3]));
      printf("\n");
      t += nbk - c + 1;
    }
    unsigned long long tend = 0;
    for (size_t i = 0; i < h.size(); ++i) if (h[i] > tend) tend = h[i];
    printf("last stamp %.2f us\n", us(tend));
    tr = nullptr; hipMemcpyToSymbol(HIP_SYMBOL(vgg::g_chol_trace), &tr, sizeof(tr));
  }
#endif
  // residual check
  std::vector<double> x(n);
  hipMemcpy(x.data(), dA + (size_t)n * n, n * 8, hipMemcpyDeviceToHost);
  double maxr = 0;
  for (int i = 0; i < n; ++i) { double s = 0; for (int j = 0; j < n; ++j) s += (j <= i ? A[(size_t)i * n + j] : A[(size_t)j * n + i]) * x[j]; maxr = fmax(maxr, fabs(s - A[(size_t)n * n + i])); }
  printf("max residual %.3e\n", maxr);
  return 0;
}
