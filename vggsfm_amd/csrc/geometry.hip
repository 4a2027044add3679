// Projection / reprojection filter / normalisation kernels (gfx950).
//
// Replaces the batched-tensor formulations of the reference
//   project_3D_points + img_from_cam      vggsfm/utils/triangulation_helpers.py:311-395
//   filter_all_points3D(_single_chunk)    vggsfm/utils/triangulation_helpers.py:133-307
//   cam_from_img + iterative_undistortion vggsfm/utils/triangulation_helpers.py:398-428,
//                                         vggsfm/utils/distortion.py:27-159
// which materialise (S,P,*) and (S*S,P) temporaries, by streaming kernels over the reference's own
// dense layout: tracks (S,P,2), one thread per track so that every frame row is read coalesced
// (64 consecutive tracks = 512 B per wave), camera parameters wave-uniform (scalar loads).
// These kernels are HBM-bound: algorithmic traffic = S*P*(8 B track [+1 B detail]) per pass.
// Compiled with -ffp-contract=off: the boolean outputs are compared bit-for-bit with the oracle.
#include "common.hpp"

namespace vgg {

constexpr double kDblMax = 1.7976931348623157e308;

// torch.nan_to_num(x, nan=0): nan -> 0, +-inf -> +-max
__device__ __forceinline__ double nan_to_num0(double x) {
  if (x != x) return 0.0;
  if (x > kDblMax) return kDblMax;
  if (x < -kDblMax) return -kDblMax;
  return x;
}

template <int KD>
__device__ __forceinline__ void distort(const double* ep, double u, double v, double& ou, double& ov) {
  // vggsfm/utils/distortion.py:102-159, same operation order
  if (KD == 0) { ou = u; ov = v; return; }
  const double u2 = u * u, v2 = v * v, r2 = u2 + v2;
  double du, dv;
  if (KD == 1) {
    const double radial = ep[0] * r2;
    du = u * radial; dv = v * radial;
  } else if (KD == 2) {
    const double radial = ep[0] * r2 + ep[1] * r2 * r2;
    du = u * radial; dv = v * radial;
  } else {
    const double uv = u * v;
    const double radial = ep[0] * r2 + ep[1] * r2 * r2;
    du = u * radial + 2 * ep[2] * uv + ep[3] * (r2 + 2 * u2);
    dv = v * radial + 2 * ep[3] * uv + ep[2] * (r2 + 2 * v2);
  }
  ou = u + du; ov = v + dv;
}

template <int KD>
__device__ __forceinline__ void project_one(const double* __restrict__ E, const double* __restrict__ K,
                                            const double* __restrict__ ep, double X, double Y, double Z, double& px,
                                            double& py, double& cx, double& cy, double& cz) {
  cx = E[0] * X + E[1] * Y + E[2] * Z + E[3];
  cy = E[4] * X + E[5] * Y + E[6] * Z + E[7];
  cz = E[8] * X + E[9] * Y + E[10] * Z + E[11];
  double u = cx / cz, v = cy / cz;
  distort<KD>(ep, u, v, u, v);
  // bmm(K, [u v 1]^T): keeps 0*inf = nan exactly like the reference
  px = nan_to_num0(K[0] * u + K[1] * v + K[2] * 1.0);
  py = nan_to_num0(K[3] * u + K[4] * v + K[5] * 1.0);
}

// ------------------------------------------------------------------ project_3D_points
template <int KD>
__global__ __launch_bounds__(256) void project_kernel(const double* __restrict__ pts, int P,
                                                      const double* __restrict__ ext, const double* __restrict__ K,
                                                      const double* __restrict__ extra, int S,
                                                      double* __restrict__ out_uv, double* __restrict__ out_cam) {
  const int s = blockIdx.y;
  const double* E = ext + 12 * s;
  const double* Ks = K ? K + 9 * s : nullptr;
  const double* ep = KD ? extra + KD * s : nullptr;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    const double X = pts[3 * p], Y = pts[3 * p + 1], Z = pts[3 * p + 2];
    double px = 0, py = 0, cx, cy, cz;
    if (Ks) {
      project_one<KD>(E, Ks, ep, X, Y, Z, px, py, cx, cy, cz);
    } else {
      cx = E[0] * X + E[1] * Y + E[2] * Z + E[3];
      cy = E[4] * X + E[5] * Y + E[6] * Z + E[7];
      cz = E[8] * X + E[9] * Y + E[10] * Z + E[11];
    }
    if (out_uv) {
      double2 o; o.x = px; o.y = py;
      reinterpret_cast<double2*>(out_uv)[(size_t)s * P + p] = o;
    }
    if (out_cam) {
      out_cam[((size_t)s * 3 + 0) * P + p] = cx;
      out_cam[((size_t)s * 3 + 1) * P + p] = cy;
      out_cam[((size_t)s * 3 + 2) * P + p] = cz;
    }
  }
}

// ------------------------------------------------------------------ filter_all_points3D
// One thread per point.  Pass 1 streams the S frame rows of the track tensor (coalesced), keeps the
// per-frame inlier bit in LDS ([word][thread], conflict free).  Pass 2 (check_triangle) searches an
// inlier pair subtending >= min_tri_angle, widest frame distance first (the reference takes "any").
constexpr int kFilterThreads = 256;

__device__ __forceinline__ double tri_angle_deg(double r1, double r2, double b) {
  // law of cosines, min(theta, pi - theta) in degrees -- triangulation_helpers.py:568-586
  double den = 2.0 * sqrt(r1 * r2);
  double nom = r1 + r2 - b;
  if (den <= 1e-12) { nom = 1.0; den = 1.0; }
  double c = nom / den;
  c = fmin(fmax(c, -1.0), 1.0);
  double th = fabs(acos(c));
  th = fmin(th, 3.141592653589793 - th);
  return th * (180.0 / 3.141592653589793);
}

__device__ __forceinline__ double sq_of_norm3(double a, double b, double c) {
  const double n = sqrt(a * a + b * b + c * c);   // reference: (x).norm(dim=-1) ** 2
  return n * n;
}

// kFilterLanes adjacent lanes share one point: lane q projects the frames s = q, q + L, q + 2L, ... (the per-frame
// projection is a long dependent fp64 chain and one thread per point left 1.5 wavefronts per SIMD at 100k
// points), the inlier bit-words are OR-combined with shuffles, every lane of the group then holds the whole
// mask and runs the (early-exit) pair search redundantly.
constexpr int kFilterLanes = 4;
constexpr int kFilterPoints = kFilterThreads / kFilterLanes;      // points per workgroup

template <int KD, typename TrackT>
__global__ __launch_bounds__(kFilterThreads) void filter_kernel(
    const double* __restrict__ pts, int P, const TrackT* __restrict__ tracks, const double* __restrict__ ext,
    const double* __restrict__ K, const double* __restrict__ extra, const double* __restrict__ centers, int S,
    double max_err_sq, double min_tri_angle, int check_triangle, double hard_max, double behind_value,
    uint8_t* __restrict__ out_mask, uint8_t* __restrict__ out_detail) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* bits = reinterpret_cast<unsigned long long*>(smem_raw);   // [NW][kFilterPoints]
  const int tid = threadIdx.x;
  const int lp = tid / kFilterLanes, q = tid % kFilterLanes;
  const int p = blockIdx.x * kFilterPoints + lp;
  const bool live = p < P;
  double X = 0, Y = 0, Z = 0;
  if (live) { X = pts[3 * p]; Y = pts[3 * p + 1]; Z = pts[3 * p + 2]; }
  int count = 0;
  const int nw = (S + 63) >> 6;
  for (int wd = 0; wd < nw; ++wd) {
    unsigned long long word = 0;
    const int s_end = min(S, 64 * (wd + 1));
    for (int s = 64 * wd + q; s < s_end; s += kFilterLanes) {
      bool inl = false;
      if (live) {
        double px, py, cx, cy, cz;
        project_one<KD>(ext + 12 * s, K + 9 * s, KD ? extra + KD * s : nullptr, X, Y, Z, px, py, cx, cy, cz);
        const double tx = (double)tracks[((size_t)s * P + p) * 2], ty = (double)tracks[((size_t)s * P + p) * 2 + 1];
        const double dx = px - tx, dy = py - ty;
        const double n = sqrt(dx * dx + dy * dy);
        double err = n * n;
        if (cz <= 0) err = behind_value;
        inl = err <= max_err_sq;
      }
      if (inl) word |= 1ull << (s & 63);
    }
#pragma unroll
    for (int off = 1; off < kFilterLanes; off <<= 1) word |= __shfl_xor(word, off, 64);
    count += __popcll(word);
    if (q == 0) bits[wd * kFilterPoints + lp] = word;
  }
  __syncthreads();
  bool valid = count >= 2;
  if (hard_max > 0) valid = valid && fabs(X) <= hard_max && fabs(Y) <= hard_max && fabs(Z) <= hard_max;
  bool tri_any = false;
  if (check_triangle && valid && live) {
    // frames a < b, both inliers ("exists a pair with angle >= threshold": the order of the search is free).
    // Start from the widest inlier pair -- first and last inlier frame -- and shrink: for a track seen in a
    // window of a long sequence this is O(1) instead of walking thousands of non-inlier (a, b) combinations.
    int lo = S, hi = -1;
    for (int wd = 0; wd < nw; ++wd) {
      const unsigned long long wbits = bits[wd * kFilterPoints + lp];
      if (wbits) {
        if (lo == S) lo = 64 * wd + __ffsll((long long)wbits) - 1;
        hi = 64 * wd + 63 - __clzll((long long)wbits);
      }
    }
    for (int dist = hi - lo; dist >= 1 && !tri_any; --dist) {
      for (int a = lo; a + dist <= hi; ++a) {
        const int b = a + dist;
        const bool ia = (bits[(a >> 6) * kFilterPoints + lp] >> (a & 63)) & 1ull;
        if (!ia) continue;
        const bool ib = (bits[(b >> 6) * kFilterPoints + lp] >> (b & 63)) & 1ull;
        if (!ib) continue;
        const double* ca = centers + 3 * a;
        const double* cb = centers + 3 * b;
        const double bsq = sq_of_norm3(ca[0] - cb[0], ca[1] - cb[1], ca[2] - cb[2]);
        const double r1 = sq_of_norm3(X - ca[0], Y - ca[1], Z - ca[2]);
        const double r2 = sq_of_norm3(X - cb[0], Y - cb[1], Z - cb[2]);
        if (tri_angle_deg(r1, r2, bsq) >= min_tri_angle) { tri_any = true; break; }
      }
    }
  }
  const bool ret = check_triangle ? (tri_any && valid) : valid;
  if (live && q == 0) out_mask[p] = ret ? 1 : 0;
  if (out_detail) {
    for (int s = q; s < S; s += kFilterLanes) {
      if (live) {
        bool d = (bits[(s >> 6) * kFilterPoints + lp] >> (s & 63)) & 1ull;
        if (check_triangle) d = d && tri_any;
        out_detail[(size_t)s * P + p] = d ? 1 : 0;
      }
    }
  }
}

// projection centres -R^T t (triangulation_helpers.py:531)
__global__ void centers_kernel(const double* __restrict__ ext, int S, double* __restrict__ centers) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  const double* E = ext + 12 * s;
  for (int j = 0; j < 3; ++j) centers[3 * s + j] = -(E[0 + j] * E[3] + E[4 + j] * E[7] + E[8 + j] * E[11]);
}

// ------------------------------------------------------------------ cam_from_img
template <typename TrackT>
__global__ __launch_bounds__(256) void normalize_kernel(const TrackT* __restrict__ tracks, const double* __restrict__ K,
                                                        int S, int P, double* __restrict__ out) {
  const int s = blockIdx.y;
  const double fx = K[9 * s + 0], fy = K[9 * s + 4], cx = K[9 * s + 2], cy = K[9 * s + 5];
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    const size_t i = ((size_t)s * P + p) * 2;
    double2 o;
    o.x = ((double)tracks[i] - cx) / fx;
    o.y = ((double)tracks[i + 1] - cy) / fy;
    reinterpret_cast<double2*>(out)[(size_t)s * P + p] = o;
  }
}

// One Newton iteration of iterative_undistortion (distortion.py:57-97) for every element, central
// differences with step max(|u|*rel, eps), 2x2 LU with partial pivoting (torch.linalg.solve).  The
// reference stops when the max squared step over the WHOLE tensor is < max_step_norm, so iteration
// `it` first looks at the global max published by iteration it-1.
template <int KD>
__global__ __launch_bounds__(256) void undistort_iter_kernel(const double* __restrict__ orig, double* __restrict__ cur,
                                                             const double* __restrict__ extra, int S, int P, int it,
                                                             double rel_step, double eps, double max_step_norm,
                                                             unsigned long long* __restrict__ max_bits) {
  if (it > 0) {
    const double prev = __longlong_as_double((long long)max_bits[it - 1]);
    if (prev < max_step_norm) return;
  }
  const int s = blockIdx.y;
  const double* ep = extra + KD * s;
  double local_max = 0.0;
  bool local_nan = false;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
    const size_t i = (size_t)s * P + p;
    const double2 o = reinterpret_cast<const double2*>(orig)[i];
    double2 c = reinterpret_cast<double2*>(cur)[i];
    double u = c.x, v = c.y;
    double ud, vd;
    distort<KD>(ep, u, v, ud, vd);
    const double dx = o.x - ud, dy = o.y - vd;
    const double su = fmax(fabs(u) * rel_step, eps), sv = fmax(fabs(v) * rel_step, eps);
    double pu0, pu1, mu0, mu1, pv0, pv1, mv0, mv1;
    distort<KD>(ep, u + su, v, pu0, pu1);
    distort<KD>(ep, u - su, v, mu0, mu1);
    distort<KD>(ep, u, v + sv, pv0, pv1);
    distort<KD>(ep, u, v - sv, mv0, mv1);
    // distortion.py:66-90 differences apply_distortion (which already returns u+du) and still adds 1 on the
    // diagonal -- reproduced literally, it only changes the Newton convergence rate.
    const double j00 = (pu0 - mu0) / (2 * su) + 1, j01 = (pv0 - mv0) / (2 * sv);
    const double j10 = (pu1 - mu1) / (2 * su), j11 = (pv1 - mv1) / (2 * sv) + 1;
    const bool swap = fabs(j10) > fabs(j00);
    const double a00 = swap ? j10 : j00, a01 = swap ? j11 : j01, a10 = swap ? j00 : j10, a11 = swap ? j01 : j11;
    const double b0 = swap ? dy : dx, b1 = swap ? dx : dy;
    const double l10 = a10 / a00;
    const double u11 = a11 - l10 * a01;
    const double y1 = b1 - l10 * b0;
    const double d1 = y1 / u11;
    const double d0 = (b0 - a01 * d1) / a00;
    c.x = u + d0; c.y = v + d1;
    reinterpret_cast<double2*>(cur)[i] = c;
    const double sq = d0 * d0 + d1 * d1;
    if (sq != sq) local_nan = true; else local_max = fmax(local_max, sq);
  }
  // torch.max propagates NaN; NaN bit pattern (0x7ff8...) is above every finite non-negative double
  // one same-address atomic per WORKGROUP (they serialise in L2: one per wavefront of a 78k-workgroup grid
  // cost 2.2 ms per iteration at 200 x 100k)
  __shared__ unsigned long long wg_bits[4];
  const double m = wave_max(local_max);
  const bool any_nan = __any(local_nan);
  if (lane_id() == 0)
    wg_bits[threadIdx.x >> 6] = any_nan ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long b = wg_bits[0];
    for (int i = 1; i < 4; ++i) b = wg_bits[i] > b ? wg_bits[i] : b;
    atomicMax(&max_bits[it], b);
  }
}

}  // namespace vgg

using namespace vgg;

extern "C" {

int vgg_project_points(const double* points3D, int P, const double* extrinsics, const double* intrinsics,
                       const double* extra_params, int num_extra, int S, double* out_uv, double* out_cam,
                       void* stream) {
  if (P < 0 || S < 0 || (num_extra != 0 && num_extra != 1 && num_extra != 2 && num_extra != 4))
    return VGG_ERR_INVALID_ARGUMENT;
  if (P == 0 || S == 0) return VGG_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(min(div_up(P, 256), 4096), S);
  const double* ep = num_extra ? extra_params : nullptr;
  switch (intrinsics ? num_extra : 0) {
    case 0: project_kernel<0><<<grid, 256, 0, st>>>(points3D, P, extrinsics, intrinsics, ep, S, out_uv, out_cam); break;
    case 1: project_kernel<1><<<grid, 256, 0, st>>>(points3D, P, extrinsics, intrinsics, ep, S, out_uv, out_cam); break;
    case 2: project_kernel<2><<<grid, 256, 0, st>>>(points3D, P, extrinsics, intrinsics, ep, S, out_uv, out_cam); break;
    default: project_kernel<4><<<grid, 256, 0, st>>>(points3D, P, extrinsics, intrinsics, ep, S, out_uv, out_cam); break;
  }
  VGG_LAUNCH_CHECK();
  return VGG_OK;
}

size_t vgg_filter_points_workspace_bytes(int S) { return (size_t)S * 3 * sizeof(double); }

int vgg_filter_points(const double* points3D, int P, const void* tracks, int tracks_are_f64, const double* extrinsics,
                      const double* intrinsics, const double* extra_params, int num_extra, int S,
                      double max_reproj_error, double min_tri_angle, int check_triangle, double hard_max,
                      double behind_value, uint8_t* out_mask, uint8_t* out_detail, void* workspace, void* stream) {
  if (P < 0 || S < 0 || (num_extra != 0 && num_extra != 1 && num_extra != 2 && num_extra != 4) || S > 4096)
    return VGG_ERR_INVALID_ARGUMENT;
  if (P == 0) return VGG_OK;
  hipStream_t st = (hipStream_t)stream;
  double* centers = (double*)workspace;
  if (check_triangle) {
    if (!workspace) return VGG_ERR_WORKSPACE;
    centers_kernel<<<div_up(S, 64), 64, 0, st>>>(extrinsics, S, centers);
  }
  const size_t lds = (size_t)((S + 63) / 64) * kFilterPoints * sizeof(unsigned long long);
  const int grid = div_up(P, kFilterPoints);
  const double thr = max_reproj_error * max_reproj_error;
#define VGG_FILTER(KD, T)                                                                                          \
  filter_kernel<KD, T><<<grid, kFilterThreads, lds, st>>>(points3D, P, (const T*)tracks, extrinsics, intrinsics,   \
                                                          extra_params, centers, S, thr, min_tri_angle,            \
                                                          check_triangle, hard_max, behind_value, out_mask, out_detail)
  if (tracks_are_f64) {
    switch (num_extra) { case 0: VGG_FILTER(0, double); break; case 1: VGG_FILTER(1, double); break;
                         case 2: VGG_FILTER(2, double); break; default: VGG_FILTER(4, double); }
  } else {
    switch (num_extra) { case 0: VGG_FILTER(0, float); break; case 1: VGG_FILTER(1, float); break;
                         case 2: VGG_FILTER(2, float); break; default: VGG_FILTER(4, float); }
  }
#undef VGG_FILTER
  VGG_LAUNCH_CHECK();
  return VGG_OK;
}

size_t vgg_cam_from_img_workspace_bytes(int S, int P, int max_iterations) {
  return (size_t)S * P * 2 * sizeof(double) + (size_t)(max_iterations + 1) * sizeof(unsigned long long);
}

// tracks (S,P,2) -> out (S,P,2) f64.  iterations_run (host int, may be NULL) forces one stream sync.
int vgg_cam_from_img(const void* tracks, int tracks_are_f64, const double* intrinsics, const double* extra_params,
                     int num_extra, int S, int P, double* out, int max_iterations, double max_step_norm,
                     double rel_step_size, double eps, void* workspace, int* iterations_run, void* stream) {
  if (P < 0 || S < 0 || (num_extra != 0 && num_extra != 1 && num_extra != 2 && num_extra != 4))
    return VGG_ERR_INVALID_ARGUMENT;
  if (iterations_run) *iterations_run = 0;
  if (P == 0 || S == 0) return VGG_OK;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(min(div_up(P, 256), 2048), S);
  // Newton iterations: ~4096 workgroups in total, grid-stride over the tracks of a frame
  const dim3 grid_it(min(div_up(P, 256), max(1, 4096 / S)), S);
  if (tracks_are_f64) normalize_kernel<double><<<grid, 256, 0, st>>>((const double*)tracks, intrinsics, S, P, out);
  else normalize_kernel<float><<<grid, 256, 0, st>>>((const float*)tracks, intrinsics, S, P, out);
  VGG_LAUNCH_CHECK();
  if (num_extra == 0) return VGG_OK;
  if (!workspace) return VGG_ERR_WORKSPACE;
  double* orig = (double*)workspace;
  unsigned long long* max_bits = (unsigned long long*)(orig + (size_t)S * P * 2);
  VGG_HIP_CHECK(hipMemcpyAsync(orig, out, (size_t)S * P * 2 * sizeof(double), hipMemcpyDeviceToDevice, st));
  VGG_HIP_CHECK(hipMemsetAsync(max_bits, 0, (size_t)(max_iterations + 1) * sizeof(unsigned long long), st));
  // The global stopping rule needs the previous iteration's max: iterations are enqueued in batches and
  // the (tiny) max array is read back between batches; kernels after convergence return immediately.
  const int batch = 8;
  int done_it = max_iterations;
  unsigned long long host_bits[8];
  for (int it0 = 0; it0 < max_iterations; it0 += batch) {
    const int n = (it0 + batch <= max_iterations) ? batch : max_iterations - it0;
    for (int it = it0; it < it0 + n; ++it) {
      switch (num_extra) {
        case 1: undistort_iter_kernel<1><<<grid_it, 256, 0, st>>>(orig, out, extra_params, S, P, it, rel_step_size, eps, max_step_norm, max_bits); break;
        case 2: undistort_iter_kernel<2><<<grid_it, 256, 0, st>>>(orig, out, extra_params, S, P, it, rel_step_size, eps, max_step_norm, max_bits); break;
        default: undistort_iter_kernel<4><<<grid_it, 256, 0, st>>>(orig, out, extra_params, S, P, it, rel_step_size, eps, max_step_norm, max_bits); break;
      }
    }
    VGG_LAUNCH_CHECK();
    VGG_HIP_CHECK(hipMemcpyAsync(host_bits, max_bits + it0, n * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    VGG_HIP_CHECK(hipStreamSynchronize(st));
    bool stop = false;
    for (int i = 0; i < n; ++i) {
      double m;
      memcpy(&m, &host_bits[i], sizeof(double));
      if (m < max_step_norm) { done_it = it0 + i + 1; stop = true; break; }
    }
    if (stop) break;
  }
  if (iterations_run) *iterations_run = done_it;
  return VGG_OK;
}

}  // extern "C"
