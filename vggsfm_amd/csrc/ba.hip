// Levenberg-Marquardt bundle adjustment with a block-sparse Schur complement, gfx950 (CDNA4).
//
// Replaces pycolmap.bundle_adjustment / pycolmap.pose_refinement as the reference uses them
// (vggsfm/utils/triangulation.py:213,387,590,1050,1142; vggsfm/runners/video_runner.py:508,831,1001):
// COLMAP's BundleAdjuster on Ceres' trust-region LM.  The Ceres control flow that decides the
// trajectory (Jacobi scaling frozen at iteration 0, D^2 = clamp(diag J^T J)/radius, step quality,
// radius update, invalid steps) is restated on the device so that the host never synchronises inside
// the loop.  SURVEY.md Appendix A is the working spec.
//
// Data layout (all resident in HBM for the whole solve, float64 unless noted):
//   cameras   cam_q[C,4] cam_t[C,3] intr[NI,4]        replicated on every GPU
//   points    pts[P,3]                                 sharded by point index across GPUs
//   observations, twice:  point-major CSR (row_ptr, obs_cam i32, obs_uv f32x2)
//                         camera-major CSR (col_ptr, cobs_pt i32, cobs_uv f32x2)
//   reduced camera system S[n,n] (lower, row-major), n = 6C + kd*NI
// Kernels per LM iteration (algorithmic bytes in DESIGN.md):
//   cam_pass<LIN>   camera-major: U_c = F^T F, g_c = F^T r, cost            (only after an accepted step)
//   point_pass      point-major, one wavefront per point: V_p = E^T E + D^2, its Cholesky inverse,
//                   h_p = V^-1 E^T r, shared-intrinsics coupling
//   cam_pass<RHS>   camera-major: F^T (r - E h_p) and the shared-intrinsics Schur terms
//   schur_tile      S -= sum_p Y_p Y_p^T on 16x16-camera tiles; thread (a,b) owns the 6x6 (7x7, 8x8)
//                   block of cameras (I*16+a, J*16+b) in registers, Y blocks staged through LDS
//   assemble        diagonal blocks, damping, constant columns
//   cholesky        chol.hip (matrix-core trailing update)
//   point_step      back-substitution, model cost change, candidate cost
//   control         Ceres' accept / reject logic, one workgroup
#include <type_traits>

#include "camera_model.hpp"
#include "../../include/vggsfm_amd.h"

namespace vgg {

int cholesky_solve_enqueue(double* A, double* b, int n, double* inv_blocks, int32_t* device_fail, const int32_t* skip_flag,
                           hipStream_t st, const CholOverlap* overlap, int split_a, int split_b, const int32_t* first_blk,
                           bool flags_cleared);
int32_t* cholesky_dataflow_flags(double* ws, int n, size_t* count);
size_t cholesky_workspace_bytes(int n);
void dataflow_signal(int32_t* flag, hipStream_t st);

constexpr int kGroup = 16;       // cameras per Schur tile side
#ifndef VGG_OFFDIAG_OCC
#define VGG_OFFDIAG_OCC 3   // wavefronts per SIMD of the off-diagonal Schur kernel (BD = 6): see DESIGN.md section 6
#endif
#ifndef VGG_DIAG_OCC
#define VGG_DIAG_OCC 4      // wavefronts per SIMD of the diagonal Schur kernel (BD = 6)
#endif
#ifndef VGG_TILE_INTERLEAVE
#define VGG_TILE_INTERLEAVE 1   // full-factor tiles: LDS writes of the next batch between the K steps of the current one
#endif
#ifndef VGG_DIAG_FULL_DEPTH
#define VGG_DIAG_FULL_DEPTH 2   // staging register sets of the full-factor diagonal tile launch
#endif
#ifndef VGG_OFF_FULL_DEPTH
#define VGG_OFF_FULL_DEPTH 2    // ... of the full-factor off-diagonal one
#endif
#ifndef VGG_TRF_ABLATE
#define VGG_TRF_ABLATE 0      // profiling builds, bits: 1 = no row products, 2 = no z loads, 4 = no store of the sums
#endif
#ifndef VGG_TRF_EARLY
#define VGG_TRF_EARLY 2   // full-factor tile_rhs: row products behind K step VGG_TRF_EARLY - 1 of the batch (0: behind the batch)
#endif
#ifndef VGG_PP_OCC_SPLIT
#define VGG_PP_OCC_SPLIT 3   // point_pass without the Y sweep: 158 VGPRs
#endif
#ifndef VGG_PP_ABLATE
#define VGG_PP_ABLATE 0              // profiling builds of point_pass_kernel: 1 = no Y stores, 2 = no Y sweep
#endif
#ifndef VGG_PP_OCC
#define VGG_PP_OCC 2
#endif
#ifndef VGG_PS_OCC_FY
#define VGG_PS_OCC_FY 2      // point_step_kernel without the Jacobian sweep
#endif
#ifndef VGG_PS_OCC
#define VGG_PS_OCC 2
#endif
#ifndef VGG_CP_OCC_RHS
#define VGG_CP_OCC_RHS 2
#endif
#ifndef VGG_CP_OCC
#define VGG_CP_OCC 2
#endif
#ifndef VGG_ABLATE
#define VGG_ABLATE 0
#endif
constexpr int kSub = 32;         // entries per strided sub-chunk of a Schur workgroup (see schur_tile_kernel)
constexpr int kMaxWG = 2048;
constexpr int kCamSplitMax = 16;  // workgroups per camera in the camera passes
constexpr int kCamNV = 48;        // >= values a camera pass accumulates (BD(BD+1)/2 + BD + 1 <= 45)
constexpr int kPackPad = 1024;    // >= ranks of a sharded solve (padding of the packed reduced system to equal slices)

struct Ctl {
  double radius, decrease_factor, x_cost, initial_cost, gmax_cams, gmax, cand_cost, mcc, step_norm, rel;
  int32_t iteration, done, termination, need_lin, scale_ready, invalid_streak, num_succ, num_unsucc;
  int32_t linear_fail, accept, rank, world, grad_logged, pad0, pad1, pad2;
};

struct Dims {
  int C, P, O, NI, model, kd, only_k, shared, BDp, n_red, kdsh, loss;
  double loss_scale;
};

// Doubles per slot (one observation) of the segment buffer Y.
// 6 x 6 camera blocks (shared or constant intrinsics): COMPRESSED -- the 6 x 3 Schur factor of an observation is
//   Y_i = s_c o (F_i^T E_i G_p) = s_c o [ [2 a]x ; I ] N_i,   N_i = Jw^T (E_i G_p) (3 x 3),  a = R_c X_p,
// because the pose Jacobian factors as F = Jw [ -2 [a]x | I ] (J_delta = -2 J_Y [R X]x, J_t = J_Y; the
// loss corrector multiplies from the left).  The slot holds N (row-major, 9) and 2 a (3): 96 bytes instead of 144; the tile
// kernel rebuilds the six rows while it stages the segment, the Jacobi scales s_c and the constant-parameter masks are
// applied to the finished tile sums (tile_reduce_kernel).  Larger blocks (per-camera intrinsics) keep the full factor.
constexpr int kYc = 12;
static inline size_t y_slot_doubles(const Dims& d) { return (d.shared || d.kd == 0) ? (size_t)kYc : (size_t)d.BDp * 3; }

struct Ws {  // device workspace carve-up (pointers into the caller's buffer)
  Ctl* ctl;
  vgg_ba_iteration* log;
  double *cand_q, *cand_t, *cand_intr, *cand_pts;
  double *scale_c, *scale_p, *colsq_c, *dsq_c, *dy;
  uint8_t* active;            // [n_red]
  double* lin;                // reduce buffer 0: U[C][BDp*BDp] | g[C][BDp] | cost[C]
  double *U, *g, *costc;
  double* sys;                // reduce buffer 1: S[n*n] | rhs[n]
  double *S, *rhs;
  double* S2;                 // [n*n] sums of the tile batches that are computed while the factorisation runs (overlap mode)
  double* gmax_pts;           // reduce buffer 2 (MAX): 1 double (padded to 8)
  double* stepsum;            // reduce buffer 3: cost, mcc, step_sq, xnorm_sq  (padded to 8)
  double *G, *hs, *Ms;        // per point: 6, 3, 3*kdsh
  double *pdamp;              // per point: the damping of its block in unscaled coordinates, D^2 = dd / s^2 (3)
  double* T;                  // [C][BDp][1+kdsh]
  double* part_B;             // [kMaxWG] per-workgroup gradient max
  double* part_F;             // [kMaxWG][4]
  double* packed;             // lower triangle of S (row by row) + rhs: the multi-GPU reduce payload, followed by kPackPad
  size_t packed_count;        //   doubles of padding (the reduce-scatter input is W equal slices: W ceil(count / W) doubles)
  double* pk_mine;            // [ceil(count / W) + 1] this rank's reduced slice + its local gradient maximum (reduce-scatter
                              //   output = all-gather input)
  double* pk_gathered;        // [W (ceil(count / W) + 1)] the all-gather output, read in place by the unpack (phase 6)
  double* cam_part;           // [C+1][3] step^2 / x^2 of camera-side parameters, camera-side share of the model cost change
  double* cam_split;          // [C][kCamSplitMax][kCamNV] partial sums of the split camera passes
  double* Y;                  // [num_segments][16][BDt*3] zero-padded per-observation Schur factors s_c o (F^T E G_p)
  size_t y_bytes;
  double* chol_inv;           // inverse diagonal blocks of the Cholesky factor
  double* tile_part;          // [num_chunks][R][R] partial Schur tiles, R = 16*BDt
  int32_t* batch_flags;       // [8] overlap mode: tile batch b >= 1 has been summed into S2 (raised by a kernel behind it)
  // reduced right-hand side from the diagonal tiles (tile_rhs, see schur_tile_body): per point Z = [z | y_0 | y_1] (3 x 3,
  // column-major) with E hs = E G z, E Ms_m = E G y_m; per diagonal chunk the sums over its entries of Y_seg Z (96 x 3);
  // per point-pass workgroup the shared-intrinsics terms sum_p Wa_p hs_p (KD) and sum_p Wa_p Ms_p^T (KD x KD)
  double *Zp, *rz_part, *part_Q;
  unsigned long long* split_off;   // split exchange: [2][n + 2] elements of part A / B in the rows above row i (i = n: the rhs; i = n + 1: the part's total)
  double* rz;                 // [ceil(C / 16)][96 x 3] rz_part summed over the chunks of a group's diagonal tile (tile_reduce_kernel)
  int tile_rhs;               // 1: cam_pass<RHS> is not launched, its sums come from the diagonal tile launch + point_pass
  int step_from_factors;      // 1: point_step_kernel takes E^T F dy from the compressed Schur factors (no Jacobian sweep)
  // round-6 A/B (vgg_ba_set_tile_dma): the off-diagonal tile launch stages by LDS-DMA from the EXPANDED image of the segments
  double* Yx;                 // [num_segments + 1][3][96]: a segment as the tile kernels hold it in LDS (component-major rows)
  size_t yx_bytes;
  int tile_dma;
  size_t lin_count, sys_count, total_bytes;
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }


static Dims make_dims(const vgg_ba_problem* pb) {
  Dims d;
  d.C = pb->num_cams; d.P = pb->num_pts; d.O = pb->num_obs; d.NI = pb->num_intr; d.model = pb->camera_model;
  const int rf = pb->refine_focal ? 1 : 0, rk = (pb->refine_extra && pb->camera_model == kSimpleRadial) ? 1 : 0;
  d.kd = rf + rk;
  d.only_k = (!rf && rk) ? 1 : 0;
  d.shared = (pb->num_intr == 1) ? 1 : 0;
  d.BDp = 6 + d.kd;
  d.n_red = 6 * d.C + d.kd * d.NI;
  d.kdsh = d.shared ? d.kd : 0;
  d.loss = pb->loss; d.loss_scale = pb->loss_scale;
  return d;
}

// overrides of the automatic launch choices (vgg_ba_tuning; the environment variables seed them): 0 / -1 = automatic
struct Tuning { int lpp, longt, cam_wgs, point_wgs, tile_rhs, step_factors, tile_dma; };
static Tuning g_tuning = [] {
  auto env = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
  return Tuning{env("VGG_LPP", 0), env("VGG_PP_LONGT", -1), env("VGG_CAM_WGS", 0), env("VGG_POINT_WGS", 0), env("VGG_TILE_RHS", 2), env("VGG_STEP_FACTORS", 0), env("VGG_TILE_DMA", 0)};
}();

static Ws carve(const Dims& d, int max_iters, int num_chunks, int num_segments, void* base) {
  Ws w;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return (char*)base + o; };
  w.ctl = (Ctl*)take(sizeof(Ctl));
  w.log = (vgg_ba_iteration*)take(sizeof(vgg_ba_iteration) * (size_t)(max_iters + 2));
  w.cand_q = (double*)take(8ull * 4 * d.C); w.cand_t = (double*)take(8ull * 3 * d.C);
  w.cand_intr = (double*)take(8ull * 4 * d.NI); w.cand_pts = (double*)take(8ull * 3 * d.P);
  w.scale_c = (double*)take(8ull * d.n_red); w.scale_p = (double*)take(8ull * 3 * d.P);
  w.colsq_c = (double*)take(8ull * d.n_red); w.dsq_c = (double*)take(8ull * d.n_red);
  w.dy = (double*)take(8ull * d.n_red);
  w.active = (uint8_t*)take(d.n_red);
  w.lin_count = (size_t)d.C * d.BDp * d.BDp + (size_t)d.C * d.BDp + d.C;
  w.lin = (double*)take(8ull * w.lin_count);
  w.U = w.lin; w.g = w.U + (size_t)d.C * d.BDp * d.BDp; w.costc = w.g + (size_t)d.C * d.BDp;
  w.sys_count = (size_t)d.n_red * d.n_red + d.n_red;
  w.sys = (double*)take(8ull * w.sys_count);
  w.S = w.sys; w.rhs = w.S + (size_t)d.n_red * d.n_red;
  w.S2 = (double*)take(8ull * (size_t)d.n_red * d.n_red);
  w.packed_count = (size_t)d.n_red * (d.n_red + 1) / 2 + d.n_red;
  w.packed = (double*)take(8ull * (w.packed_count + kPackPad));
  w.pk_mine = (double*)take(8ull * (w.packed_count + 2));                      // (W = 1: the whole payload)
  w.pk_gathered = (double*)take(8ull * (w.packed_count + 2 * kPackPad));
  w.gmax_pts = (double*)take(64);
  w.stepsum = (double*)take(64);
  w.G = (double*)take(8ull * 6 * d.P); w.hs = (double*)take(8ull * 3 * d.P);
  w.pdamp = (double*)take(8ull * 3 * d.P);
  w.Ms = (double*)take(8ull * 3 * (d.kdsh ? d.kdsh : 1) * d.P);
  w.T = (double*)take(8ull * d.C * d.BDp * (1 + d.kdsh));
  w.part_B = (double*)take(8ull * kMaxWG);
  w.part_F = (double*)take(8ull * kMaxWG * 4);
  w.cam_part = (double*)take(8ull * (d.C + 1) * 3);
  w.cam_split = (double*)take(8ull * (size_t)d.C * kCamSplitMax * kCamNV);
  // + one all-zero segment behind the last real one (target of the tile kernel's loads past the end of a list)
  w.y_bytes = 8ull * ((size_t)(num_segments > 0 ? num_segments : 0) + 1) * kGroup * y_slot_doubles(d);
  w.Y = (double*)take(w.y_bytes);
  w.yx_bytes = (g_tuning.tile_dma && (d.shared || d.kd == 0)) ? 8ull * ((size_t)(num_segments > 0 ? num_segments : 0) + 1) * kGroup * 18 : 0;
  w.Yx = w.yx_bytes ? (double*)take(w.yx_bytes) : nullptr;
  w.tile_dma = 0;
  w.chol_inv = (double*)take(cholesky_workspace_bytes(d.n_red));
  {
    const size_t bdt = d.shared ? 6 : d.BDp;
    w.tile_part = (double*)take(8ull * (size_t)(num_chunks > 0 ? num_chunks : 1) * bdt * bdt * 256);   // R*R, R = 16*bdt
  }
  w.batch_flags = (int32_t*)take(64);
  w.Zp = (double*)take(8ull * 9 * (d.P > 0 ? d.P : 1));
  w.rz_part = (double*)take(8ull * (size_t)(num_chunks > 0 ? num_chunks : 1) * kGroup * 6 * 3);
  w.part_Q = (double*)take(8ull * kMaxWG * 8);
  w.rz = (double*)take(8ull * (size_t)((d.C + kGroup - 1) / kGroup) * kGroup * 6 * 3);
  w.split_off = (unsigned long long*)take(8ull * 2 * ((size_t)d.n_red + 2));
  w.tile_rhs = 0;
  w.total_bytes = off;
  return w;
}

struct DevProblem {  // by-value kernel argument
  Dims d;
  const double *cam_q, *cam_t, *intr, *pts;
  const int32_t *row_ptr, *obs_cam, *col_ptr, *cobs_pt, *obs_slot;
  const float2 *obs_uv, *cobs_uv;
  const uint8_t *cam_const, *intr_const, *pt_const;
};

static DevProblem dev_problem(const vgg_ba_problem* pb, const Dims& d) {
  DevProblem p;
  p.d = d; p.cam_q = pb->cam_q; p.cam_t = pb->cam_t; p.intr = pb->intr; p.pts = pb->pts;
  p.row_ptr = pb->row_ptr; p.obs_cam = pb->obs_cam; p.col_ptr = pb->col_ptr; p.cobs_pt = pb->cobs_pt;
  p.obs_slot = pb->obs_slot;
  p.obs_uv = (const float2*)pb->obs_uv; p.cobs_uv = (const float2*)pb->cobs_uv;
  p.cam_const = pb->cam_const; p.intr_const = pb->intr_const; p.pt_const = pb->pt_const;
  return p;
}

// ---------------------------------------------------------------------------------------------
// One observation: corrected residual and corrected, constant-masked, UNSCALED Jacobians.
//   F[2][6+KD]  (pose tangent 6, refined intrinsics KD), E[2][3] (point); Rm = rotation matrix of the camera (row-major)
// (camera, pixel, slot) of the first N observations of a lane, prefetched with the point.  Named scalar members and
// explicit selects: arrays filled in (later unrolled) loops stayed dynamically indexed stack objects -- scratch.
template <int N>
struct ObsPf {
  static_assert(N == 2 || N == 4, "prefetch depth");
  int c0 = 0, c1 = 0, c2 = 0, c3 = 0, s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  float2 u0 = make_float2(0.f, 0.f), u1 = make_float2(0.f, 0.f), u2 = make_float2(0.f, 0.f), u3 = make_float2(0.f, 0.f);
  template <bool SLOT>
  __device__ __forceinline__ void load(const int32_t* cam, const float2* uv, const int32_t* slot, int o0, int o1, int stride, int sl) {
    int o = o0 + sl;
    if (o < o1) { c0 = cam[o]; u0 = uv[o]; if (SLOT) s0 = slot[o]; }
    o += stride;
    if (o < o1) { c1 = cam[o]; u1 = uv[o]; if (SLOT) s1 = slot[o]; }
    if (N == 4) {
      o += stride;
      if (o < o1) { c2 = cam[o]; u2 = uv[o]; if (SLOT) s2 = slot[o]; }
      o += stride;
      if (o < o1) { c3 = cam[o]; u3 = uv[o]; if (SLOT) s3 = slot[o]; }
    }
  }
  template <typename T>
  __device__ __forceinline__ static T pick(T a0, T a1, T a2, T a3, int k) {
    const T lo = (k & 1) ? a1 : a0, hi = (k & 1) ? a3 : a2;
    return (N == 4 && (k & 2)) ? hi : lo;
  }
  __device__ __forceinline__ int cam(int k, const int32_t* src, int o) const { return k < N ? pick(c0, c1, c2, c3, k) : src[o]; }
  __device__ __forceinline__ int slot(int k, const int32_t* src, int o) const { return k < N ? pick(s0, s1, s2, s3, k) : src[o]; }
  __device__ __forceinline__ float2 uv(int k, const float2* src, int o) const {
    if (k >= N) return src[o];
    return make_float2(pick(u0.x, u1.x, u2.x, u3.x, k), pick(u0.y, u1.y, u2.y, u3.y, k));
  }
};

struct CamR {                                    // rotation matrix of a quaternion in global memory (paths without the LDS cache)
  double R[9];
  __device__ __forceinline__ explicit CamR(const double* q) { quat_to_R(q, R); }
};

template <int KD>
__device__ __forceinline__ double eval_full(const Dims& d, const double* Rm, const double* t, const double* in4,
                                            const double* X, float2 uv, unsigned camflag, bool intr_c, bool pt_c,
                                            double* r, double* F, double* E, double* Jw = nullptr) {
  constexpr int BD = 6 + KD;
  double Jp[12], Ji[4];
  obs_eval_R(d.model, Rm, t, in4, X, (double)uv.x, (double)uv.y, r, Jp, Ji, E);
#pragma unroll
  for (int row = 0; row < 2; ++row) {
#pragma unroll
    for (int k = 0; k < 6; ++k) F[row * BD + k] = Jp[row * 6 + k];
    if (KD == 2) { F[row * BD + 6] = Ji[row * 2]; F[row * BD + 7] = Ji[row * 2 + 1]; }
    if (KD == 1) F[row * BD + 6] = d.only_k ? Ji[row * 2 + 1] : Ji[row * 2];
  }
  const double s = r[0] * r[0] + r[1] * r[1];
  double rho[3];
  loss_eval(d.loss, d.loss_scale, s, rho);
  if (d.loss != kLossTrivial) {
    Corrector c(s, rho);
    c.jac<BD>(r, F);
    c.jac<3>(r, E);
    r[0] *= c.residual_scaling; r[1] *= c.residual_scaling;
  }
  // Jw = corrected d r / d (R X + t) (2 x 3) BEFORE the constant-parameter masks: the translation columns of F
  if (Jw) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { Jw[k] = F[3 + k]; Jw[3 + k] = F[BD + 3 + k]; }
  }
  if (camflag) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (camflag & 1u) { F[k] = 0; F[BD + k] = 0; }
      if ((camflag & 1u) || (camflag & (2u << k))) { F[3 + k] = 0; F[BD + 3 + k] = 0; }
    }
  }
  if (intr_c) {
#pragma unroll
    for (int k = 0; k < KD; ++k) { F[6 + k] = 0; F[BD + 6 + k] = 0; }
  }
  if (pt_c) {
#pragma unroll
    for (int k = 0; k < 6; ++k) E[k] = 0;
  }
  return rho[0];
}

__device__ __forceinline__ double loss_rho0(const Dims& d, double s) {
  double rho[3];
  loss_eval(d.loss, d.loss_scale, s, rho);
  return rho[0];
}

// block-wide sum of NV per-thread values (256 threads); result valid on thread 0..NV-1 via out[]
template <int NV>
__device__ __forceinline__ void block_sum(double* v, double* lds /* [4][NV] */, double* out /* lds [NV] */) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const double s = wave_sum(v[i]);
    if (lane == 0) lds[wave * NV + i] = s;
  }
  __syncthreads();
  if (threadIdx.x < NV) out[threadIdx.x] = lds[threadIdx.x] + lds[NV + threadIdx.x] + lds[2 * NV + threadIdx.x] + lds[3 * NV + threadIdx.x];
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
__global__ void init_kernel(DevProblem pb, Ws w, vgg_ba_options opt, int rank, int world) {
  const Dims& d = pb.d;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j == 0) {
    Ctl c;
    memset(&c, 0, sizeof(c));
    c.radius = opt.initial_trust_region_radius; c.decrease_factor = 2.0; c.need_lin = 1; c.rank = rank; c.world = world;
    *w.ctl = c;
    w.gmax_pts[0] = 0; for (int i = 0; i < 4; ++i) w.stepsum[i] = 0;
  }
  if (j < d.n_red) {
    bool act;
    if (j < 6 * d.C) {
      const int c = j / 6, k = j - 6 * c;
      const unsigned f = pb.cam_const ? pb.cam_const[c] : 0u;
      act = (pb.col_ptr[c + 1] > pb.col_ptr[c]) || world > 1;
      if (f & 1u) act = false;
      if (k >= 3 && (f & (2u << (k - 3)))) act = false;
    } else {
      const int a = (j - 6 * d.C) / d.kd;
      act = !(pb.intr_const && pb.intr_const[a]);
      if (!d.shared && world == 1 && !(pb.col_ptr[a + 1] > pb.col_ptr[a])) act = false;
    }
    w.active[j] = act ? 1 : 0;
    w.scale_c[j] = 1.0;
  }
}

// Zeroes the part of the reduced system S | rhs that anything reads: only the lower triangle is ever filled (tile_reduce_kernel
// and assemble_kernel write S[hi][lo], the factorisation and pack_lower_kernel read it), so row i is cleared up to the end of
// the 64-column block of its diagonal element and the upper triangle is left alone -- half the stores of a full clear (at the
// 6002 x 6002 system of the final joint adjustment of configs[4]: 144 of 288 MB per iteration).  Workgroup `wg` of `nwg` takes
// the rows wg, wg + nwg, ...; row n is the right-hand side.
__device__ __forceinline__ void zero_system_lower(const Ws& w, int n, int wg, int nwg) {
  const bool even = (n & 1) == 0;                 // (rows 16-byte aligned)
  for (int row = wg; row <= n; row += nwg) {
    const int len = (row < n) ? min(n, 64 * (row / 64 + 1)) : n;
    double* dst = w.sys + (size_t)row * n;
    if (even) {
      double2* d2 = reinterpret_cast<double2*>(dst);
      for (int c = threadIdx.x; c < (len + 1) / 2; c += 256) d2[c] = make_double2(0.0, 0.0);
    } else {
      for (int c = threadIdx.x; c < len; c += 256) dst[c] = 0.0;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// camera-major pass.  MODE 0: linearisation terms U_c, g_c, cost.  MODE 1: T_c = F^T [r - E hs | -E Ms].
template <int KD, int MODE>
__global__ __launch_bounds__(256, (MODE == 1) ? VGG_CP_OCC_RHS : VGG_CP_OCC) void cam_pass_kernel(DevProblem pb, Ws w) {
  constexpr int BD = 6 + KD;
  constexpr int NU = BD * (BD + 1) / 2;
  constexpr int NV = (MODE == 0) ? (NU + BD + 1) : (BD * (1 + KD));
  __shared__ double red[4 * NV];
  __shared__ double tot[NV];
  if (w.ctl->done) return;
  if (MODE == 0 && !w.ctl->need_lin) return;
  if (MODE == 1) {
    // Zero the reduced system S | rhs for the tile sums and assemble_kernel that follow (nothing touches it in between, and
    // the previous iteration is done with it): a few 16-byte stores per thread here instead of a fill launch
    zero_system_lower(w, pb.d.n_red, (int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
  }
  const Dims& d = pb.d;
  // grid (C, split): a camera's observations are cut into `split` contiguous slices, one workgroup each (one
  // workgroup per camera left 200 workgroups = 0.8 wavefronts per SIMD on the chip); cam_reduce_kernel adds the
  // slices in a fixed order
  const int c = blockIdx.x, split = gridDim.y;
  double q[9], t[3], in4[4];                   // q: the camera's rotation MATRIX
  quat_to_R(pb.cam_q + 4 * c, q);
#pragma unroll
  for (int k = 0; k < 3; ++k) t[k] = pb.cam_t[3 * c + k];
  const int a = d.shared ? 0 : c;
#pragma unroll
  for (int k = 0; k < 4; ++k) in4[k] = pb.intr[4 * a + k];
  const unsigned camflag = pb.cam_const ? pb.cam_const[c] : 0u;
  const bool intr_c = pb.intr_const ? pb.intr_const[a] != 0 : false;
  const int kdsh = d.kdsh;
  double acc[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) acc[i] = 0.0;
  const int j_begin = pb.col_ptr[c], j_count = pb.col_ptr[c + 1] - j_begin;
  const int per = (j_count + split - 1) / split;
  const int j0 = j_begin + (int)blockIdx.y * per, j1 = min(j0 + per, j_begin + j_count);
  // Software pipeline over this thread's observations: (point index, pixel) are loaded AHEAD + 1 iterations ahead, the point
  // coordinates (and, MODE 1, its back-substitution terms) -- a gather that depends on the index -- AHEAD iterations ahead:
  // the two dependent memory round trips of an observation overlap the arithmetic of the observations before it.
  // (Without it the pass ran 5x off its instruction-issue bound at 3-4 wavefronts per SIMD; one stage less: 52 % of the
  // wave cycles still waiting.)
  constexpr int KM = (KD > 0) ? KD : 1;
  struct Gathered { double X[3], h[3], M[3 * KM]; float2 uv; int c; };   // (c: the constant-point flag as loaded, tested where it is used)
  auto gather = [&](Gathered& g, int p, float2 uv) __attribute__((always_inline)) {
    g.uv = uv;
    g.X[0] = pb.pts[3 * p]; g.X[1] = pb.pts[3 * p + 1]; g.X[2] = pb.pts[3 * p + 2];
    g.c = pb.pt_const ? (int)pb.pt_const[p] : 0;
    if (MODE == 1) {
      g.h[0] = w.hs[3 * p]; g.h[1] = w.hs[3 * p + 1]; g.h[2] = w.hs[3 * p + 2];
#pragma unroll
      for (int m = 0; m < KD; ++m)
        if (m < kdsh) {
          const double* M = w.Ms + ((size_t)p * kdsh + m) * 3;
          g.M[3 * m] = M[0]; g.M[3 * m + 1] = M[1]; g.M[3 * m + 2] = M[2];
        }
    }
  };
  // (MODE 1 carries more per observation and runs out of registers with the second gather stage: 16 bytes of scratch and
  //  0.133 -> 0.136 ms at c3, while MODE 0 goes from 0.099 to 0.087 ms -- so one stage there, two here)
  constexpr int AHEAD = (MODE == 0 && KD > 0) ? 2 : 1;     // gather stages (KD = 0, MODE 0: the second stage would cost the third wavefront per SIMD)
  Gathered g1 = {}, g2 = {};
  int j = j0 + threadIdx.x;
  int p3 = 0; float2 uv3 = make_float2(0.f, 0.f);
  if (j < j1) gather(g1, pb.cobs_pt[j], pb.cobs_uv[j]);
  if (AHEAD == 2 && j + 256 < j1) gather(g2, pb.cobs_pt[j + 256], pb.cobs_uv[j + 256]);
  if (j + 256 * AHEAD < j1) { p3 = pb.cobs_pt[j + 256 * AHEAD]; uv3 = pb.cobs_uv[j + 256 * AHEAD]; }
  for (; j < j1; j += 256) {
    const Gathered cur = g1;
    const float2 uv = cur.uv;
    const double X[3] = {cur.X[0], cur.X[1], cur.X[2]};
    const bool pt_c = cur.c != 0;
    const double h0 = cur.h[0], h1v = cur.h[1], h2 = cur.h[2];
    double Mc[3 * KM];
#pragma unroll
    for (int i = 0; i < 3 * KM; ++i) Mc[i] = cur.M[i];
    // advance the pipeline: gather for iteration j + 512 (its index arrived an iteration ago), index for j + 768
    if (AHEAD == 2) {
      g1 = g2;
      if (j + 512 < j1) gather(g2, p3, uv3);
    } else if (j + 256 < j1) gather(g1, p3, uv3);
    if (j + 256 * (AHEAD + 1) < j1) { p3 = pb.cobs_pt[j + 256 * (AHEAD + 1)]; uv3 = pb.cobs_uv[j + 256 * (AHEAD + 1)]; }
    double r[2], F[2 * BD], E[6];
    const double rho0 = eval_full<KD>(d, q, t, in4, X, uv, camflag, intr_c, pt_c, r, F, E);
    if (MODE == 0) {
      int u = 0;
#pragma unroll
      for (int i = 0; i < BD; ++i)
#pragma unroll
        for (int k = i; k < BD; ++k) acc[u++] += F[i] * F[k] + F[BD + i] * F[BD + k];
#pragma unroll
      for (int i = 0; i < BD; ++i) acc[NU + i] += F[i] * r[0] + F[BD + i] * r[1];
      acc[NU + BD] += rho0;
    } else {
      double R0[1 + KD], R1[1 + KD];
      R0[0] = r[0] - (E[0] * h0 + E[1] * h1v + E[2] * h2);
      R1[0] = r[1] - (E[3] * h0 + E[4] * h1v + E[5] * h2);
#pragma unroll
      for (int m = 0; m < KD; ++m) {
        if (m < kdsh) {
          const double* M = Mc + 3 * m;
          R0[1 + m] = -(E[0] * M[0] + E[1] * M[1] + E[2] * M[2]);
          R1[1 + m] = -(E[3] * M[0] + E[4] * M[1] + E[5] * M[2]);
        } else { R0[1 + m] = 0; R1[1 + m] = 0; }
      }
#pragma unroll
      for (int i = 0; i < BD; ++i)
#pragma unroll
        for (int m = 0; m < 1 + KD; ++m) acc[i * (1 + KD) + m] += F[i] * R0[m] + F[BD + i] * R1[m];
    }
  }
  block_sum<NV>(acc, red, tot);
  static_assert(NV <= kCamNV, "cam_split row");
  if (threadIdx.x < NV) w.cam_split[((size_t)c * kCamSplitMax + blockIdx.y) * kCamNV + threadIdx.x] = tot[threadIdx.x];
}

// sums the `split` slices of a camera in order and stores U, g, cost (MODE 0) or T (MODE 1)
template <int KD, int MODE>
__global__ __launch_bounds__(64) void cam_reduce_kernel(DevProblem pb, Ws w, int split, int point_parts) {
  constexpr int BD = 6 + KD;
  constexpr int NU = BD * (BD + 1) / 2;
  constexpr int NV = (MODE == 0) ? (NU + BD + 1) : (BD * (1 + KD));
  if (w.ctl->done) return;
  if (MODE == 0 && !w.ctl->need_lin) return;
  if (MODE == 1 && blockIdx.x == 0) {            // (rides along: max of the point passes' per-workgroup gradient norms)
    double m = 0;
    for (int i = threadIdx.x; i < point_parts; i += 64) m = fmax(m, w.part_B[i]);
    m = wave_max(m);
    if (threadIdx.x == 0) w.gmax_pts[0] = m;
  }
  const Dims& d = pb.d;
  const int c = blockIdx.x, kdsh = d.kdsh;
  if (threadIdx.x >= NV) return;
  double tot = 0.0;
#pragma unroll 8
  for (int sp = 0; sp < split; ++sp) tot += w.cam_split[((size_t)c * kCamSplitMax + sp) * kCamNV + threadIdx.x];
  if (MODE == 0) {
    double* U = w.U + (size_t)c * BD * BD;
    if (threadIdx.x < NU) {
      int i = 0, rem = threadIdx.x;
      while (rem >= BD - i) { rem -= BD - i; ++i; }
      const int k = i + rem;
      U[i * BD + k] = tot;
      U[k * BD + i] = tot;
    } else if (threadIdx.x < NU + BD) {
      w.g[(size_t)c * BD + threadIdx.x - NU] = tot;
    } else {
      w.costc[c] = tot;
    }
  } else {
    // stored with row stride (1 + kdsh)
    const int i = threadIdx.x / (1 + KD), m = threadIdx.x - i * (1 + KD);
    if (m < 1 + kdsh) w.T[((size_t)c * BD + i) * (1 + kdsh) + m] = tot;
  }
}

// after (the all-reduce of) buffer 0: column norms, Jacobi scaling (first time), camera-side gradient max, cost
template <int KD>
__device__ __forceinline__ void prep_body(const DevProblem& pb, const Ws& w, const vgg_ba_options& opt, double* red) {
  constexpr int BD = 6 + KD;
  Ctl* ctl = w.ctl;
  const Dims& d = pb.d;
  const bool first = !ctl->scale_ready;
  double gmax = 0.0, cost = 0.0;
  constexpr int NS = 1 + 2 * (KD > 0 ? KD : 1);
  double sums[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) sums[k] = 0.0;
  for (int c = threadIdx.x; c < d.C; c += 256) {
    const double* U = w.U + (size_t)c * BD * BD;
    const double* g = w.g + (size_t)c * BD;
    cost += w.costc[c];
    if (d.shared && KD > 0) {                      // shared intrinsics: column norm and gradient are sums over all cameras
#pragma unroll
      for (int k = 0; k < KD; ++k) { sums[1 + k] += U[(6 + k) * BD + 6 + k]; sums[1 + KD + k] += g[6 + k]; }
    }
    for (int k = 0; k < 6; ++k) w.colsq_c[6 * c + k] = U[k * BD + k];
    if (!d.shared) for (int k = 0; k < KD; ++k) w.colsq_c[6 * d.C + KD * c + k] = U[(6 + k) * BD + 6 + k];
    // |Plus(x, -g) - x|_inf for this camera (Ceres projected-gradient norm)
    double dl[3], qn[4];
    for (int k = 0; k < 3; ++k) dl[k] = w.active[6 * c + k] ? -g[k] : 0.0;
    quat_plus(pb.cam_q + 4 * c, dl, qn);
    for (int k = 0; k < 4; ++k) gmax = fmax(gmax, fabs(qn[k] - pb.cam_q[4 * c + k]));
    for (int k = 3; k < 6; ++k) if (w.active[6 * c + k]) gmax = fmax(gmax, fabs(g[k]));
    if (!d.shared) for (int k = 0; k < KD; ++k) if (w.active[6 * d.C + KD * c + k]) gmax = fmax(gmax, fabs(g[6 + k]));
  }
  // one block reduction for everything (round 4: six 256-thread trees of eight barriers each took 15 us of every iteration):
  // wave shuffles, then the four wavefronts' partials in a fixed order
  sums[0] = cost;
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const double v = wave_sum(sums[k]);
    if (ln == 0) red[wv * 8 + k] = v;
  }
  gmax = wave_max(gmax);
  if (ln == 0) red[32 + wv] = gmax;
  __syncthreads();
  const double cost_total = (red[0] + red[8]) + (red[16] + red[24]);
  double gm = fmax(fmax(red[32], red[33]), fmax(red[34], red[35]));
  if (d.shared && KD > 0) {
#pragma unroll
    for (int k = 0; k < KD; ++k) {
      const double cs = (red[1 + k] + red[9 + k]) + (red[17 + k] + red[25 + k]);
      const double gs = (red[1 + KD + k] + red[9 + KD + k]) + (red[17 + KD + k] + red[25 + KD + k]);
      if (threadIdx.x == 0) w.colsq_c[6 * d.C + k] = cs;
      if (w.active[6 * d.C + k]) gm = fmax(gm, fabs(gs));
    }
  }
  __syncthreads();                                 // (colsq_c is read below by other threads)
  if (first) {
    for (int j = threadIdx.x; j < d.n_red; j += 256)
      w.scale_c[j] = opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(w.colsq_c[j])) : 1.0;
  }
  if (threadIdx.x == 0) {
    ctl->gmax_cams = gm;
    if (first) { ctl->x_cost = 0.5 * cost_total; ctl->initial_cost = ctl->x_cost; }
  }
}


// after (the all-reduce of) buffer 0: prep_body when a new linearisation is in place, then -- every iteration -- the LM
// damping of the reduced columns for the current radius (one launch: both are a few hundred elements)
template <int KD>
__global__ __launch_bounds__(256) void prep_kernel(DevProblem pb, Ws w, vgg_ba_options opt) {
  __shared__ double red[256];
  Ctl* ctl = w.ctl;
  if (ctl->done) return;
  if (ctl->need_lin) prep_body<KD>(pb, w, opt, red);       // (uniform branch: the body synchronises the workgroup)
  __syncthreads();
  const int n_red = pb.d.n_red;
  const double radius = ctl->radius;
#pragma unroll 4
  for (int j = threadIdx.x; j < n_red; j += 256) {
    const double sc = w.scale_c[j];
    double dd = w.colsq_c[j] * sc * sc;
    dd = fmin(fmax(dd, opt.min_lm_diagonal), opt.max_lm_diagonal);
    w.dsq_c[j] = dd / radius;
  }
}

// VGG_SPLIT_POINT_PASS=1 in the environment selects the split form of the point pass -- wave-per-point reductions without
// the Y sweep (158 VGPRs, 3 wavefronts per SIMD) + y_write_kernel (thread per observation, 127 VGPRs).  Built and measured
// in round 2 (c3): 0.25 + 0.47 ms against 0.43 ms for the fused pass -- the reductions did not speed up with the third
// wavefront and a thread-per-observation writer without the LDS camera table is slower than the in-wave sweep -- so
// the fused pass stays the default.
// Lanes per point of the point passes from the mean track length (VGG_LPP=8|16|32|64 overrides).  A point's lanes share
// its serial work (reductions, 3 x 3 factorisation), which outweighs the sweeps over its observations up to ~70 of them:
// measured per LM iteration, point_pass + point_step -- c2 (mean 12.5 observations) 64: 0.112, 32: 0.075, 16: 0.065,
// 8: 0.059 ms; c3 (mean 50) 64: 0.674, 32: 0.549, 16: 0.499, 8: 0.535 ms; one c4 shard (mean 100) 32: 0.509, 16: 0.544 ms.
static int lanes_per_point(int P, int O) {
  const int forced = g_tuning.lpp;
  if (forced == 8 || forced == 16 || forced == 32 || forced == 64) return forced;
  const double mean = P > 0 ? (double)O / P : 64.0;
  return mean <= 72.0 ? 16 : 32;
}
// long-track variants of the point passes (four prefetched observations per lane, no cached Jacobians)
static bool long_tracks(int lpp, int P, int O) {
  return g_tuning.longt >= 0 ? g_tuning.longt != 0 : (double)O > 1.5 * lpp * (double)P;
}

// ---------------------------------------------------------------------------------------------
// point-major pass: LPP lanes per point, 64 / LPP points per wavefront (LPP = 64: one wavefront per point; 32 / 16 for
// short tracks -- the per-point work that every lane repeats (nine reductions, the 3 x 3 factorisation, ~360 of the ~640
// instructions of a point at 50 observations) is shared by 2 / 4 points, and a 12-observation track fills 12 of 16 lanes
// instead of 12 of 64).  The lanes of a point reduce among themselves (xor offsets < LPP); observations beyond the first
// LPP of a track are re-evaluated in the Y sweep.
// CY: the tile blocks are 6 x 6 (shared or constant intrinsics) and the segment buffer holds the COMPRESSED factors
// (kYc doubles per observation: N and 2 a, see y_slot_doubles); otherwise the full BD x 3 factor.
template <int KD, bool LDSCAM, bool CY, int LPP, bool LONGT = false>
__global__ __launch_bounds__(256, VGG_PP_OCC) void point_pass_kernel(DevProblem pb, Ws w, vgg_ba_options opt) {
  constexpr int BD = 6 + KD;
  __shared__ double wmax[4];
  extern __shared__ double cam_cache[];          // LDSCAM: R[9C] t[3C] pose scales[6C] flags[C] (as doubles)
  Ctl* ctl = w.ctl;
  if (ctl->done) return;
  const Dims& d = pb.d;
  constexpr int PPW = 64 / LPP;                  // points per wavefront
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane / LPP, sl = lane % LPP;   // point of the wavefront's group, lane inside the point
  const int nw = gridDim.x * 4 * PPW;            // points per sweep of the grid
  const bool first = !ctl->scale_ready;
  const double radius = ctl->radius;
  const int kdsh = d.kdsh;
  double gmax = 0.0;
  const bool trhs = w.tile_rhs != 0;
  if (trhs) {
    // (cam_pass<RHS> is not launched: its other job -- zeroing the reduced system S | rhs for the tile sums and
    //  assemble_kernel that follow -- is done here, a few 16-byte stores per thread)
    zero_system_lower(w, pb.d.n_red, (int)blockIdx.x, (int)gridDim.x);
  }
  // shared-intrinsics terms of the reduced system that cam_pass<RHS> summed per observation (tile_rhs): they are per-POINT
  // quantities -- sum_i Ji^T E_i hs_p = Wa_p hs_p, sum_i Ji^T E_i Ms_p = Wa_p Ms_p -- added up here, one lane per point
  // (accumulators in LDS, one row per (wavefront, point slot) owned by that point's first lane: the pass has no registers to
  //  spare -- six more doubles per lane put the long-track variant on the stack)
  constexpr int KQ = (KD > 0) ? KD + KD * KD : 1;
  __shared__ double qs[4][64 / LPP][KQ];
  // compressed factors: a wavefront's 64 records of a sweep go through LDS so that SIX adjacent lanes write the six 16-byte
  // pieces of ONE record (see flush_records below)
  __shared__ __attribute__((aligned(16))) double2 ystage[CY ? 4 : 1][CY ? 64 * (kYc / 2) : 1];
  __shared__ int32_t slotstage[CY ? 4 : 1][CY ? 64 : 1];
  if (trhs && KD > 0 && (lane % LPP) == 0) {
#pragma unroll
    for (int i = 0; i < KQ; ++i) qs[wave][lane / LPP][i] = 0.0;
  }
  // The per-observation camera gather is the second of three dependent memory round trips of a point; with the
  // cameras in LDS it is an LDS read.  (The wavefronts are latency bound: SQ_WAIT_ANY 60 %, 2 waves/SIMD.)
  const double* lq = cam_cache;                  // rotation matrices [9C]
  const double* lt = lq + 9 * d.C;
  const double* lsc = lt + 3 * d.C;
  const double* lfl = lsc + 6 * d.C;
  if (LDSCAM) {
    for (int i = threadIdx.x; i < d.C; i += 256) {
      quat_to_R(pb.cam_q + 4 * i, cam_cache + 9 * i);
      cam_cache[18 * d.C + i] = pb.cam_const ? (double)pb.cam_const[i] : 0.0;
    }
    for (int i = threadIdx.x; i < 3 * d.C; i += 256) cam_cache[9 * d.C + i] = pb.cam_t[i];
    for (int i = threadIdx.x; i < 6 * d.C; i += 256) cam_cache[12 * d.C + i] = w.scale_c[i];
    __syncthreads();
  }
  // software pipeline over the points of this wavefront: the row bounds / coordinates of the NEXT point and the
  // camera index, pixel and slot of its first 64 observations are loaded while the current point is processed
  int p = (blockIdx.x * 4 + wave) * PPW + sub;   // (the lanes of one point run the same control flow: per-lane loops below)
  // (camera, pixel, slot) of the lane's first NPF observations (o0 + sl + k LPP) are prefetched with the point; later
  // ones are loaded where they are used
  // LONGT (tracks of several sweeps: mean length > 1.5 LPP): four observations per lane prefetched and NO cached Jacobians
  // (the cache serves one sweep in four there and costs 36 registers) -- every load of a point is then issued before the
  // Y stores of the previous one (loads issued behind a burst of stores wait for the stores' acknowledgements)
  constexpr int NPF = LONGT ? 4 : 2;
  constexpr bool CACHEJ = !LONGT;
  int n_o0 = 0, n_o1 = 0;
  ObsPf<NPF> n_pf;
  double n_X0 = 0, n_X1 = 0, n_X2 = 0;
  int n_ptc = 0;                                // (constant-point flag as loaded: tested where it is used -- a test right
                                                //  behind the load is a wait for it, and for every load issued before it)
  // (two stages: the row bounds / coordinates are loaded TWO points ahead, the observations ONE point ahead, so
  //  that the observation loads never wait for the row-bound load they depend on)
  int m_o0 = 0, m_o1 = 0;
  double m_X0 = 0, m_X1 = 0, m_X2 = 0;
  int m_ptc = 0;
  if (p < d.P) {
    n_o0 = pb.row_ptr[p]; n_o1 = pb.row_ptr[p + 1];
    n_X0 = pb.pts[3 * p]; n_X1 = pb.pts[3 * p + 1]; n_X2 = pb.pts[3 * p + 2];
    n_ptc = pb.pt_const ? (int)pb.pt_const[p] : 0;
    n_pf.template load<true>(pb.obs_cam, pb.obs_uv, pb.obs_slot, n_o0, n_o1, LPP, sl);
    if (p + nw < d.P) {
      const int pm = p + nw;
      m_o0 = pb.row_ptr[pm]; m_o1 = pb.row_ptr[pm + 1];
      m_X0 = pb.pts[3 * pm]; m_X1 = pb.pts[3 * pm + 1]; m_X2 = pb.pts[3 * pm + 2];
      m_ptc = pb.pt_const ? (int)pb.pt_const[pm] : 0;
    }
  }
  // (compressed factors: the loop is WAVE-uniform -- the record flush below needs all 64 lanes; a lane group whose point index
  //  has run past the end goes through a last trip with an empty point: o0 = o1 = 0, nothing read or written for it)
  for (; CY ? __any(p < d.P) : (p < d.P); p += nw) {
    const bool pvalid = p < d.P;
    const int o0 = pvalid ? n_o0 : 0, o1 = pvalid ? n_o1 : 0;
    const double X[3] = {n_X0, n_X1, n_X2};
    const bool pt_c = n_ptc != 0;
    const ObsPf<NPF> f_pf = n_pf;
    {
      // stage 1 -> current of the next iteration: observations of point p + nw (its bounds arrived an iteration ago)
      n_o0 = m_o0; n_o1 = m_o1; n_X0 = m_X0; n_X1 = m_X1; n_X2 = m_X2; n_ptc = m_ptc;
      if (p + nw < d.P) n_pf.template load<true>(pb.obs_cam, pb.obs_uv, pb.obs_slot, n_o0, n_o1, LPP, sl);
      // stage 2: bounds / coordinates of point p + 2 nw
      const int pm = p + 2 * nw;
      if (pm < d.P) {
        m_o0 = pb.row_ptr[pm]; m_o1 = pb.row_ptr[pm + 1];
        m_X0 = pb.pts[3 * pm]; m_X1 = pb.pts[3 * pm + 1]; m_X2 = pb.pts[3 * pm + 2];
        m_ptc = pb.pt_const ? (int)pb.pt_const[pm] : 0;
      }
    }
    double V[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0}, Wa[3 * (KD ? KD : 1)];
#pragma unroll
    for (int i = 0; i < 3 * (KD ? KD : 1); ++i) Wa[i] = 0;
    // (the point's Jacobi scales are requested in front of the sweep that hides their round trip, not behind it)
    double sp[3] = {1.0, 1.0, 1.0};
    if (!first && pvalid) { sp[0] = w.scale_p[3 * (size_t)p]; sp[1] = w.scale_p[3 * (size_t)p + 1]; sp[2] = w.scale_p[3 * (size_t)p + 2]; }
    // Jacobians of this lane's first observation for the Y sweep (tracks > LPP recompute): F and E, or -- compressed
    // factors -- the 2 x 3 d r / d (R X + t) instead of F
    double cF[CY ? 6 : 2 * BD], cE[6];
    // compressed factors (round 3): the 2 x 3 d r / d (R X + t) of the lane's first NPF observations is kept for the Y sweep,
    // which then needs no second evaluation (E = Jw R, a = R X from the camera table); the sweeps are unrolled over those
    // slices so that the kept values sit in registers; longer tracks re-evaluate the rest
    constexpr bool KEEPJ = CY;
    double cJ[KEEPJ ? NPF : 1][6];
    auto sweep1 = [&](const int pass, const int o, double* keepJ) __attribute__((always_inline)) {
      const bool head = pass == 0;
      const int c = f_pf.cam(pass, pb.obs_cam, o);
      const float2 uv = f_pf.uv(pass, pb.obs_uv, o);
      const int a = d.shared ? 0 : c;
      double r[2], F[2 * BD], E[6], Jw[6];
      if (LDSCAM)
        eval_full<KD>(d, lq + 9 * c, lt + 3 * c, pb.intr + 4 * a, X, uv, (unsigned)lfl[c],
                      pb.intr_const ? pb.intr_const[a] != 0 : false, pt_c, r, F, E, CY ? Jw : nullptr);
      else
        eval_full<KD>(d, CamR(pb.cam_q + 4 * c).R, pb.cam_t + 3 * c, pb.intr + 4 * a, X, uv,
                      pb.cam_const ? pb.cam_const[c] : 0u, pb.intr_const ? pb.intr_const[a] != 0 : false, pt_c, r, F, E,
                      CY ? Jw : nullptr);
      if (CACHEJ && !KEEPJ && head) {
#pragma unroll
        for (int i = 0; i < (CY ? 6 : 2 * BD); ++i) cF[i] = CY ? Jw[i] : F[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) cE[i] = E[i];
      }
      if (KEEPJ && keepJ) {
#pragma unroll
        for (int i = 0; i < 6; ++i) keepJ[i] = Jw[i];
      }
      V[0] += E[0] * E[0] + E[3] * E[3]; V[1] += E[0] * E[1] + E[3] * E[4]; V[2] += E[0] * E[2] + E[3] * E[5];
      V[3] += E[1] * E[1] + E[4] * E[4]; V[4] += E[1] * E[2] + E[4] * E[5]; V[5] += E[2] * E[2] + E[5] * E[5];
      g[0] += E[0] * r[0] + E[3] * r[1]; g[1] += E[1] * r[0] + E[4] * r[1]; g[2] += E[2] * r[0] + E[5] * r[1];
      if (KD > 0 && kdsh) {
#pragma unroll
        for (int m = 0; m < KD; ++m)
#pragma unroll
          for (int b = 0; b < 3; ++b) Wa[m * 3 + b] += F[6 + m] * E[b] + F[BD + 6 + m] * E[3 + b];
      }
    };
    if constexpr (KEEPJ) {
#pragma unroll
      for (int ps = 0; ps < NPF; ++ps) {
        const int o = o0 + sl + ps * LPP;
        if (o < o1) sweep1(ps, o, cJ[ps]);
      }
      for (int o = o0 + sl + NPF * LPP; o < o1; o += LPP) sweep1((o - o0) / LPP, o, nullptr);
    } else {
      for (int o = o0 + sl; o < o1; o += LPP) sweep1((o - o0) / LPP, o, nullptr);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) V[i] = group_sum<LPP>(V[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) g[i] = group_sum<LPP>(g[i]);
    if (KD > 0 && kdsh) {
#pragma unroll
      for (int i = 0; i < 3 * KD; ++i) Wa[i] = group_sum<LPP>(Wa[i]);
    }
    // every lane of the point now holds the totals; its lane 0 writes
    double s[3];
    const double colsq[3] = {V[0], V[3], V[5]};
    if (first) {
#pragma unroll
      for (int k = 0; k < 3; ++k) s[k] = opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(colsq[k])) : 1.0;
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) s[k] = sp[k];
    }
    double Gm[6] = {0, 0, 0, 0, 0, 0}, hs[3] = {0, 0, 0}, pd[3] = {0, 0, 0};
    double Ms[3 * (KD ? KD : 1)];
#pragma unroll
    for (int i = 0; i < 3 * (KD ? KD : 1); ++i) Ms[i] = 0;
    bool z_written = false;                           // tile_rhs: Z = [z | y_0 | y_1] with hs = G z, Ms_m = G y_m
    if (!pt_c && o1 > o0) {
      double dd[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) dd[k] = fmin(fmax(colsq[k] * s[k] * s[k], opt.min_lm_diagonal), opt.max_lm_diagonal) / radius;
#pragma unroll
      for (int k = 0; k < 3; ++k) pd[k] = dd[k] / (s[k] * s[k]);
      const double a00 = V[0] * s[0] * s[0] + dd[0], a10 = V[1] * s[0] * s[1], a20 = V[2] * s[0] * s[2];
      const double a11 = V[3] * s[1] * s[1] + dd[1], a21 = V[4] * s[1] * s[2], a22 = V[5] * s[2] * s[2] + dd[2];
      bool ok = a00 > 0;
      const double l00 = sqrt(a00);
      const double l10 = a10 / l00, l20 = a20 / l00;
      const double d11 = a11 - l10 * l10;
      ok = ok && d11 > 0;
      const double l11 = sqrt(d11);
      const double l21 = (a21 - l20 * l10) / l11;
      const double d22 = a22 - l20 * l20 - l21 * l21;
      ok = ok && d22 > 0;
      const double l22 = sqrt(d22);
      if (!ok) { if (sl == 0) ctl->linear_fail = 1; }
      // Linv (lower): i00 i10 i11 i20 i21 i22
      // (measured, round 4: reciprocal square roots + Newton steps instead of sqrt and the six IEEE divisions -- a third of
      //  the dependent instructions of this per-point chain -- change nothing in the launch time; the oracle's arithmetic stays)
      const double i00 = 1 / l00, i11 = 1 / l11, i22 = 1 / l22;
      const double i10 = -l10 * i00 * i11;
      const double i21 = -l21 * i11 * i22;
      const double i20 = -(l20 * i00 + l21 * i10) * i22;
      // G = S_p * Linv^T (upper triangular): G[b][m], b <= m   stored as G00 G01 G02 G11 G12 G22
      Gm[0] = s[0] * i00; Gm[1] = s[0] * i10; Gm[2] = s[0] * i20; Gm[3] = s[1] * i11; Gm[4] = s[1] * i21; Gm[5] = s[2] * i22;
      // z = Linv * (s o g) ; hs = G z
      const double gs0 = s[0] * g[0], gs1 = s[1] * g[1], gs2 = s[2] * g[2];
      const double z0 = i00 * gs0, z1 = i10 * gs0 + i11 * gs1, z2 = i20 * gs0 + i21 * gs1 + i22 * gs2;
      hs[0] = Gm[0] * z0 + Gm[1] * z1 + Gm[2] * z2; hs[1] = Gm[3] * z1 + Gm[4] * z2; hs[2] = Gm[5] * z2;
      if (trhs && sl == 0) {
        double* Z = w.Zp + 9 * (size_t)p;
        Z[0] = z0; Z[1] = z1; Z[2] = z2;
        if (KD == 0) { Z[3] = 0; Z[4] = 0; Z[5] = 0; Z[6] = 0; Z[7] = 0; Z[8] = 0; }   // (KD > 0, per-camera intrinsics: the full-factor tiles read z only)
        if (KD == 1 && kdsh) { Z[6] = 0; Z[7] = 0; Z[8] = 0; }
      }
      z_written = true;
      if (KD > 0 && kdsh) {
#pragma unroll
        for (int m = 0; m < KD; ++m) {
          const double sa = w.scale_c[6 * d.C + m];
          // W_a (scaled) row m: sa * Wa[m][b] * s[b];  Ms[:,m] = G G^T (S_p^-1 ...)  -> S_p V^-1 W_a^T = G (Linv (s o Wa^T)) * sa
          const double w0 = sa * s[0] * Wa[m * 3], w1 = sa * s[1] * Wa[m * 3 + 1], w2 = sa * s[2] * Wa[m * 3 + 2];
          const double y0 = i00 * w0, y1 = i10 * w0 + i11 * w1, y2 = i20 * w0 + i21 * w1 + i22 * w2;
          Ms[m * 3] = Gm[0] * y0 + Gm[1] * y1 + Gm[2] * y2; Ms[m * 3 + 1] = Gm[3] * y1 + Gm[4] * y2; Ms[m * 3 + 2] = Gm[5] * y2;
          if (trhs && sl == 0) { double* Z = w.Zp + 9 * (size_t)p + 3 + 3 * m; Z[0] = y0; Z[1] = y1; Z[2] = y2; }
        }
        if (trhs && sl == 0) {
          double* q = qs[wave][sub];
#pragma unroll
          for (int i = 0; i < KD; ++i) {
            q[i] += Wa[i * 3] * hs[0] + Wa[i * 3 + 1] * hs[1] + Wa[i * 3 + 2] * hs[2];
#pragma unroll
            for (int j = 0; j < KD; ++j)
              q[KD + i * KD + j] += Wa[i * 3] * Ms[j * 3] + Wa[i * 3 + 1] * Ms[j * 3 + 1] + Wa[i * 3 + 2] * Ms[j * 3 + 2];
          }
        }
      }
      gmax = fmax(gmax, fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2]))));
    }
    // per-observation Schur factors Y_i = s_c o ((F_i^T E_i) G) -> slot obs_slot[o] of the zero-padded
    // segment buffer consumed by schur_tile_kernel
    if (VGG_PP_ABLATE != 2) {
      const int bdt = d.shared ? 6 : BD;          // rows of the tile block (intrinsics only when per camera)
      auto emit = [&](const int pass, const int o, const double* keptJ) __attribute__((always_inline)) {
        const bool head = pass == 0;
        const int c = f_pf.cam(pass, pb.obs_cam, o);
        double F[CY ? 6 : 2 * BD], E[6];          // (CY: F holds the 2 x 3 Jw)
        if (KEEPJ && keptJ) {                     // kept from the first sweep: E = Jw R as obs_eval_R forms it
          double Rl[9];
          const double* Rc = Rl;
          if (LDSCAM) Rc = lq + 9 * c; else quat_to_R(pb.cam_q + 4 * c, Rl);
#pragma unroll
          for (int i = 0; i < 6; ++i) F[i] = keptJ[i];
#pragma unroll
          for (int row = 0; row < 2; ++row)
#pragma unroll
            for (int k = 0; k < 3; ++k)
              E[3 * row + k] = pt_c ? 0.0 : keptJ[3 * row] * Rc[k] + keptJ[3 * row + 1] * Rc[3 + k] + keptJ[3 * row + 2] * Rc[6 + k];
        } else if (CACHEJ && !KEEPJ && head) {    // cached Jacobians of the first slice
#pragma unroll
          for (int i = 0; i < (CY ? 6 : 2 * BD); ++i) F[i] = cF[i];
#pragma unroll
          for (int i = 0; i < 6; ++i) E[i] = cE[i];
        } else {
          const int a = d.shared ? 0 : c;
          double r[2], Ff[2 * BD];
          const float2 uv = f_pf.uv(pass, pb.obs_uv, o);
          if (LDSCAM)
            eval_full<KD>(d, lq + 9 * c, lt + 3 * c, pb.intr + 4 * a, X, uv, (unsigned)lfl[c],
                          pb.intr_const ? pb.intr_const[a] != 0 : false, pt_c, r, CY ? Ff : F, E, CY ? F : nullptr);
          else
            eval_full<KD>(d, CamR(pb.cam_q + 4 * c).R, pb.cam_t + 3 * c, pb.intr + 4 * a, X, uv,
                          pb.cam_const ? pb.cam_const[c] : 0u, pb.intr_const ? pb.intr_const[a] != 0 : false, pt_c, r,
                          CY ? Ff : F, E, CY ? F : nullptr);
        }
        const int slot = f_pf.slot(pass, pb.obs_slot, o);
        if constexpr (CY) {
          // compressed factor: N = Jw^T (E G) (3 x 3, row-major) and 2 a = 2 R X -- one run of 96 bytes per observation
          double M[6];
          M[0] = E[0] * Gm[0]; M[1] = E[0] * Gm[1] + E[1] * Gm[3]; M[2] = E[0] * Gm[2] + E[1] * Gm[4] + E[2] * Gm[5];
          M[3] = E[3] * Gm[0]; M[4] = E[3] * Gm[1] + E[4] * Gm[3]; M[5] = E[3] * Gm[2] + E[4] * Gm[4] + E[5] * Gm[5];
          double rec[kYc];
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int m = 0; m < 3; ++m) rec[3 * i + m] = F[i] * M[m] + F[3 + i] * M[3 + m];
          double Rl[9];
          const double* Rc = Rl;
          if (LDSCAM) Rc = lq + 9 * c; else quat_to_R(pb.cam_q + 4 * c, Rl);
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const double ai = Rc[3 * i] * X[0] + Rc[3 * i + 1] * X[1] + Rc[3 * i + 2] * X[2];
            rec[9 + i] = ai + ai;
          }
          // (into the wavefront's LDS image, row = lane; flush_records writes it out)
          slotstage[wave][lane] = slot;
#pragma unroll
          for (int i = 0; i < kYc / 2; ++i) ystage[wave][lane * (kYc / 2) + i] = make_double2(rec[2 * i], rec[2 * i + 1]);
        } else {
        // segment layout: [component 0..2][slot 0..15][row 0..bdt-1]  (three rows of the K dimension)
        const int rt = kGroup * bdt;
        double* y = w.Y + (size_t)(slot >> 4) * (3 * rt) + (slot & 15) * bdt;
        // the lane's three runs of bdt doubles go out 16 bytes at a time when bdt is even (the runs are then 16-byte
        // aligned): half the store instructions and half the partial-line transactions of 8-byte stores.
        const bool pairs = (bdt & 1) == 0;
        double prev[3] = {0, 0, 0};
#pragma unroll
        for (int i = 0; i < BD; ++i) {
          if (i < bdt) {
            const double sc = (i < 6) ? (LDSCAM ? lsc[6 * c + i] : w.scale_c[6 * c + i])
                                      : w.scale_c[6 * d.C + KD * c + (i - 6)];
            const double w0 = F[i] * E[0] + F[BD + i] * E[3], w1 = F[i] * E[1] + F[BD + i] * E[4],
                         w2 = F[i] * E[2] + F[BD + i] * E[5];
            const double y0 = sc * (w0 * Gm[0]), y1 = sc * (w0 * Gm[1] + w1 * Gm[3]), y2 = sc * (w0 * Gm[2] + w1 * Gm[4] + w2 * Gm[5]);
#if VGG_PP_ABLATE == 1                            // profiling build: the Y arithmetic without its stores
            if (Gm[0] == 12345.678) {
#endif
            if (!pairs) { y[i] = y0; y[rt + i] = y1; y[2 * rt + i] = y2; }
            else if (i & 1) {
              *reinterpret_cast<double2*>(y + i - 1) = make_double2(prev[0], y0);
              *reinterpret_cast<double2*>(y + rt + i - 1) = make_double2(prev[1], y1);
              *reinterpret_cast<double2*>(y + 2 * rt + i - 1) = make_double2(prev[2], y2);
            } else { prev[0] = y0; prev[1] = y1; prev[2] = y2; }
#if VGG_PP_ABLATE == 1
            }
#endif
          }
        }
        }
      };
      if constexpr (KEEPJ) {
        // A sweep's records leave the wavefront through LDS: lane q of store k carries piece (k 64 + q) % 6 of record
        // (k 64 + q) / 6, so six adjacent lanes cover one 96-byte record and a store instruction touches ~16 cache-line
        // sectors instead of 64.  (Lane = record, six 16-byte stores per lane: every store was its own L2 transaction,
        // 30 M per launch at configs[2].  Ablations: the launch without these stores 0.152 ms, without the whole sweep
        // 0.117, with them 0.249; this form 0.244 against 0.262 on the same box, profiles/r04_ab_point_pass_records.jsonl --
        // what is left of the stores' cost is their 0.5 GB at the HBM write rate.)  In-order LDS per wavefront: no barrier.
        auto flush_records = [&]() __attribute__((always_inline)) {
#if VGG_PP_ABLATE != 1                            // (profiling build 1: the Y arithmetic without its stores)
#pragma unroll
          for (int k = 0; k < kYc / 2; ++k) {
            const int q = k * 64 + lane;
            const int r = (q * 10923) >> 16;        // q / 6 for q < 384
            const int pc = q - 6 * r;
            const int sdst = slotstage[wave][r];
            const double2 v = ystage[wave][q];
            if (sdst >= 0) reinterpret_cast<double2*>(w.Y)[(size_t)sdst * (kYc / 2) + pc] = v;
          }
#endif
        };
#pragma unroll
        for (int ps = 0; ps < NPF; ++ps) {
          const int o = o0 + sl + ps * LPP;
          if (!__any(o < o1)) break;
          if (o < o1) emit(ps, o, cJ[ps]); else slotstage[wave][lane] = -1;
          flush_records();
        }
        for (int pass = NPF; __any(o0 + sl + pass * LPP < o1); ++pass) {
          const int o = o0 + sl + pass * LPP;
          if (o < o1) emit(pass, o, nullptr); else slotstage[wave][lane] = -1;
          flush_records();
        }
      } else {
        for (int o = o0 + sl; o < o1; o += LPP) emit((o - o0) / LPP, o, nullptr);
      }
    }
    if (sl == 0 && pvalid) {
      if (first) { w.scale_p[3 * p] = s[0]; w.scale_p[3 * p + 1] = s[1]; w.scale_p[3 * p + 2] = s[2]; }
#pragma unroll
      for (int i = 0; i < 6; ++i) w.G[6 * (size_t)p + i] = Gm[i];
#pragma unroll
      for (int i = 0; i < 3; ++i) { w.hs[3 * (size_t)p + i] = hs[i]; w.pdamp[3 * (size_t)p + i] = pd[i]; }
      if (KD > 0 && kdsh) {
#pragma unroll
        for (int i = 0; i < 3 * KD; ++i) w.Ms[(size_t)p * 3 * kdsh + i] = Ms[i];
      }
      if (trhs && !z_written) {                       // constant / unobserved point: no step, no contribution
#pragma unroll
        for (int i = 0; i < 9; ++i) w.Zp[9 * (size_t)p + i] = 0.0;
      }
    }
  }
  gmax = wave_max(gmax);
  if (lane == 0) wmax[wave] = gmax;
  __syncthreads();
  if (threadIdx.x == 0) w.part_B[blockIdx.x] = fmax(fmax(wmax[0], wmax[1]), fmax(wmax[2], wmax[3]));
  if (trhs && KD > 0) {
    __syncthreads();                                  // (every slot's owner lane has finished its points)
    if (threadIdx.x < KQ) {
      double v = 0.0;                                 // fixed order: wavefront, then point slot
      for (int wv = 0; wv < 4; ++wv)
        for (int sb = 0; sb < 64 / LPP; ++sb) v += qs[wv][sb][threadIdx.x];
      w.part_Q[8 * blockIdx.x + threadIdx.x] = v;
    }
  }
}

// start-of-iteration checks (Ceres FinalizeIterationAndCheckIfMinimizerCanContinue)
__device__ __forceinline__ void begin_iteration(const Ws& w, const vgg_ba_options& opt) {
  Ctl* c = w.ctl;
  if (c->need_lin) {
    c->gmax = fmax(c->gmax_cams, w.gmax_pts[0]);
    w.log[c->iteration].gradient_max_norm = c->gmax;
    if (c->iteration == 0) {
      vgg_ba_iteration z;
      z.iteration = 0; z.successful = 1; z.cost = c->x_cost; z.cost_change = 0; z.gradient_max_norm = c->gmax;
      z.step_norm = 0; z.relative_decrease = 0; z.radius = c->radius;
      w.log[0] = z;
    }
    if (c->gmax <= opt.gradient_tolerance) { c->done = 1; c->termination = 1; return; }
  }
  if (c->iteration >= opt.max_num_iterations) { c->done = 1; c->termination = 0; return; }
  if (c->radius <= opt.min_trust_region_radius) { c->done = 1; c->termination = 4; return; }
  c->scale_ready = 1;
  c->need_lin = 0;
  c->iteration += 1;
}


// one workgroup: thread 0 runs the checks; then the constant / unobserved columns of the reduced system get a unit
// diagonal and a zero right-hand side (their Jacobian columns are zero)
__global__ __launch_bounds__(256) void begin_iteration_kernel(Ws w, vgg_ba_options opt, int n, int32_t* chol_flags, int chol_flag_count) {
  if (w.ctl->done) return;
  // the hand-off flags of the factorisation that follows (chol.hip, cholesky_dataflow_flags): cleared here, not by a fill launch
  for (int j = threadIdx.x; j < chol_flag_count; j += 256) chol_flags[j] = 0;
  if (threadIdx.x == 0) begin_iteration(w, opt);
  for (int j = threadIdx.x; j < n; j += 256) {
    if (w.active[j]) continue;
    w.S[(size_t)j * n + j] = 1.0;
    w.rhs[j] = 0.0;
  }
}

// ---------------------------------------------------------------------------------------------
// Block-sparse Schur complement on the matrix cores.
// Tile (gI,gJ) of the reduced system is the R x R matrix (R = 16 cameras x BD rows)
//     C = sum over entries  YA_e (R x 3) * YB_e^T (3 x R),
// where YA_e / YB_e are the two *segments* of the entry: the per-observation factors
// Y_i = s_c o ((F_i^T E_i) G_p) of one point inside one camera group, stored by point_pass at the slot of
// their camera in a zero-padded block of 16 slots (cameras of the group that do not see the point stay
// zero for the whole solve).  A segment is therefore one contiguous, 16-byte aligned run of 16*BD*3 doubles:
// staging is a pure linear copy global -> LDS (dwordx4, coalesced, no masks, no index arithmetic), double
// buffered against the MFMAs.  Four entries are packed along K (4 x 3 = 12 = three K=4 steps of
// v_mfma_f64_16x16x4_f64).  The 4 wavefronts form a 2x2 grid over the NT x NT sub-tiles, each keeps up to
// NH x NH accumulators in registers for the whole chunk; all MFMAs are unconditional (scalar loop bounds).
typedef double f64x4_t __attribute__((ext_vector_type(4)));

// value of the lane with the lowest bit of its index flipped (DPP quad_perm [1,0,3,2]: no LDS, no index register)
__device__ __forceinline__ double dpp_swap_xor1(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

#ifndef VGG_TILE_TRACE
#define VGG_TILE_TRACE 0             // debug builds: per-wavefront phase cycle sums of the off-diagonal schur_tile launch
#endif
#if VGG_TILE_TRACE
__device__ long long g_tile_trace[2048 * 4 * 8];   // [workgroup][wave][batches, fetch+issue, matrix phase, LDS write phase, barrier wait, total]
#endif
#ifndef VGG_TILE_PRIO
#define VGG_TILE_PRIO 2              // wave priority: 2 = raised outside the matrix phase (staging, LDS writes, barrier): a wavefront
#endif                               // gets back to its matrix instructions sooner (round 3: off-diagonal launch 0.620 -> 0.606 ms); 1 = raised inside
#ifndef VGG_DIAG_PARITY
#define VGG_DIAG_PARITY 1           // diagonal tiles: sub-tiles dealt to the wavefronts by parity class (0: every fourth, round 3)
#endif
#ifndef VGG_TILE_EARLY_WRITE
#define VGG_TILE_EARLY_WRITE 1      // pipelined off-diagonal step: the LDS writes of batch b + 1 in front of batch b's matrix instructions (0: behind them)
#endif
#ifndef VGG_TILE_PIPE
#define VGG_TILE_PIPE 1             // compressed off-diagonal tiles: the last K step of a batch behind its barrier (round 5; 0: plain order)
#endif
#ifndef VGG_NO_SKIP
#define VGG_NO_SKIP 0               // profiling builds: 1 = every sub-tile of every batch runs (no presence skipping)
#endif
// bits of a segment's 16-slot presence mask whose cameras have rows in the 16-row block `blk` of a tile with BD rows per
// camera (0 for a block past the last one)
template <int BD>
__device__ __forceinline__ uint32_t block_slot_bits(int blk) {
  if (blk * 16 >= kGroup * BD) return 0u;
  const int s0 = (16 * blk) / BD, s1 = min(kGroup - 1, (16 * blk + 15) / BD);
  return ((2u << s1) - 1u) & ~((1u << s0) - 1u);
}

template <int BD, bool DIAG>
__device__ __forceinline__ void schur_tile_body(const Ws& w, const int32_t* __restrict__ chunk_desc,
                                                const int32_t* __restrict__ entries, int chunk, int zero_seg,
                                                double* __restrict__ ops) {
  constexpr int YS = BD * 3;                      // doubles per Y block
  constexpr int SEG = kGroup * YS;                // doubles per segment (16 slots)
  constexpr int R = kGroup * BD;                  // rows / cols of the tile
  constexpr int NT = R / 16;                      // 16x16 sub-tiles per side
  constexpr int NH = (NT + 1) / 2;                // sub-tile rows (cols) of one wavefront
  // [buffer][side][entry of the batch][component c][tile row]: a staged segment is a linear copy of the global
  // one, except that the tile rows of the odd entries are XOR-swizzled by 16 (double2 index ^ 8) when the entry
  // stride is a multiple of 256 bytes.  K step ks of the MFMAs takes component c = ks of the four entries
  // (lane group lk = entry), so the four 128-byte runs one operand fetch touches fall two and two into the
  // two halves of the LDS banks (conflict-free ds_read_b64).
  constexpr int SWZ = ((SEG * 8) % 256 == 0) ? 16 : 0;
  constexpr int SIDES = DIAG ? 1 : 2;             // LDS image: ops[buffer][side][entry of the batch][SEG]
  // BD = 6: the global segments are COMPRESSED (kYc = 12 doubles per slot: N row-major, 2 a; y_slot_doubles) and the six
  // rows of a slot -- [2 a]x N on top of N -- are rebuilt on the way into LDS.  The LDS image and everything behind it are
  // those of the full factors (without the Jacobi scales: tile_reduce_kernel applies them).
  constexpr bool CY = (BD == 6);
  constexpr int CSEG = kGroup * kYc;              // doubles per compressed segment
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: MFMAs only behind scalar control flow
  // chunk = the j-th of the J workgroups of tile (gI,gJ).  The tile's entry list [e0,e1) is sorted by point;
  // workgroup j takes the sub-chunks j, j+J, j+2J, ... of kSub entries, so all workgroups of all tiles sweep
  // the point range at the same relative rate and the (up to ~G) re-reads of one point's segments by
  // different tiles fall close together in time (they hit the L2 / Infinity Cache instead of HBM).
  const int e0 = chunk_desc[6 * chunk + 2], e1 = chunk_desc[6 * chunk + 3];
  const int cj = chunk_desc[6 * chunk + 4], cJ = chunk_desc[6 * chunk + 5];
  constexpr int BPS = kSub / 4;                   // batches of 4 entries per sub-chunk
  const int nsub = (e1 - e0 + kSub - 1) / kSub;
  const int nb = ((nsub - cj + cJ - 1) / cJ) * BPS;                          // batches of this workgroup
  auto ebase = [&](int b) -> int { return e0 + ((b / BPS) * cJ + cj) * kSub + (b % BPS) * 4; };
  f64x4_t acc[NH][NH];
#pragma unroll
  for (int i = 0; i < NH; ++i)
#pragma unroll
    for (int j = 0; j < NH; ++j) acc[i][j] = (f64x4_t){0.0, 0.0, 0.0, 0.0};

  // Staging of a batch = linear copy of its 8 (diagonal tile: 4) segments, 32 (64) threads per segment.  It is split so that
  // no wavefront ever waits on a dependent global load: the segment index of batch n+2 is loaded while
  // the segment data of batch n+1 is in flight, and that data is written to LDS only after the MFMAs of
  // batch n (async-stage split).
  // Compressed: a PAIR of lanes per slot, three double2 each (the even lane rows 0, 1 of N, the odd lane row 2 and 2 a);
  // the halves are exchanged with one DPP quad permute per register, the even lane then writes the rows 3..5 of the slot
  // (N itself), the odd lane the rows 0..2 (the cross products).  A diagonal tile stages 4 segments: waves 2, 3 idle here.
  constexpr int V = SEG / 2;                      // double2 per segment (full factors)
  constexpr int TPS = CY ? 32 : (DIAG ? 64 : 32); // threads per segment
  constexpr int NV = CY ? 3 : (V + TPS - 1) / TPS; // double2 per thread per batch
  const int sseg = tid / TPS, l32 = tid % TPS;
  const int se = DIAG ? (sseg & 3) : (sseg >> 1), sside = DIAG ? 0 : (sseg & 1);
  // tile_rhs (compressed diagonal tiles): every observation sits in exactly one segment and every segment is exactly one
  // diagonal entry, so  sum over the entries of a diagonal tile of  Y_seg (96 x 3) Z_p (3 x 3)  is, per camera, what
  // cam_pass<RHS> summed per observation: F^T E hs (column 0: Z = [z | y_0 | y_1], E hs = E G z) and F^T E Ms_m (columns
  // 1, 2) -- from operands that are in LDS anyway.  The wavefronts 2, 3 have no segment to stage here (four segments, 32
  // lanes each): they are the HELPERS -- they fetch the Z of the batch's four points through the same two-stage prefetch
  // (entry -> point index, point index -> Z: field 0 of an entry is the point) into a 96-double LDS image per buffer and,
  // while the other two wavefronts write the next batch's segments to LDS, add the products of their row (a lane per tile
  // row) for the four entries of the batch that has just been multiplied.  The two roles are two instantiations of the loop
  // (`run` below): the stagers carry the staging registers, the helpers the three sums, both within the 128 registers of four
  // wavefronts per SIMD.  Scales and constant-parameter masks are applied by assemble_kernel, like tile_reduce_kernel does
  // for the tile sums.
  constexpr bool TR = CY && DIAG;
  constexpr bool INTERLEAVE = VGG_TILE_INTERLEAVE && !CY;
  const bool trhs = TR && w.tile_rhs != 0;
  // tile_rhs with FULL factors (7 x 7 / 8 x 8 blocks: per-camera intrinsics, round 6).  All four wavefronts stage here, so
  // there are no helpers: once a batch is in LDS, thread t adds the products of tile row t % R for two of the batch's four
  // entries (t / R picks the pair) -- six LDS reads and six multiply-adds per batch beside 48 matrix instructions per wavefront.
  // Only z is needed (column 0 of Z: without a shared camera the intrinsics border has no per-point terms), and the factors
  // carry their Jacobi scales and constant-parameter masks: the sums go into the right-hand side as they are.  The 12
  // doubles of a batch's four z come in through threads 0..11 with the staging's own two-stage prefetch.
  constexpr bool TRF = !CY && DIAG;
  const bool trf = TRF && w.tile_rhs != 0;
  constexpr int ZS = 12;                              // doubles per entry of the Z image (9 used; 16-byte aligned rows)
  double* zs = ops + 2 * (DIAG ? 1 : 2) * 4 * SEG;    // [buffer][entry][col][k]
  auto run = [&](auto role) __attribute__((always_inline)) {
  constexpr bool HELPER = decltype(role)::value;      // (only with TR: wavefronts 2, 3)
  const bool zthread = HELPER && trhs && l32 < 9;
  double racc[3] = {0.0, 0.0, 0.0};
  // The segment index of a batch is loaded unconditionally (clamped entry) and only CONSUMED one iteration later;
  // entries past the end of the list are redirected to the all-zero segment behind the last real one.  Nothing in
  // the current iteration depends on the loaded value, so no s_waitcnt sits between the prefetch and the MFMAs.
  const int32_t* seg_field = entries + (HELPER ? 0 : 1 + sside);   // entries[e] = (point, segA, segB, masks)
  auto load_seg_index = [&](int eb) -> int { return seg_field[4 * (size_t)min(eb + se, e1 - 1)]; };
  auto seg_valid = [&](int eb) -> bool { return eb + se < e1; };
  // two staging register sets: the loads of batch b + 2 are issued while batch b is multiplied and batch b + 1 (loaded an
  // iteration earlier) is written to LDS -- twice the bytes in flight per workgroup for NV more double2 registers
  constexpr int DEPTH = (CY && !DIAG) ? 3 : ((!CY && DIAG) ? VGG_DIAG_FULL_DEPTH : ((!CY && !DIAG) ? VGG_OFF_FULL_DEPTH : 2));   // staging register sets = batches in flight ahead of the one being multiplied
  double2 sv[DEPTH][NV];
  // (full-factor tile_rhs) lane l of EVERY wavefront requests component l & 3 of z of entry (l >> 2) & 3 (16 distinct addresses
  //  per wavefront, four of them padding), unconditionally and without a validity select: an entry past the end of the list
  //  multiplies the all-zero segment, so any finite z will do there (the clamped last entry's).  The first formulation had
  //  the twelve lanes of wavefront 0 do it inside `if (tid < 12)`: the loop-carried point index then went through a copy at
  //  the join of that divergent region, which the compiler guards with s_waitcnt vmcnt(0) right behind the load -- wavefront 0
  //  waited for every load in flight, every batch (rocprofv3, one configs[3] shard: diagonal launch 293 -> 345 us from the z
  //  loads alone).  Threads 0..15 (minus the padding lanes) write the image to LDS.
  constexpr bool ZLOAD = TRF && !(VGG_TRF_ABLATE & 2);
  const int zte = (lane >> 2) & 3, ztc = lane & 3;
  const bool zt = ZLOAD && trf && tid < 16 && ztc < 3;
  auto z_point = [&](int eb) -> int { return entries[4 * (size_t)min(eb + zte, e1 - 1)]; };
  double zv[DEPTH];
  int zpt_next = 0;
  double rfull = 0.0;
  // compressed staging: the lane's two LDS targets inside a staged segment (see write_lds)
  const bool cy_top = (l32 & 1) != 0;
  const int cy_sw = (se & 1) * SWZ, cy_r0 = 6 * (l32 >> 1);
  const int cy_row16 = (cy_top ? cy_r0 : cy_r0 + 4) ^ cy_sw, cy_row8 = (cy_top ? cy_r0 + 2 : cy_r0 + 3) ^ cy_sw;
  auto issue_loads = [&](double2 (&sv)[NV], int seg_index, bool valid) __attribute__((always_inline)) {
    if constexpr (CY) {
      if constexpr (!HELPER) {
        // (no `if (stager)` here, and no `if (zthread)` / validity select around the helpers' load below: a load inside a
        //  conditional region is one the compiler cannot count on, so every s_waitcnt vmcnt(n) behind it is computed as if
        //  it had not been issued -- the stagers' wait for the batch loaded a step earlier came out as vmcnt(1) with four
        //  younger loads in flight, i.e. it waited for most of THIS step's loads, every step.  The non-helper loop only
        //  runs on staging wavefronts; a helper lane >= 9 fetches a ninth Z component nobody reads; an entry past the end
        //  of the list multiplies the all-zero segment, so the clamped entry's finite Z does no harm.)
        const double2* src = reinterpret_cast<const double2*>(w.Y) + (size_t)(valid ? seg_index : zero_seg) * (CSEG / 2) + 3 * l32;
#pragma unroll
        for (int i = 0; i < NV; ++i) sv[i] = src[i];
      } else {
        sv[0].x = w.Zp[9 * (size_t)seg_index + min(l32, 8)];   // (seg_index = the entry's point here)
      }
    } else {
      const double2* src = reinterpret_cast<const double2*>(w.Y) + (size_t)(valid ? seg_index : zero_seg) * V;
#pragma unroll
      for (int i = 0; i < NV; ++i) { const int off = l32 + TPS * i; sv[i] = (off < V) ? src[off] : make_double2(0.0, 0.0); }
    }
  };
  // (third k of 3 of a batch's LDS image: one component of the compressed factors, a third of the double2 of the full ones)
  auto write_lds_k = [&](const double2 (&sv)[NV], int buf, const int k) __attribute__((always_inline)) {
    if constexpr (CY && HELPER) {
      if (k == 0 && zthread) zs[(buf * 4 + se) * ZS + l32] = sv[0].x;
    } else if constexpr (CY) {
      {
        // odd lane:  mine = (N20 N21 N22, 2a0 2a1 2a2), other = (N00 N01 N02, N10 N11 N12) -> rows 0..2 = (2 a) x N[:, k]:
        //            16 bytes at row 0, 8 at row 2
        // even lane: mine = (N00 N01 N02, N10 N11 N12), other = (N20 N21 N22, ...)         -> rows 3..5 = N[:, k]:
        //            8 bytes at row 3, 16 at row 4
        // (the odd entries' tile rows are XOR-swizzled by 16; k R is a multiple of 32, so the swizzle acts on the row alone
        //  and the component enters the LDS address as an immediate offset)
        double* dst = ops + (size_t)((buf * SIDES + sside) * 4 + se) * SEG;
        const double m[6] = {sv[0].x, sv[0].y, sv[1].x, sv[1].y, sv[2].x, sv[2].y};
#if VGG_ABLATE == 4                               // (profiling build: the staging without its exchange / arithmetic)
        *reinterpret_cast<double2*>(dst + k * R + cy_row16) = make_double2(m[3 + k], m[k]);
        dst[k * R + cy_row8] = m[k];
#else
        const double ok = dpp_swap_xor1(m[k]), o3k = dpp_swap_xor1(m[3 + k]);   // the other half, one component at a time
        const double t0 = m[4] * m[k] - m[5] * o3k, t1 = m[5] * ok - m[3] * m[k], t2 = m[3] * o3k - m[4] * ok;
        *reinterpret_cast<double2*>(dst + k * R + cy_row16) = cy_top ? make_double2(t0, t1) : make_double2(m[3 + k], ok);
        dst[k * R + cy_row8] = cy_top ? t2 : m[k];
#endif
      }
    } else {
      double2* dst = reinterpret_cast<double2*>(ops + (size_t)((buf * SIDES + sside) * 4 + se) * SEG);
      constexpr int PER3 = (NV + 2) / 3;
#pragma unroll
      for (int i = k * PER3; i < (k + 1) * PER3 && i < NV; ++i) {
        const int off = l32 + TPS * i;
        if (off < V) dst[off ^ ((se & 1) * (SWZ / 2))] = sv[i];
      }
    }
  };
  auto write_lds = [&](const double2 (&sv)[NV], int buf) __attribute__((always_inline)) {
    write_lds_k(sv, buf, 0); write_lds_k(sv, buf, 1); write_lds_k(sv, buf, 2);
  };
  // operand addressing: entry e = lane >> 4 of the batch, component c = ks, tile row rr = 16 rb + (lane & 15)
  const int li = lane & 15, lk = lane >> 4;
  // (ks enters the address as a compile-time immediate: ks R doubles)
  const int kbase = lk * SEG, swz = (lk & 1) * SWZ;

  // presence of a batch (quad): field 3 of its first entry, wave-uniform (scalar load), fetched one batch ahead
  auto load_quad_mask = [&](int eb) -> uint32_t { return (uint32_t)entries[4 * (size_t)min(eb, e1 - 1) + 3]; };
  // DEPTH staging register sets (batch n lives in set n % DEPTH): the loads of batch b + DEPTH are issued while batch b is
  // multiplied and batch b + 1, requested DEPTH - 1 steps earlier, is written to LDS (round 3, off-diagonal launch with
  // compressed segments: with two sets the LDS write phase spent most of its ~1000 cycles per batch waiting for loads
  // issued one step -- ~2 us -- before; the other variants have no registers to spare for a third set)
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) issue_loads(sv[u], load_seg_index(ebase(u)), seg_valid(ebase(u)));
  int seg_next = load_seg_index(ebase(DEPTH));
  bool valid_next = seg_valid(ebase(DEPTH));
  if constexpr (ZLOAD) {
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) zv[u] = w.Zp[9 * (size_t)z_point(ebase(u)) + ztc];
    zpt_next = z_point(ebase(DEPTH));
    if (zt) zs[(0 * 4 + zte) * ZS + ztc] = zv[0];
  }
  uint32_t qmask = load_quad_mask(ebase(0)), qmask_next = load_quad_mask(ebase(1));
  write_lds(sv[0], 0);
  __syncthreads();
  // software pipeline shared by the two sub-tile assignments below
  auto sweep = [&](auto&& mfma_batch) __attribute__((always_inline)) {
#if VGG_TILE_TRACE
    long long tr_issue = 0, tr_mfma = 0, tr_write = 0, tr_bar = 0;
    const long long tr_begin = __builtin_amdgcn_s_memtime(), tr_wall = (long long)wall_clock64();
#define VGG_TT(var) { const long long now_ = __builtin_amdgcn_s_memtime(); var += now_ - tr_last; tr_last = now_; }
#else
#define VGG_TT(var)
#endif
    auto step = [&](int b, int buf, double2 (&sv_load)[NV], const double2 (&sv_write)[NV], double& zv_load, const double& zv_write) __attribute__((always_inline)) {
#if VGG_TILE_TRACE
      long long tr_last = __builtin_amdgcn_s_memtime();
#endif
#if VGG_ABLATE != 2                               // (profiling builds only: 1 = no MFMA, 2 = no global loads)
      // (the index of batch b+DEPTH+1 is requested BEFORE the data of batch b+DEPTH: waiting for it next time round then
      //  leaves the younger data loads in flight -- vmcnt(3), not vmcnt(0))
      const int seg_after = load_seg_index(ebase(b + DEPTH + 1));
      // (full-factor tile_rhs: the point index and the z are requested IN FRONT of the segment data, like the segment index)
      if constexpr (ZLOAD) {
        const int zpt_after = z_point(ebase(b + DEPTH + 1));
        zv_load = w.Zp[9 * (size_t)zpt_next + ztc];
        zpt_next = zpt_after;
      }
      issue_loads(sv_load, seg_next, valid_next); // batch b+DEPTH (the zero segment past the end of the tile's list)
      seg_next = seg_after;
      valid_next = seg_valid(ebase(b + DEPTH + 1));
#endif
      VGG_TT(tr_issue)
#if VGG_ABLATE != 1
#if VGG_TILE_PRIO == 1                            // experiment: matrix phase at raised wave priority
      __builtin_amdgcn_s_setprio(2);
#elif VGG_TILE_PRIO == 2                          // experiment: staging phases at raised wave priority
      __builtin_amdgcn_s_setprio(0);
#endif
      auto rhs_rows = [&](int buf) __attribute__((always_inline)) {
        if (trf && tid < 2 * R && !(VGG_TRF_ABLATE & 1)) {
          const int rrow = tid % R, half = tid / R;
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int e = 2 * half + q;
            const double* zz = zs + (buf * 4 + e) * ZS;
            const double* op = ops + (size_t)(buf * 4 + e) * SEG + (rrow ^ ((e & 1) * SWZ));
            rfull += op[0] * zz[0] + op[R] * zz[1] + op[2 * R] * zz[2];
          }
        }
      };
      // INTERLEAVE (full factors, 7 x 7 / 8 x 8 blocks): the LDS image of batch b + 1 is written in three pieces BETWEEN
      // the K steps of batch b instead of in a phase of its own behind the last matrix instruction -- the ds_writes issue
      // while the matrix pipe works.  Same-box A/B (round 4): configs[3] whole, off-diagonal launch 13.66 -> 13.18 ms,
      // iteration 21.1 -> 20.6; with the compressed 6 x 6 factors (whose write phase is arithmetic: exchanges and cross
      // products on the pipe the matrix instructions use) nothing for the off-diagonal launch and +6 % for the diagonal one
      mfma_batch(buf, qmask, [&](const int k) __attribute__((always_inline)) {
#if VGG_ABLATE != 5
        if constexpr (INTERLEAVE) write_lds_k(sv_write, buf ^ 1, k);
        if constexpr (INTERLEAVE && TRF) { if (k == 0 && zt) zs[((buf ^ 1) * 4 + zte) * ZS + ztc] = zv_write; }
#endif
        // (full-factor tile_rhs: the row's products are read and added BETWEEN K steps 1 and 2, under the matrix instructions
        //  of the batch -- behind the last one they were a tail of twelve exposed LDS reads per wavefront in front of the barrier)
        if constexpr (TRF && VGG_TRF_EARLY) { if (k == VGG_TRF_EARLY - 1) rhs_rows(buf); }
      });
      if constexpr (TRF && !VGG_TRF_EARLY) rhs_rows(buf);
      if constexpr (TR && HELPER) {
        if (trhs) {
          const int rrow = tid & 127;                 // (wavefront 2: tile rows 0..63, wavefront 3: 64..95)
          if (rrow < R) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const double* zz = zs + (buf * 4 + e) * ZS;
              const double2* z2 = reinterpret_cast<const double2*>(zz);
              const double2 za = z2[0], zb = z2[1], zc = z2[2], zd = z2[3];   // z0 z1 | z2 y00 | y01 y02 | y10 y11
              const double ze = zz[8];                                          // y12
              const double* op = ops + (size_t)(buf * 4 + e) * SEG + (rrow ^ ((e & 1) * SWZ));
              const double v0 = op[0], v1 = op[R], v2 = op[2 * R];
              racc[0] += v0 * za.x + v1 * za.y + v2 * zb.x;
              racc[1] += v0 * zb.y + v1 * zc.x + v2 * zc.y;
              racc[2] += v0 * zd.x + v1 * zd.y + v2 * ze;
            }
          }
        }
      }
#if VGG_TILE_PRIO == 1
      __builtin_amdgcn_s_setprio(0);
#elif VGG_TILE_PRIO == 2
      __builtin_amdgcn_s_setprio(3);
#endif
#endif
      VGG_TT(tr_mfma)
      qmask = qmask_next;
      qmask_next = load_quad_mask(ebase(b + 2));
#if VGG_ABLATE != 5                               // (profiling build: 5 = no LDS writes)
      if constexpr (!INTERLEAVE) write_lds(sv_write, buf ^ 1);   // batch b+1, in flight for two steps
      if constexpr (!INTERLEAVE && TRF) { if (zt) zs[((buf ^ 1) * 4 + zte) * ZS + ztc] = zv_write; }
#endif
      VGG_TT(tr_write)
#if VGG_ABLATE != 3
      __syncthreads();
#endif
      VGG_TT(tr_bar)
    };
    // (2 DEPTH steps per trip: the LDS buffers alternate, the register sets rotate)
    constexpr int TRIP = 2 * DEPTH;
    int b = 0;
    for (; b + TRIP - 1 < nb; b += TRIP) {
#pragma unroll
      for (int u = 0; u < TRIP; ++u) step(b + u, u & 1, sv[u % DEPTH], sv[(u + 1) % DEPTH], zv[u % DEPTH], zv[(u + 1) % DEPTH]);
    }
#pragma unroll
    for (int u = 0; u < TRIP - 1; ++u)
      if (b + u < nb) step(b + u, u & 1, sv[u % DEPTH], sv[(u + 1) % DEPTH], zv[u % DEPTH], zv[(u + 1) % DEPTH]);
#if VGG_TILE_TRACE
    if (!DIAG && lane == 0 && blockIdx.x < 2048) {
      long long* t = g_tile_trace + ((size_t)blockIdx.x * 4 + wave) * 8;
      t[0] = nb; t[1] = tr_issue; t[2] = tr_mfma; t[3] = tr_write; t[4] = tr_bar; t[5] = __builtin_amdgcn_s_memtime() - tr_begin;
      t[6] = (long long)wall_clock64() - tr_wall;
    }
#endif
  };
  // partial tile of this chunk, row-major R x R (f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 reg);
  // tile_reduce_kernel sums the chunks of a tile in a fixed order (deterministic, no atomics)
  double* part = w.tile_part + (size_t)chunk * R * R;
  auto store_subtile = [&](int rb, int cb, const f64x4_t& v) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) part[(size_t)(16 * rb + lk + 4 * reg) * R + 16 * cb + li] = v[reg];
  };
  if constexpr (!DIAG) {
    // off-diagonal tile: 2x2 wave grid, wave (wr,wc) owns the sub-tiles (wr + 2 i, wc + 2 j): strided, so that a run of
    // absent cameras (a track that starts or ends inside the group) thins out all four wavefronts alike.  A sub-tile is
    // skipped for the batch when none of its four entries has a camera in its 16 rows or in its 16 columns (scalar
    // branches on the quad mask; a block index >= NT has no rows and never runs).
    const int wr = wave >> 1, wc = wave & 1;
    int rowoffA[NH], rowoffB[NH];
    uint32_t bitsA[NH], bitsB[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      rowoffA[i] = kbase + ((16 * min(wr + 2 * i, NT - 1) + li) ^ swz);
      rowoffB[i] = kbase + ((16 * min(wc + 2 * i, NT - 1) + li) ^ swz);
      bitsA[i] = block_slot_bits<BD>(wr + 2 * i);
      bitsB[i] = block_slot_bits<BD>(wc + 2 * i) << 16;
    }
#if VGG_TILE_PIPE && !VGG_ABLATE && !VGG_TILE_TRACE     // (ablation / trace builds take sweep(), which carries their hooks)
    if constexpr (CY) {
      // PIPELINED BARRIER (round 5, compressed 6 x 6 off-diagonal tiles): the LAST K step of a batch is issued BEHIND the
      // batch's barrier.  Per step:  [loads of b+DEPTH] [K steps 0, 1 of b; operands of K step 2 fetched] [LDS image of b+1]
      // [barrier] [operands of K step 0 of b+1 requested] [K step 2 of b].  The matrix instructions of K step 2 (their
      // operands sit in registers since before the barrier) cover the LDS round trip of the next batch's first operands,
      // which the plain order exposes on every wavefront right behind every barrier (profiles/r05_tile_diag_ablations.jsonl:
      // without LDS operand reads the launch is 8 % shorter).  Hazards: every read of buffer `buf` (K steps 0..2 of b) is
      // complete before barrier(b) -- the fetches are waited for by the `pin`s in front of it --, buffer buf is rewritten
      // (batch b + 2) in step b + 1's write phase, behind that barrier; buffer buf^1 (batch b + 1) is read behind barrier(b),
      // which follows its writes.  Same matrix instructions in the same order per accumulator: bit-identical sums.
      // Same-box A/B (profiles/r05_ab_tile_pipe_c3.jsonl): off-diagonal launch 0.578 -> 0.560 ms.  The same order for the
      // diagonal tiles (a second operand set at 128 registers) measured SLOWER (0.256 -> 0.269 ms), and so did the same order
      // for the 8 x 8 full-factor tiles with their interleaved LDS writes (configs[3] whole: 13.13 -> 13.35 ms,
      // profiles/r05_ab_tile_pipe_c4full.jsonl): neither is in the tree.
      double a[2][NH], bq[2][NH];
      auto fetch = [&](int set, int buf, int ks) __attribute__((always_inline)) {
        const double* As = ops + (size_t)(buf * SIDES) * 4 * SEG;
        const double* Bs = ops + (size_t)(buf * SIDES + 1) * 4 * SEG;
#pragma unroll
        for (int i = 0; i < NH; ++i) { a[set][i] = As[rowoffA[i] + ks * R]; bq[set][i] = Bs[rowoffB[i] + ks * R]; }
      };
      auto pin = [&](int set) __attribute__((always_inline)) {
        static_assert(NH == 3, "operand sets");
        asm volatile("" : "+v"(a[set][0]), "+v"(a[set][1]), "+v"(a[set][2]), "+v"(bq[set][0]), "+v"(bq[set][1]), "+v"(bq[set][2]));
      };
      uint32_t on = 0u;
      auto skip_bits = [&](uint32_t qm) -> uint32_t {
        uint32_t colm = 0u, o = 0u;
#pragma unroll
        for (int j = 0; j < NH; ++j) colm |= ((qm & bitsB[j]) != 0 ? 1u : 0u) << j;
#pragma unroll
        for (int i = 0; i < NH; ++i) o |= ((qm & bitsA[i]) != 0 ? colm : 0u) << (NH * i);
        return VGG_NO_SKIP ? 0xFFFFFFFFu : (uint32_t)__builtin_amdgcn_readfirstlane((int)o);
      };
      auto group = [&](int set) __attribute__((always_inline)) {
        uint32_t m = on;
        asm volatile("" : "+s"(m));
#pragma unroll
        for (int i = 0; i < NH; ++i)
#pragma unroll
          for (int j = 0; j < NH; ++j)
            if (m & (1u << (NH * i + j))) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[set][i], bq[set][j], acc[i][j], 0, 0, 0);
      };
      // step of batch b; P = the operand set that holds K step 0 of b on entry (alternates from step to step)
      auto pstep = [&](int b, int buf, const int P, double2 (&sv_load)[NV], const double2 (&sv_write)[NV]) __attribute__((always_inline)) {
        const int seg_after = load_seg_index(ebase(b + DEPTH + 1));
        issue_loads(sv_load, seg_next, valid_next);
        seg_next = seg_after;
        valid_next = seg_valid(ebase(b + DEPTH + 1));
#if VGG_TILE_EARLY_WRITE
        // the LDS image of batch b + 1 is written FIRST: buffer buf^1 is free since barrier(b - 1) -- its last readers fetched
        // K step 2 of batch b - 1 in front of that barrier -- so the writes have landed long before this step's barrier and
        // its lgkmcnt(0) costs nothing (same-box A/B profiles/r05_ab_tile_pipe_c3.jsonl: 0.531 -> 0.521 ms)
        write_lds(sv_write, buf ^ 1);
#endif
        __builtin_amdgcn_s_setprio(0);
        on = skip_bits(qmask);
        fetch(1 - P, buf, 1); pin(P);
        group(P);                                     // K step 0
        fetch(P, buf, 2); pin(1 - P);
        group(1 - P);                                 // K step 1
        pin(P);                                       // (the operands of K step 2 have landed: no read of `buf` is left)
        __builtin_amdgcn_s_setprio(3);
        qmask = qmask_next;
        qmask_next = load_quad_mask(ebase(b + 2));
#if !VGG_TILE_EARLY_WRITE
        write_lds(sv_write, buf ^ 1);
#endif
        __syncthreads();
        fetch(1 - P, buf ^ 1, 0);                     // K step 0 of batch b + 1 (stale bytes behind the last batch: never used)
        __builtin_amdgcn_s_setprio(0);
        pin(P);                                       // (keeps the matrix instructions below behind the barrier)
        group(P);                                     // K step 2 of batch b
      };
      fetch(0, 0, 0);
      constexpr int TRIP = 2 * DEPTH;
      int b = 0;
      for (; b + TRIP - 1 < nb; b += TRIP) {
#pragma unroll
        for (int u = 0; u < TRIP; ++u) pstep(b + u, u & 1, u & 1, sv[u % DEPTH], sv[(u + 1) % DEPTH]);
      }
#pragma unroll
      for (int u = 0; u < TRIP - 1; ++u)
        if (b + u < nb) pstep(b + u, u & 1, u & 1, sv[u % DEPTH], sv[(u + 1) % DEPTH]);
    } else
#endif
    sweep([&](int buf, uint32_t qm, auto&& wr) {
      const double* As = ops + (size_t)(buf * SIDES) * 4 * SEG;
      const double* Bs = ops + (size_t)(buf * SIDES + 1) * 4 * SEG;
      // one scalar bit per sub-tile of this wavefront (bit NH i + j), built once per batch: every skip test below is a bit
      // test + branch.  (With the conditions kept as booleans the compiler built each of the 27 tests of a batch from 64-bit
      // lane masks -- six dependent scalar instructions and two branches per matrix instruction.)
      uint32_t colm = 0u, on = 0u;
#pragma unroll
      for (int j = 0; j < NH; ++j) colm |= ((qm & bitsB[j]) != 0 ? 1u : 0u) << j;
#pragma unroll
      for (int i = 0; i < NH; ++i) on |= ((qm & bitsA[i]) != 0 ? colm : 0u) << (NH * i);
      on = VGG_NO_SKIP ? 0xFFFFFFFFu : (uint32_t)__builtin_amdgcn_readfirstlane((int)on);
      // operands of K step ks + 1 are requested before the matrix instructions of step ks; `pin` keeps the compiler from
      // sinking the LDS reads to their first use (it then waited for each one right in front of its matrix instruction),
      // `opaque` from turning the bit tests into lane-mask booleans shared by the three K steps
      double a[2][NH], bq[2][NH];
      auto fetch = [&](int set, int ks) __attribute__((always_inline)) {
#if VGG_ABLATE == 6                               // (profiling build: no LDS operand reads)
#pragma unroll
        for (int i = 0; i < NH; ++i) { a[set][i] = 1.0 + ks; bq[set][i] = 2.0 + i; }
#else
#pragma unroll
        for (int i = 0; i < NH; ++i) { a[set][i] = As[rowoffA[i] + ks * R]; bq[set][i] = Bs[rowoffB[i] + ks * R]; }
#endif
      };
      auto pin = [&](int set) __attribute__((always_inline)) {
        static_assert(NH == 3 || NH == 4, "operand sets");
        if constexpr (NH == 3)
          asm volatile("" : "+v"(a[set][0]), "+v"(a[set][1]), "+v"(a[set][2]), "+v"(bq[set][0]), "+v"(bq[set][1]), "+v"(bq[set][2]));
        else
          asm volatile("" : "+v"(a[set][0]), "+v"(a[set][1]), "+v"(a[set][2]), "+v"(a[set][3]), "+v"(bq[set][0]), "+v"(bq[set][1]),
                       "+v"(bq[set][2]), "+v"(bq[set][3]));
      };
      auto group = [&](int set) __attribute__((always_inline)) {
        uint32_t m = on;
        asm volatile("" : "+s"(m));
#pragma unroll
        for (int i = 0; i < NH; ++i)
#pragma unroll
          for (int j = 0; j < NH; ++j)
            if (m & (1u << (NH * i + j))) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[set][i], bq[set][j], acc[i][j], 0, 0, 0);
      };
      fetch(0, 0);
      fetch(1, 1); pin(0);
      group(0);
      wr(0);
      fetch(0, 2); pin(1);
      group(1);
      wr(1);
      pin(0);
      group(0);
      wr(2);
    });
#pragma unroll
    for (int i = 0; i < NH; ++i)
#pragma unroll
      for (int j = 0; j < NH; ++j)
        if (wr + 2 * i < NT && wc + 2 * j < NT) store_subtile(wr + 2 * i, wc + 2 * j, acc[i][j]);
  } else {
#if VGG_DIAG_PARITY
    // diagonal tile (A == B, symmetric): only the NT (NT + 1) / 2 sub-tiles of the lower triangle are computed.  Round 4: a
    // wavefront owns a PARITY CLASS of them -- rows wr + 2 i, columns wc + 2 j, row block >= column block -- like the
    // off-diagonal variant, instead of every fourth one of the row-by-row enumeration: its sub-tiles then share their row
    // and column operands (3..6 LDS reads per K step for 3..6 matrix instructions; dealt round-robin every matrix instruction
    // fetched its own two: 10..12 reads), and the classes still thin out alike under a run of absent cameras.  The classes
    // (0,0), (1,1), (0,1), (1,0) hold 6, 6, 3 and 6 sub-tiles at NT = 6.
    // With the stager / helper split of the compressed tiles (TR) the class follows the role, so that each of the two loop
    // bodies needs the accumulators i >= j only (six: the 128-register budget of four wavefronts per SIMD): stagers 0, 1
    // take the classes (0,0), (1,1) -- column blocks = row blocks, ONE operand set --, the helpers (0,1) -- three sub-tiles,
    // the wavefront with the 64-row share of the right-hand side -- and (1,0).
    const int mrole = wave;
    const int wr = (mrole == 1 || mrole == 3) ? 1 : 0, wc = (mrole == 1 || mrole == 2) ? 1 : 0;   // (0,0), (1,1), (0,1), (1,0)
    constexpr bool SAME = TR && !HELPER;              // compile-time: column operands are the row operands
    const bool same_rt = !TR && wr == wc;             // (full factors: one loop body for the four classes)
    int rowoffA[NH], rowoffB[NH];
    uint32_t bitsA[NH], bitsB[NH];
#pragma unroll
    for (int i = 0; i < NH; ++i) {
      rowoffA[i] = kbase + ((16 * min(wr + 2 * i, NT - 1) + li) ^ swz);
      rowoffB[i] = kbase + ((16 * min(wc + 2 * i, NT - 1) + li) ^ swz);
      bitsA[i] = block_slot_bits<BD>(wr + 2 * i);
      bitsB[i] = block_slot_bits<BD>(wc + 2 * i);
    }
    // bit NH i + j: sub-tile (wr + 2 i, wc + 2 j) exists and lies in the lower triangle (wave-uniform)
    uint32_t lower = 0u;
#pragma unroll
    for (int i = 0; i < NH; ++i)
#pragma unroll
      for (int j = 0; j < NH; ++j)
        if (wr + 2 * i < NT && wc + 2 * j < NT && wr + 2 * i >= wc + 2 * j) lower |= 1u << (NH * i + j);
    lower = (uint32_t)__builtin_amdgcn_readfirstlane((int)lower);
    sweep([&](int buf, uint32_t qm, auto&& wr_lds) {
      const double* As = ops + (size_t)(buf * SIDES) * 4 * SEG;
      uint32_t colm = 0u, on = 0u;
#pragma unroll
      for (int j = 0; j < NH; ++j) colm |= ((qm & bitsB[j]) != 0 ? 1u : 0u) << j;
#pragma unroll
      for (int i = 0; i < NH; ++i) on |= ((qm & bitsA[i]) != 0 ? colm : 0u) << (NH * i);
      on = (VGG_NO_SKIP ? 0xFFFFFFFFu : (uint32_t)__builtin_amdgcn_readfirstlane((int)on)) & lower;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        uint32_t m = on;
        asm volatile("" : "+s"(m));                 // (keeps the tests scalar bit tests; see the off-diagonal variant)
        double a[NH], bq[NH];
#pragma unroll
        for (int i = 0; i < NH; ++i) a[i] = As[rowoffA[i] + ks * R];
        if (SAME || same_rt) {                      // (scalar branch)
#pragma unroll
          for (int i = 0; i < NH; ++i) bq[i] = a[i];
        } else {
#pragma unroll
          for (int i = 0; i < NH; ++i) bq[i] = As[rowoffB[i] + ks * R];
        }
#pragma unroll
        for (int i = 0; i < NH; ++i)
#pragma unroll
          for (int j = 0; j < NH; ++j) {
            if (TR && j > i) continue;                // (classes of the compressed tiles: i >= j only)
            if (m & (1u << (NH * i + j))) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], bq[j], acc[i][j], 0, 0, 0);
          }
        wr_lds(ks);
      }
    });
#pragma unroll
    for (int i = 0; i < NH; ++i)
#pragma unroll
      for (int j = 0; j < NH; ++j) {
        if (TR && j > i) continue;
        if (lower & (1u << (NH * i + j))) store_subtile(wr + 2 * i, wc + 2 * j, acc[i][j]);
      }
#else
    // diagonal tile (A == B, symmetric): only the NT (NT + 1) / 2 sub-tiles of the lower triangle are computed;
    // they are enumerated row by row and dealt round-robin to the 4 waves (wave-uniform scalar tables); skipped like
    // the off-diagonal ones when the quad has no camera in the rows or in the columns of the sub-tile
    constexpr int NL = NT * (NT + 1) / 2;
    constexpr int PER = (NL + 3) / 4;
    static_assert(PER <= NH * NH, "accumulators");
    const int nmine = __builtin_amdgcn_readfirstlane((NL - wave + 3) / 4);
    int rbs[PER], cbs[PER], offA[PER], offB[PER];
    uint32_t bitsR[PER], bitsC[PER];
    {
      int rb = 0, cb = 0;                           // walk to sub-tile `wave`
      for (int t = 0; t < wave; ++t) { if (cb == rb) { ++rb; cb = 0; } else ++cb; }
#pragma unroll
      for (int t = 0; t < PER; ++t) {
        rbs[t] = min(rb, NT - 1); cbs[t] = min(cb, NT - 1);
        offA[t] = kbase + ((16 * rbs[t] + li) ^ swz); offB[t] = kbase + ((16 * cbs[t] + li) ^ swz);
        bitsR[t] = (t < nmine) ? block_slot_bits<BD>(rbs[t]) : 0u;
        bitsC[t] = (t < nmine) ? block_slot_bits<BD>(cbs[t]) : 0u;
        for (int u = 0; u < 4; ++u) { if (cb == rb) { ++rb; cb = 0; } else ++cb; }
      }
    }
    sweep([&](int buf, uint32_t qm, auto&& wr) {
      const double* As = ops + (size_t)(buf * SIDES) * 4 * SEG;
      uint32_t on = 0u;                             // bit t: sub-tile t of this wavefront has a camera in its rows and in its columns
#pragma unroll
      for (int t = 0; t < PER; ++t) on |= (((qm & bitsR[t]) != 0 && (qm & bitsC[t]) != 0) ? 1u : 0u) << t;
      on = VGG_NO_SKIP ? ((1u << nmine) - 1u) : (uint32_t)__builtin_amdgcn_readfirstlane((int)on);
      // (a sub-tile's row and column operands are fetched per K step, right in front of its matrix instruction: the diagonal
      //  launch runs at four workgroups per CU and has no registers for a second operand set)
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        uint32_t m = on;
        asm volatile("" : "+s"(m));                 // (keeps the tests scalar bit tests; see the off-diagonal variant)
        double a[PER], bq[PER];
#pragma unroll
        for (int t = 0; t < PER; ++t) { a[t] = As[offA[t] + ks * R]; bq[t] = As[offB[t] + ks * R]; }
#pragma unroll
        for (int t = 0; t < PER; ++t)
          if (m & (1u << t))
            acc[t / NH][t % NH] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], bq[t], acc[t / NH][t % NH], 0, 0, 0);
        wr(ks);
      }
    });
#pragma unroll
    for (int t = 0; t < PER; ++t)
      if (t < nmine) store_subtile(rbs[t], cbs[t], acc[t / NH][t % NH]);
#endif
    if constexpr (TR && HELPER) {
      if (trhs) {                                     // the chunk's 96 x 3 block, a lane per tile row
        const int rrow = tid & 127;
        if (rrow < R) {
          double* dst = w.rz_part + ((size_t)chunk * R + rrow) * 3;
          dst[0] = racc[0]; dst[1] = racc[1]; dst[2] = racc[2];
        }
      }
    }
    if constexpr (TRF) {                              // full factors: [chunk][pair of entries][tile row], 288 doubles per chunk
      if (trf && tid < 2 * R && !(VGG_TRF_ABLATE & 4)) w.rz_part[(size_t)chunk * (kGroup * 18) + tid] = rfull;
    }
  }
  };   // run
  if constexpr (TR) {
    if (wave < 2) run(std::false_type{}); else run(std::true_type{});
  } else {
    run(std::false_type{});
  }
}

// the two launches of a tile batch (off-diagonal tiles, diagonal tiles) ...
template <int BD, bool DIAG>
__global__ __launch_bounds__(256, (BD == 6 ? (DIAG ? VGG_DIAG_OCC : VGG_OFFDIAG_OCC) : 2)) void schur_tile_kernel(Ws w, const int32_t* __restrict__ chunk_desc,
                                                         const int32_t* __restrict__ entries, int chunk0, int zero_seg) {
  __shared__ __attribute__((aligned(16))) double ops[2 * (DIAG ? 1 : 2) * 4 * kGroup * BD * 3 + (DIAG ? 96 : 0)];
  if (w.ctl->done) return;
  const int chunk = chunk0 + blockIdx.x;
  schur_tile_body<BD, DIAG>(w, chunk_desc, entries, chunk, zero_seg, ops);
}
// ... or ONE launch for both (vgg_ba_tuning / VGG_TILE_MERGED): the diagonal tiles then sweep the points together with the
// off-diagonal ones and find the segments those have just brought into the Infinity Cache
template <int BD>
__global__ __launch_bounds__(256, (BD == 6 ? VGG_OFFDIAG_OCC : 2)) void schur_tile_merged_kernel(Ws w, const int32_t* __restrict__ chunk_desc,
                                                         const int32_t* __restrict__ entries, int chunk0, int zero_seg) {
  __shared__ __attribute__((aligned(16))) double ops[2 * 2 * 4 * kGroup * BD * 3];
  if (w.ctl->done) return;
  const int chunk = chunk0 + blockIdx.x;
  if (chunk_desc[6 * chunk] == chunk_desc[6 * chunk + 1]) schur_tile_body<BD, true>(w, chunk_desc, entries, chunk, zero_seg, ops);
  else schur_tile_body<BD, false>(w, chunk_desc, entries, chunk, zero_seg, ops);
}

// ------------------------------------------------------------------------------------------------------------------
// Round-6 A/B (vgg_ba_set_tile_dma): the off-diagonal launch of 6 x 6 tiles with LDS-DMA staging in
// today's three-workgroups-per-CU shape.  The segments are read from their EXPANDED image Yx[seg][component][96 rows] -- what
// schur_tile_body rebuilds from the compressed records on the way into LDS (12 DPP moves, 18 fp64 operations, 19 selects and six
// ds_writes per lane and batch) -- by global_load_lds_dwordx4: 1 KB per wavefront instruction, straight into the LDS image, no
// staging registers, no write phase.  The XOR swizzle of the odd entries sits on the SOURCE address (the destination of an
// LDS-DMA instruction is lane-linear), the operand reads are those of schur_tile_body.
// Two LDS buffers (three would not fit three workgroups per CU: 3 x 18 KB each), so the image of batch b + 1 is requested
// behind barrier(b - 1) and has ONE matrix phase to land; mode 2 adds a touch of batch b + 2's lines (one dword load per 128
// bytes) so that the DMA finds them in the L2.  For the measurement the expanded image is produced by a kernel of its own from
// the compressed records (expand_segments_kernel, timed apart); in a product form point_pass would write it.
// MEASURED, NOT ADOPTED (profiles/r06_ab_tile_ldsdma_c3.jsonl, configs[2], same box, two rounds, sums bit-identical to the
// register-staged launch at equal chunking): register staging 0.528-0.532 ms; LDS-DMA at three workgroups per CU 0.559-0.565;
// at FOUR per CU (the kernel needs 114 registers and 36 KB: it fits) 0.516-0.525; with the touch 0.66-0.72; with the last K step
// behind the barrier (-DVGG_DMA_PIPE) 2.3.  The write phase is gone, but six DMA instructions per wavefront and batch cost
// their own issue time (the guide: 60-185 cycles a piece next to matrix instructions) and the image has one matrix phase to land
// in front of a vmcnt(0) + barrier; the -1..2 % of the best variant is less than the 0.24 GB per iteration the expanded image
// would add to point_pass' writes (0.33 ms for the stand-alone expansion).  Closed.
__global__ __launch_bounds__(256) void expand_segments_kernel(Ws w, int num_segments) {
  if (w.ctl->done) return;
  const size_t slot = (size_t)blockIdx.x * 256 + threadIdx.x;            // (segment, camera slot)
  if (slot >= ((size_t)num_segments + 1) * kGroup) return;
  const size_t seg = slot / kGroup;
  const int sl = (int)(slot - seg * kGroup);
  const double* rec = w.Y + slot * kYc;                                 // N (3 x 3, row-major), 2 a
  double m[kYc];
#pragma unroll
  for (int i = 0; i < kYc; ++i) m[i] = rec[i];
  double* dst = w.Yx + seg * (kGroup * 18) + 6 * sl;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double n0 = m[k], n1 = m[3 + k], n2 = m[6 + k];
    // rows 0..2 = (2 a) x N[:, k] with the operations of schur_tile_body's rebuild, rows 3..5 = N[:, k]
    dst[k * 96 + 0] = m[10] * n2 - m[11] * n1;
    dst[k * 96 + 1] = m[11] * n0 - m[9] * n2;
    dst[k * 96 + 2] = m[9] * n1 - m[10] * n0;
    dst[k * 96 + 3] = n0; dst[k * 96 + 4] = n1; dst[k * 96 + 5] = n2;
  }
}

#ifndef VGG_DMA_PIPE
#define VGG_DMA_PIPE 0               // 1: the last K step of a batch behind its barrier, as schur_tile_body's pipelined order
#endif
template <int MODE>
__global__ __launch_bounds__(256, VGG_OFFDIAG_OCC) void schur_tile_dma_kernel(Ws w, const int32_t* __restrict__ chunk_desc,
                                                                              const int32_t* __restrict__ entries, int chunk0, int zero_seg) {
  constexpr int BD = 6, R = kGroup * BD, SEG = 3 * R, NT = R / 16, NH = (NT + 1) / 2, SWZ = 16;
  static_assert((SEG * 8) % 256 == 0, "swizzle");
  __shared__ __attribute__((aligned(16))) double ops[2 * 2 * 4 * SEG];
  if (w.ctl->done) return;
  const int chunk = chunk0 + blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int e0 = chunk_desc[6 * chunk + 2], e1 = chunk_desc[6 * chunk + 3];
  const int cj = chunk_desc[6 * chunk + 4], cJ = chunk_desc[6 * chunk + 5];
  constexpr int BPS = kSub / 4;
  const int nsub = (e1 - e0 + kSub - 1) / kSub;
  const int nb = ((nsub - cj + cJ - 1) / cJ) * BPS;
  auto ebase = [&](int b) -> int { return e0 + ((b / BPS) * cJ + cj) * kSub + (b % BPS) * 4; };
  f64x4_t acc[NH][NH];
#pragma unroll
  for (int i = 0; i < NH; ++i)
#pragma unroll
    for (int j = 0; j < NH; ++j) acc[i][j] = (f64x4_t){0.0, 0.0, 0.0, 0.0};
  // wavefront `wave` brings in the two segments of entry `wave` of a batch: 144 sixteen-byte pieces each = two full
  // instructions + one of 16 lanes.  Piece y of the LDS image holds piece y ^ 8 of the segment when the entry is odd.
  const char* Yx = reinterpret_cast<const char*>(w.Yx);
  const int lane_off = ((lane ^ ((wave & 1) * (SWZ / 2))) * 16);
  auto seg_of = [&](int b, int side) -> int {
    const int e = ebase(b) + wave;
    return (e < e1) ? entries[4 * (size_t)e + 1 + side] : zero_seg;
  };
  // (the segment indices of a batch are scalar loads issued one batch ahead: no round trip in front of the DMA)
  int seg_nextA = seg_of(0, 0), seg_nextB = seg_of(0, 1);
  auto issue_dma = [&](int b, int buf) __attribute__((always_inline)) {
    const int segs[2] = {seg_nextA, seg_nextB};
    seg_nextA = seg_of(b + 1, 0); seg_nextB = seg_of(b + 1, 1);
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const char* src = Yx + (size_t)segs[side] * (SEG * 8) + lane_off;
      double* dst = ops + (size_t)((buf * 2 + side) * 4 + wave) * SEG;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 1024),
                                       (__attribute__((address_space(3))) void*)(dst + 128), 16, 0, 0);
      // the 16-lane tail: NOT behind a lane condition -- the compiler duplicates the neighbouring DMA statements into the two arms
      // of such a branch and merges their tails with the LDS destination as a per-lane value (its readfirstlane then serves one
      // arm: batch 0 of every chunk got side B's pieces from the wrong place).  All lanes run the statement; the execution mask
      // is narrowed inside it (M0 = the wave-uniform LDS byte address, written in the statement that reads it).
      {
        const unsigned lds_tail = (unsigned)__builtin_amdgcn_readfirstlane((int)(uintptr_t)(__attribute__((address_space(3))) void*)(dst + 256));
        const char* gtail = src + 2048;
        unsigned long long keep;
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 0xffff\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, off\n\ts_mov_b64 exec, %0"
                     : "=&s"(keep) : "v"(gtail), "s"(lds_tail) : "memory");
      }
    }
  };
  // (mode 2) one dword of every 128-byte line of the two segments of batch b: lanes 0..17 side A, 32..49 side B
  auto touch = [&](int b) -> int {
    int v = 0;
    if constexpr (MODE == 2) {
      const int sub = lane & 31;
      if (sub < 18) {
        const char* src = Yx + (size_t)seg_of(b, lane >> 5) * (SEG * 8) + sub * 128;
        asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(src) : "memory");
      }
    }
    return v;
  };
  const int li = lane & 15, lk = lane >> 4;
  const int kbase = lk * SEG, swz = (lk & 1) * SWZ;
  auto load_quad_mask = [&](int eb) -> uint32_t { return (uint32_t)entries[4 * (size_t)min(eb, e1 - 1) + 3]; };
  const int wr = wave >> 1, wc = wave & 1;
  int rowoffA[NH], rowoffB[NH];
  uint32_t bitsA[NH], bitsB[NH];
#pragma unroll
  for (int i = 0; i < NH; ++i) {
    rowoffA[i] = kbase + ((16 * min(wr + 2 * i, NT - 1) + li) ^ swz);
    rowoffB[i] = kbase + ((16 * min(wc + 2 * i, NT - 1) + li) ^ swz);
    bitsA[i] = block_slot_bits<BD>(wr + 2 * i);
    bitsB[i] = block_slot_bits<BD>(wc + 2 * i) << 16;
  }
  double a[2][NH], bq[2][NH];
  auto fetch = [&](int set, int buf, int ks) __attribute__((always_inline)) {
    const double* As = ops + (size_t)(buf * 2) * 4 * SEG;
    const double* Bs = ops + (size_t)(buf * 2 + 1) * 4 * SEG;
#pragma unroll
    for (int i = 0; i < NH; ++i) { a[set][i] = As[rowoffA[i] + ks * R]; bq[set][i] = Bs[rowoffB[i] + ks * R]; }
  };
  auto pin = [&](int set) __attribute__((always_inline)) {
    static_assert(NH == 3, "operand sets");
    asm volatile("" : "+v"(a[set][0]), "+v"(a[set][1]), "+v"(a[set][2]), "+v"(bq[set][0]), "+v"(bq[set][1]), "+v"(bq[set][2]));
  };
  uint32_t on = 0u;
  auto skip_bits = [&](uint32_t qm) -> uint32_t {
    uint32_t colm = 0u, o = 0u;
#pragma unroll
    for (int j = 0; j < NH; ++j) colm |= ((qm & bitsB[j]) != 0 ? 1u : 0u) << j;
#pragma unroll
    for (int i = 0; i < NH; ++i) o |= ((qm & bitsA[i]) != 0 ? colm : 0u) << (NH * i);
    return VGG_NO_SKIP ? 0xFFFFFFFFu : (uint32_t)__builtin_amdgcn_readfirstlane((int)o);
  };
  auto group = [&](int set) __attribute__((always_inline)) {
    uint32_t m = on;
    asm volatile("" : "+s"(m));
#pragma unroll
    for (int i = 0; i < NH; ++i)
#pragma unroll
      for (int j = 0; j < NH; ++j)
        if (m & (1u << (NH * i + j))) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[set][i], bq[set][j], acc[i][j], 0, 0, 0);
  };
  uint32_t qmask = load_quad_mask(ebase(0)), qmask_next = load_quad_mask(ebase(1));
  issue_dma(0, 0);
  int tv = touch(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                  // (the compiler puts the vmcnt(0) of the LDS-DMA in front of it)
  asm volatile("" :: "v"(tv));
#if VGG_DMA_PIPE
  fetch(0, 0, 0);
#endif
  for (int b = 0; b < nb; ++b) {
    const int buf = b & 1;
    issue_dma(b + 1, buf ^ 1);                      // buffer buf^1: its last readers (batch b - 1) finished in front of barrier(b - 1)
    tv = touch(b + 2);
    on = skip_bits(qmask);
#if VGG_DMA_PIPE
    // (operand set P = b & 1 holds K step 0 of batch b; the sets alternate roles from batch to batch -- but the set index must be
    //  a compile-time constant, so the loop body is written for both parities)
    if (buf == 0) {
      fetch(1, buf, 1); pin(0); group(0);
      fetch(0, buf, 2); pin(1); group(1);
      pin(0);
      qmask = qmask_next; qmask_next = load_quad_mask(ebase(b + 2));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the tail pieces are asm statements: not in the compiler's count)
      __syncthreads();
      asm volatile("" :: "v"(tv));
      fetch(1, buf ^ 1, 0);
      pin(0); group(0);
    } else {
      fetch(0, buf, 1); pin(1); group(1);
      fetch(1, buf, 2); pin(0); group(0);
      pin(1);
      qmask = qmask_next; qmask_next = load_quad_mask(ebase(b + 2));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the tail pieces are asm statements: not in the compiler's count)
      __syncthreads();
      asm volatile("" :: "v"(tv));
      fetch(0, buf ^ 1, 0);
      pin(1); group(1);
    }
#else
    fetch(0, buf, 0);
    fetch(1, buf, 1); pin(0); group(0);
    fetch(0, buf, 2); pin(1); group(1);
    pin(0); group(0);
    qmask = qmask_next; qmask_next = load_quad_mask(ebase(b + 2));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    asm volatile("" :: "v"(tv));
#endif
  }
  double* part = w.tile_part + (size_t)chunk * R * R;
#pragma unroll
  for (int i = 0; i < NH; ++i)
#pragma unroll
    for (int j = 0; j < NH; ++j)
      if (wr + 2 * i < NT && wc + 2 * j < NT) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) part[(size_t)(16 * (wr + 2 * i) + lk + 4 * reg) * R + 16 * (wc + 2 * j) + li] = acc[i][j][reg];
      }
}

// (Second formulation, commit a638f73, taken out again: WAVE SPECIALISATION -- five wavefronts per workgroup, four consumers that do
//  nothing but [operand reads, matrix instructions, barrier] and one producer that issues the 18 one-KB DMA pieces of the next
//  batch: sums bit-identical, 0.956 ms against 0.529 -- one producer cannot feed four consumers through two LDS buffers, and a
//  third buffer does not fit three workgroups per CU.  profiles/r06_ab_tile_ldsdma_c3.jsonl.)
// S[(cI,a,i),(cJ,b,j)] = - sum over the chunks of tile (gI,gJ) of the partial tiles (plain stores: every
// element of S outside the per-camera diagonal terms belongs to exactly one (tile, element)).
// grid = (R*R/256, num_tiles): one element per thread, chunks summed in order.
template <int BD>
__global__ __launch_bounds__(256) void tile_reduce_kernel(Ws w, int n_red, int C, int KD,
                                                          const int32_t* __restrict__ tile_desc, int tile0,
                                                          double* __restrict__ dst, int want) {
  constexpr int R = kGroup * BD;
  if (w.ctl->done) return;
  const int tile = tile0 + blockIdx.y;
  const int gI = tile_desc[4 * tile], gJ = tile_desc[4 * tile + 1];
  // (want: -1 = every tile of the range; 0 / 1 = only its off-diagonal / diagonal tiles -- the split exchange of the sharded
  //  solve sums the off-diagonal launch's tiles while the diagonal launch has not run yet, phases 7 / 9)
  if (want >= 0 && (gI == gJ) != (want == 1)) return;
  const int c0 = tile_desc[4 * tile + 2], c1 = tile_desc[4 * tile + 3];
  const int n = n_red;
  if constexpr (BD == 6) {
    // tile_rhs: the blocks behind the R x R elements add up, for a diagonal tile, the chunks' 96 x 3 right-hand-side blocks
    // (schur_tile_body's helpers) in the same fixed order -- assemble_kernel then reads 18 numbers per camera instead of
    // walking ~80 chunks with 18 of its 64 threads (24 us of every iteration at configs[2])
    const int nb2 = (R * R + 255) / 256;
    if ((int)blockIdx.x >= nb2) {
      if (gI != gJ || !w.tile_rhs) return;
      const int e = ((int)blockIdx.x - nb2) * 256 + threadIdx.x;
      if (e >= R * 3) return;
      double s0 = 0.0, s1 = 0.0;
      int ch = c0;
      const double* src = w.rz_part + e;
      for (; ch + 7 < c1; ch += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(ch + u) * R * 3];
#pragma unroll
        for (int u = 0; u < 8; u += 2) { s0 += v[u]; s1 += v[u + 1]; }
      }
      for (; ch + 1 < c1; ch += 2) { s0 += src[(size_t)ch * R * 3]; s1 += src[(size_t)(ch + 1) * R * 3]; }
      if (ch < c1) s0 += src[(size_t)ch * R * 3];
      w.rz[(size_t)gI * R * 3 + e] = s0 + s1;
      return;
    }
  }
  if constexpr (BD != 6) {
    // tile_rhs with full factors: the chunks' [pair of entries][tile row] right-hand-side sums of a diagonal tile, in chunk order
    const int nb2 = (R * R + 255) / 256;
    if ((int)blockIdx.x >= nb2) {
      if (gI != gJ || !w.tile_rhs) return;
      const int e = ((int)blockIdx.x - nb2) * 256 + threadIdx.x;
      if (e >= 2 * R) return;
      double s0 = 0.0, s1 = 0.0;
      int ch = c0;
      const double* src = w.rz_part + e;
      constexpr size_t STR = kGroup * 18;
      for (; ch + 7 < c1; ch += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(ch + u) * STR];
#pragma unroll
        for (int u = 0; u < 8; u += 2) { s0 += v[u]; s1 += v[u + 1]; }
      }
      for (; ch + 1 < c1; ch += 2) { s0 += src[(size_t)ch * STR]; s1 += src[(size_t)(ch + 1) * STR]; }
      if (ch < c1) s0 += src[(size_t)ch * STR];
      w.rz[(size_t)gI * STR + e] = s0 + s1;
      return;
    }
  }
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= R * R) return;
  const int row = e / R, col = e - row * R;
  const int a = row / BD, i = row - a * BD, b = col / BD, j = col - b * BD;
  const int ca = gI * kGroup + a, cb = gJ * kGroup + b;
  if (ca >= C || cb >= C) return;
  if (gI == gJ && col > row) return;             // diagonal tile is symmetric: the lower half is the unique source
  const int ri = (i < 6) ? 6 * ca + i : 6 * C + KD * ca + (i - 6);
  const int cj = (j < 6) ? 6 * cb + j : 6 * C + KD * cb + (j - 6);
  // (eight loads in flight per thread; the order of the additions -- even chunks into s0, odd ones into s1 -- is fixed)
  double s0 = 0.0, s1 = 0.0;
  int ch = c0;
  const double* src = w.tile_part + e;
  for (; ch + 7 < c1; ch += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(ch + u) * R * R];
#pragma unroll
    for (int u = 0; u < 8; u += 2) { s0 += v[u]; s1 += v[u + 1]; }
  }
  for (; ch + 1 < c1; ch += 2) { s0 += src[(size_t)ch * R * R]; s1 += src[(size_t)(ch + 1) * R * R]; }
  if (ch < c1) s0 += src[(size_t)ch * R * R];
  const int hi = ri > cj ? ri : cj, lo = ri > cj ? cj : ri;
  double v = -(s0 + s1);
  if constexpr (BD == 6) {
    // compressed factors carry neither the Jacobi scales nor the constant-parameter masks (y_slot_doubles): both act on
    // whole rows / columns of the sum
    const double mi = w.active[ri] ? w.scale_c[ri] : 0.0, mj = w.active[cj] ? w.scale_c[cj] : 0.0;
    v *= mi * mj;
  }
  dst[(size_t)hi * n + lo] = v;
}

// diagonal blocks, camera/intrinsics coupling, damping, right-hand side.  One workgroup (64) per camera,
// plus one extra for the shared-intrinsics block.  Only rank 0 adds the terms that were already summed
// over ranks (U, g, damping); the per-rank systems are then all-reduced.
// tile_rhs: the pose rows of T_c = F^T [r - E hs | -E Ms] are  [g_c | 0] - (sum over the chunks of the camera's diagonal
// tile, in order, of rz_part)  -- g_c from the linearisation pass, rz_part from schur_tile_body -- and the intrinsics rows
// of the shared block come from point_pass' per-workgroup sums part_Q; T and cam_pass<RHS> are not used.
template <int KD>
__global__ __launch_bounds__(64) void assemble_kernel(DevProblem pb, Ws w, const int32_t* __restrict__ tile_desc, int num_tiles,
                                                      int point_parts) {
  constexpr int BD = 6 + KD;
  __shared__ double rz[6][3];
  __shared__ double rzf[BD];                       // tile_rhs with full factors (per-camera intrinsics): BD rows, scaled and masked
  if (w.ctl->done) return;
  const int global_terms = (w.ctl->rank == 0);
  const Dims& d = pb.d;
  const int n = d.n_red, kdsh = d.kdsh, tw = 1 + kdsh;
  const int c = blockIdx.x, tid = threadIdx.x;
  const bool trhs = w.tile_rhs != 0;
  if (c < d.C) {
    const double* U = w.U + (size_t)c * BD * BD;
    const double* T = w.T + (size_t)c * BD * tw;
    const int ia = 6 * d.C + (d.shared ? 0 : KD * c);
    const bool full = trhs && !d.shared && KD > 0;     // full factors: rz carries scales and masks, one column, BD rows
    if (full) {
      if (tid < BD) {
        const int g = c / kGroup;
        const bool has_tile = pb.col_ptr[min((g + 1) * kGroup, d.C)] > pb.col_ptr[g * kGroup];
        const size_t base = (size_t)g * kGroup * 18 + (size_t)(c % kGroup) * BD + tid;
        rzf[tid] = has_tile ? w.rz[base] + w.rz[base + kGroup * BD] : 0.0;      // (the two entry pairs of a batch)
      }
      __syncthreads();
    } else if (trhs) {
      // (a group without observations has no diagonal tile: nothing was summed for it)
      if (tid < 18) {
        const int i = tid / 3, col = tid - 3 * i;
        const int g = c / kGroup;
        const bool has_tile = pb.col_ptr[min((g + 1) * kGroup, d.C)] > pb.col_ptr[g * kGroup];
        rz[i][col] = has_tile ? w.rz[(size_t)g * kGroup * 18 + ((size_t)(c % kGroup) * 6 + i) * 3 + col] : 0.0;
      }
      __syncthreads();
    }
    // (the compressed factors behind rz carry no constant-parameter masks -- cam_pass<RHS> evaluated MASKED Jacobians, so a
    //  constant pose / translation component contributed exact zeros: the mask acts on the row here, like in tile_reduce)
    auto Tc = [&](int i, int m) -> double {
      if (!trhs) return T[i * tw + m];
      if (!w.active[6 * c + i]) return 0.0;
      return (m == 0 ? (global_terms ? w.g[(size_t)c * BD + i] : 0.0) : 0.0) - rz[i][m];
    };
    // BD x BD block (lower part), rows/cols mapped to reduced indices
    for (int e = tid; e < BD * BD; e += 64) {
      const int i = e / BD, j = e - i * BD;
      const int ri = (i < 6) ? 6 * c + i : ia + (i - 6), cj = (j < 6) ? 6 * c + j : ia + (j - 6);
      if (cj > ri) continue;
      if (d.shared && i >= 6 && j >= 6) continue;           // intr-intr of the shared block: extra workgroup
      if (trhs && !full && !d.shared && (i >= 6 || j >= 6)) continue;   // (compressed tile_rhs without shared intrinsics: none are refined)
      double v = 0.0;
      if (global_terms) {
        v = w.scale_c[ri] * w.scale_c[cj] * U[i * BD + j];
        if (ri == cj) v += w.dsq_c[ri];
      }
      if (d.shared && i >= 6) v += w.scale_c[cj] * Tc(j, 1 + (i - 6));   // -F_pose^T E M
      w.S[(size_t)ri * n + cj] += v;
    }
    if (full) {
      if (tid < BD) {
        const int ri = (tid < 6) ? 6 * c + tid : ia + (tid - 6);
        const double v = (global_terms ? w.scale_c[ri] * w.g[(size_t)c * BD + tid] : 0.0) - rzf[tid];
        w.rhs[ri] += w.active[ri] ? v : 0.0;
      }
    } else {
      if (tid < 6) w.rhs[6 * c + tid] += w.scale_c[6 * c + tid] * Tc(tid, 0);
      if (!trhs && !d.shared && tid >= 6 && tid < BD) w.rhs[ia + tid - 6] += w.scale_c[ia + tid - 6] * T[tid * tw];
    }
  } else {
    if (trhs) {                                    // (rides along: max of the point passes' per-workgroup gradient norms,
      double m = 0;                                //  what cam_reduce_kernel<KD, 1> did)
      // (these loops run in ONE wavefront and each trip is a round trip to memory: unrolled, so that several are in flight --
      //  the extra workgroup was the long pole of the launch, ~20 of its 23 us at configs[2])
#pragma unroll 8
      for (int i = tid; i < point_parts; i += 64) m = fmax(m, w.part_B[i]);
      m = wave_max(m);
      if (tid == 0) w.gmax_pts[0] = m;
    }
    if (d.shared && KD > 0) {
      // shared-intrinsics block: sum the per-camera parts (one wavefront, lanes stride over cameras)
      const int ia = 6 * d.C;
      constexpr int NS = (KD > 0) ? KD * KD + KD : 1;
      double sums[NS];
#pragma unroll
      for (int i = 0; i < NS; ++i) sums[i] = 0.0;
#pragma unroll 4
      for (int cc = tid; cc < d.C; cc += 64) {
#pragma unroll
        for (int i = 0; i < KD; ++i) {
#pragma unroll
          for (int j = 0; j <= i; ++j) {
            double v = trhs ? 0.0 : w.scale_c[ia + i] * w.T[((size_t)cc * BD + 6 + i) * tw + 1 + j];
            if (global_terms) v += w.scale_c[ia + i] * w.scale_c[ia + j] * w.U[(size_t)cc * BD * BD + (6 + i) * BD + 6 + j];
            sums[i * KD + j] += v;
          }
          sums[KD * KD + i] += trhs ? (global_terms ? w.g[(size_t)cc * BD + 6 + i] : 0.0) : w.T[((size_t)cc * BD + 6 + i) * tw];
        }
      }
      if (trhs) {
        // - sum_p Wa_p Ms_p^T (scaled like the T terms above) and - sum_p Wa_p hs_p, workgroup partials in launch order
#pragma unroll 8
        for (int b = tid; b < point_parts; b += 64) {
          const double* q = w.part_Q + 8 * (size_t)b;
#pragma unroll
          for (int i = 0; i < KD; ++i) {
#pragma unroll
            for (int j = 0; j <= i; ++j) sums[i * KD + j] -= w.scale_c[ia + i] * q[KD + i * KD + j];
            sums[KD * KD + i] -= q[i];
          }
        }
      }
#pragma unroll
      for (int i = 0; i < NS; ++i) sums[i] = wave_sum(sums[i]);
      if (tid == 0) {
#pragma unroll
        for (int i = 0; i < KD; ++i) {
#pragma unroll
          for (int j = 0; j <= i; ++j) {
            double v = sums[i * KD + j];
            if (global_terms && i == j) v += w.dsq_c[ia + i];
            w.S[(size_t)(ia + i) * n + ia + j] += v;
          }
          w.rhs[ia + i] += w.scale_c[ia + i] * sums[KD * KD + i];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// candidate cameras: x (+) (-s o y); also the camera-side part of |step| and |x|
template <int KD>
__global__ void cam_update_kernel(DevProblem pb, Ws w) {
  if (w.ctl->done) return;
  const Dims& d = pb.d;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > d.C) return;
  double step = 0, xn = 0;
  if (c < d.C) {
    double dl[6];
    for (int k = 0; k < 6; ++k) {
      const int j = 6 * c + k;
      const double dyv = w.active[j] ? w.scale_c[j] * w.rhs[j] : 0.0;
      w.dy[j] = dyv;
      dl[k] = -dyv;
    }
    double qn[4];
    quat_plus(pb.cam_q + 4 * c, dl, qn);
    // |x| counts only blocks Ceres keeps in the reduced program (a constant pose is removed from it)
    const bool pose_var = !(pb.cam_const && (pb.cam_const[c] & 1u));
    for (int k = 0; k < 4; ++k) { w.cand_q[4 * c + k] = qn[k]; const double df = qn[k] - pb.cam_q[4 * c + k]; step += df * df; if (pose_var) xn += pb.cam_q[4 * c + k] * pb.cam_q[4 * c + k]; }
    for (int k = 0; k < 3; ++k) { const double tn = pb.cam_t[3 * c + k] + dl[3 + k]; w.cand_t[3 * c + k] = tn; step += dl[3 + k] * dl[3 + k]; if (pose_var) xn += pb.cam_t[3 * c + k] * pb.cam_t[3 * c + k]; }
  }
  // intrinsics block a == c (per camera) or block 0 handled by the extra thread c == C (shared)
  const int a = d.shared ? ((c == d.C) ? 0 : -1) : ((c < d.C) ? c : -1);
  if (a >= 0) {
    double in4[4];
    for (int k = 0; k < 4; ++k) in4[k] = pb.intr[4 * a + k];
    for (int k = 0; k < KD; ++k) {
      const int j = 6 * d.C + KD * a + k;
      const double dyv = w.active[j] ? w.scale_c[j] * w.rhs[j] : 0.0;
      w.dy[j] = dyv;
      const int slot = (KD == 2) ? (k == 0 ? 0 : 3) : (d.only_k ? 3 : 0);
      in4[slot] -= dyv;
      step += dyv * dyv;
    }
    const bool intr_var = KD > 0 && !(pb.intr_const && pb.intr_const[a]);
    const int np = (d.model == kSimpleRadial) ? 4 : 3;
    for (int k = 0; k < 4; ++k) { w.cand_intr[4 * a + k] = in4[k]; if (intr_var && k < np) xn += pb.intr[4 * a + k] * pb.intr[4 * a + k]; }
  }
  // Model cost change, the part that is a function of the camera-side step alone (compressed Schur factors: point_step_kernel
  // no longer re-evaluates the Jacobians): sum over the observations of [F dy . r - |F dy|^2 / 2] = sum over the cameras of
  // dy_c . g_c - dy_c^T U_c dy_c / 2, with U_c = sum F^T F and g_c = sum F^T r of the linearisation in place (cam_pass
  // <linearize>, all-reduced over the ranks) and dy_c the camera's pose step followed by the step of its intrinsics block.
  double mc = 0.0;
  if (c < d.C && w.step_from_factors) {
    constexpr int BD = 6 + KD;
    double dv[BD];
    for (int k = 0; k < 6; ++k) dv[k] = w.active[6 * c + k] ? w.scale_c[6 * c + k] * w.rhs[6 * c + k] : 0.0;
    const int ai = d.shared ? 0 : c;
    for (int k = 0; k < KD; ++k) { const int j = 6 * d.C + KD * ai + k; dv[6 + k] = w.active[j] ? w.scale_c[j] * w.rhs[j] : 0.0; }
    const double* U = w.U + (size_t)c * BD * BD;
    const double* g = w.g + (size_t)c * BD;
    double lin = 0.0, quad = 0.0;
    for (int i = 0; i < BD; ++i) {
      double ui = 0.0;
      for (int k = 0; k < BD; ++k) ui += U[i * BD + k] * dv[k];
      quad += dv[i] * ui;
      lin += dv[i] * g[i];
    }
    mc = lin - 0.5 * quad;
  }
  w.cam_part[3 * c] = step;
  w.cam_part[3 * c + 1] = xn;
  w.cam_part[3 * c + 2] = mc;
}

// back-substitution, model cost change, candidate point and candidate cost: LPP lanes per point (see point_pass_kernel)
// FYM = 1 (round 4, OPT-IN: vgg_ba_set_step_from_factors / VGG_STEP_FACTORS; measured SLOWER, see below): the tile blocks
// are 6 x 6 and the segment buffer holds the compressed Schur factors N = Jw^T (E G), 2 a of every observation
// (point_pass_kernel, CY).  Then E^T F dy -- all this kernel needs of the Jacobians -- is there already:
// F_pose dy = Jw (dy_t - 2 a x dy_w), so G^T sum E^T F dy = sum N^T (dy_t - 2 a x dy_w): one 96-byte record and ~25
// multiply-adds per observation instead of a second evaluation of the projection and its Jacobians (~150 FP64 operations);
// the shared intrinsics' share is Ms (dy_a / scale) per point, and the part of the model cost change that only depends on
// the camera step comes from the cameras' U, g (cam_update_kernel).  The candidate's residuals are still evaluated here.
// Same step to rounding (tests/test_gpu_ba.py::test_ba_step_from_factors_matches_evaluation).  c3, same box
// (profiles/r04_ab_step_from_factors_c3.jsonl): 0.143 ms re-evaluating (210 VGPRs, 2 wavefronts / SIMD, FP64 issue bound),
// 0.170 ms from the factors (160 VGPRs, 3 wavefronts / SIMD): a point's records lie in as many segments as it has
// observations, each 96-byte record touches 1.5 cache lines on average -- 5 M x 192 B = 0.96 GB of line traffic in 0.17 ms
// is the memory system's limit for this access pattern.  FYM = 2, every other sweep of a lane from the factors so that
// both pipes have work: 0.147 ms -- no better than re-evaluating.  So the default stays 0.
template <int KD, bool LDSCAM, int LPP, bool LONGT = false, int FYM = 0>
__global__ __launch_bounds__(256, FYM == 1 ? VGG_PS_OCC_FY : VGG_PS_OCC) void point_step_kernel(DevProblem pb, Ws w) {
  constexpr bool FY = FYM == 1;                  // every observation from its factor
  constexpr bool HY = FYM == 2;                  // (measurement only) odd sweeps from the factors, even sweeps re-evaluated
  constexpr int BD = 6 + KD;
  __shared__ double red[4][4];
  extern __shared__ double cam_cache[];   // LDSCAM: R[9C] t[3C] dy_pose[6C] cand R[9C] cand_t[3C] flags[C]   (FY: dy, cand R, cand t only)
  if (w.ctl->done) return;
  const Dims& d = pb.d;
  constexpr int PPW = 64 / LPP;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane / LPP, sl = lane % LPP;
  const int nw = gridDim.x * 4 * PPW;
  double s_cost = 0, s_mcc = 0, s_step = 0, s_xn = 0;
  const double* lq = cam_cache;                  // rotation matrices [9C]
  const double* lt = lq + (FY ? 0 : 9 * d.C);
  const double* ldy = lt + (FY ? 0 : 3 * d.C);
  const double* lcq = ldy + 6 * d.C;             // rotation matrices of the candidate [9C]
  const double* lct = lcq + 9 * d.C;
  const double* lfl = lct + 3 * d.C;
  if (LDSCAM) {
    double* cc = cam_cache;
    for (int i = threadIdx.x; i < d.C; i += 256) {
      if (!FY) {
        quat_to_R(pb.cam_q + 4 * i, cc + 9 * i);
        cc[30 * d.C + i] = pb.cam_const ? (double)pb.cam_const[i] : 0.0;
      }
      quat_to_R(w.cand_q + 4 * i, cc + (lcq - cam_cache) + 9 * i);
    }
    for (int i = threadIdx.x; i < 3 * d.C; i += 256) { if (!FY) cc[9 * d.C + i] = pb.cam_t[i]; cc[(lct - cam_cache) + i] = w.cand_t[i]; }
    for (int i = threadIdx.x; i < 6 * d.C; i += 256) cc[(ldy - cam_cache) + i] = w.dy[i];
    __syncthreads();
  }
  (void)lfl;
  // same software pipeline over the points of a wavefront as in point_pass_kernel
  int p = (blockIdx.x * 4 + wave) * PPW + sub;
  constexpr int NPF = LONGT ? 4 : 2;             // prefetched observations per lane (see point_pass_kernel)
  int n_o0 = 0, n_o1 = 0;
  ObsPf<NPF> n_pf;
  double n_X0 = 0, n_X1 = 0, n_X2 = 0;
  int n_ptc = 0;                                // (constant-point flag as loaded: tested where it is used -- a test right
                                                //  behind the load is a wait for it, and for every load issued before it)
  // (two stages: the row bounds / coordinates are loaded TWO points ahead, the observations ONE point ahead, so
  //  that the observation loads never wait for the row-bound load they depend on)
  int m_o0 = 0, m_o1 = 0;
  double m_X0 = 0, m_X1 = 0, m_X2 = 0;
  int m_ptc = 0;
  if (p < d.P) {
    n_o0 = pb.row_ptr[p]; n_o1 = pb.row_ptr[p + 1];
    n_X0 = pb.pts[3 * p]; n_X1 = pb.pts[3 * p + 1]; n_X2 = pb.pts[3 * p + 2];
    n_ptc = pb.pt_const ? (int)pb.pt_const[p] : 0;
    n_pf.template load<(FYM != 0)>(pb.obs_cam, pb.obs_uv, pb.obs_slot, n_o0, n_o1, LPP, sl);
    if (p + nw < d.P) {
      const int pm = p + nw;
      m_o0 = pb.row_ptr[pm]; m_o1 = pb.row_ptr[pm + 1];
      m_X0 = pb.pts[3 * pm]; m_X1 = pb.pts[3 * pm + 1]; m_X2 = pb.pts[3 * pm + 2];
      m_ptc = pb.pt_const ? (int)pb.pt_const[pm] : 0;
    }
  }
  for (; p < d.P; p += nw) {
    const int o0 = n_o0, o1 = n_o1;
    const double X[3] = {n_X0, n_X1, n_X2};
    const bool pt_c = n_ptc != 0;
    const ObsPf<NPF> f_pf = n_pf;
    {
      // stage 1 -> current of the next iteration: observations of point p + nw (its bounds arrived an iteration ago)
      n_o0 = m_o0; n_o1 = m_o1; n_X0 = m_X0; n_X1 = m_X1; n_X2 = m_X2; n_ptc = m_ptc;
      if (p + nw < d.P) n_pf.template load<(FYM != 0)>(pb.obs_cam, pb.obs_uv, pb.obs_slot, n_o0, n_o1, LPP, sl);
      // stage 2: bounds / coordinates of point p + 2 nw
      const int pm = p + 2 * nw;
      if (pm < d.P) {
        m_o0 = pb.row_ptr[pm]; m_o1 = pb.row_ptr[pm + 1];
        m_X0 = pb.pts[3 * pm]; m_X1 = pb.pts[3 * pm + 1]; m_X2 = pb.pts[3 * pm + 2];
        m_ptc = pb.pt_const ? (int)pb.pt_const[pm] : 0;
      }
    }
    double t3[3] = {0, 0, 0};
    // (the point's G and hs are requested here, in front of the Jacobian sweep, not behind it where they are used: behind
    //  the sweep every point paid their round trip -- ~1 us of ~5 -- with nothing left to overlap it)
    const double* Gp = w.G + 6 * (size_t)p;
    const double G00 = Gp[0], G01 = Gp[1], G02 = Gp[2], G11 = Gp[3], G12 = Gp[4], G22 = Gp[5];
    const double hs0 = w.hs[3 * (size_t)p], hs1 = w.hs[3 * (size_t)p + 1], hs2 = w.hs[3 * (size_t)p + 2];
    const double pd0 = w.pdamp[3 * (size_t)p], pd1 = w.pdamp[3 * (size_t)p + 1], pd2 = w.pdamp[3 * (size_t)p + 2];
    // Model cost change without a second evaluation of the Jacobians.  With m = -(F dy + E ys) the model residual of an
    // observation (Ceres: -sum m.(r + m/2)), the sum over the observations of a point is
    //   sum [F dy . r - |F dy|^2 / 2]  +  ys^T g - ys^T t3 - ys^T V ys / 2        (g = sum E^T r, t3 = sum E^T F dy, V = sum E^T E)
    // and with (V + D^2) ys = g - t3 and (V + D^2)^-1 = G G^T (point_pass):  ys^T (g - t3) = |G^-1 ys|^2 =: |z|^2, so
    //   = sum [F dy . r - |F dy|^2 / 2]  +  |z|^2 / 2  +  ys^T D^2 ys / 2.
    // The bracket is accumulated in the first sweep; the second sweep only evaluates the candidate's residuals.
    double uf[3] = {0, 0, 0};                      // G^T sum E^T F_pose dy over the observations taken from their factors
    auto from_factor = [&](int pass, int o) __attribute__((always_inline)) {
      const int c = f_pf.cam(pass, pb.obs_cam, o);
      const int slot = f_pf.slot(pass, pb.obs_slot, o);
      const double2* y = reinterpret_cast<const double2*>(w.Y + (size_t)slot * kYc);
      const double2 n01 = y[0], n23 = y[1], n45 = y[2], n67 = y[3], n8a = y[4], a12 = y[5];
      const double* dyc = LDSCAM ? ldy + 6 * c : w.dy + 6 * c;
      const double a0 = n8a.y, a1 = a12.x, a2 = a12.y;                 // 2 a
      const double v0 = dyc[3] - (a1 * dyc[2] - a2 * dyc[1]);
      const double v1 = dyc[4] - (a2 * dyc[0] - a0 * dyc[2]);
      const double v2 = dyc[5] - (a0 * dyc[1] - a1 * dyc[0]);
      uf[0] += n01.x * v0 + n23.y * v1 + n67.x * v2;                   // N row-major: N[i][m] = rec[3 i + m]
      uf[1] += n01.y * v0 + n45.x * v1 + n67.y * v2;
      uf[2] += n23.x * v0 + n45.y * v1 + n8a.x * v2;
    };
    auto evaluated = [&](int pass, int o) __attribute__((always_inline)) {
      const int c = f_pf.cam(pass, pb.obs_cam, o);
      const float2 uv = f_pf.uv(pass, pb.obs_uv, o);
      const int a = d.shared ? 0 : c;
      double r[2], F[2 * BD], E[6];
      if (LDSCAM)
        eval_full<KD>(d, lq + 9 * c, lt + 3 * c, pb.intr + 4 * a, X, uv, (unsigned)lfl[c],
                      pb.intr_const ? pb.intr_const[a] != 0 : false, pt_c, r, F, E);
      else
        eval_full<KD>(d, CamR(pb.cam_q + 4 * c).R, pb.cam_t + 3 * c, pb.intr + 4 * a, X, uv,
                      pb.cam_const ? pb.cam_const[c] : 0u, pb.intr_const ? pb.intr_const[a] != 0 : false, pt_c, r, F, E);
      double fy0 = 0, fy1 = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) { const double v = LDSCAM ? ldy[6 * c + k] : w.dy[6 * c + k]; fy0 += F[k] * v; fy1 += F[BD + k] * v; }
      if (!HY) {                                   // (HY: the intrinsics' share comes from Ms for every observation)
#pragma unroll
        for (int k = 0; k < KD; ++k) { const double v = w.dy[6 * d.C + KD * a + k]; fy0 += F[6 + k] * v; fy1 += F[BD + 6 + k] * v; }
      }
      t3[0] += E[0] * fy0 + E[3] * fy1; t3[1] += E[1] * fy0 + E[4] * fy1; t3[2] += E[2] * fy0 + E[5] * fy1;
      if (!HY) s_mcc += fy0 * (r[0] - 0.5 * fy0) + fy1 * (r[1] - 0.5 * fy1);      // (HY / FY: from the cameras' U, g)
    };
    for (int o = o0 + sl; o < o1; o += LPP) {
      const int pass = (o - o0) / LPP;
      if (FY || (HY && (pass & 1))) from_factor(pass, o);
      else evaluated(pass, o);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) t3[i] = group_sum<LPP>(t3[i]);
    // ys = hs - G (G^T t3 + uf)   (the shared intrinsics add Ms (dy_a / scale_a) when factors are used)
    double u0 = 0, u1 = 0, u2 = 0;
    if (!FY) { u0 = G00 * t3[0]; u1 = G01 * t3[0] + G11 * t3[1]; u2 = G02 * t3[0] + G12 * t3[1] + G22 * t3[2]; }
    if (FYM != 0) { u0 += group_sum<LPP>(uf[0]); u1 += group_sum<LPP>(uf[1]); u2 += group_sum<LPP>(uf[2]); }
    double ys[3];
    ys[0] = hs0 - (G00 * u0 + G01 * u1 + G02 * u2);
    ys[1] = hs1 - (G11 * u1 + G12 * u2);
    ys[2] = hs2 - (G22 * u2);
    if constexpr (FYM != 0 && KD > 0) {
      const double* Msp = w.Ms + (size_t)p * 3 * KD;
#pragma unroll
      for (int m = 0; m < KD; ++m) {
        const int j = 6 * d.C + m;
        const double ds = w.active[j] ? w.rhs[j] : 0.0;
        ys[0] -= Msp[3 * m] * ds; ys[1] -= Msp[3 * m + 1] * ds; ys[2] -= Msp[3 * m + 2] * ds;
      }
    }
    const double Xn[3] = {X[0] - ys[0], X[1] - ys[1], X[2] - ys[2]};
    if (sl == 0) {
      w.cand_pts[3 * (size_t)p] = Xn[0]; w.cand_pts[3 * (size_t)p + 1] = Xn[1]; w.cand_pts[3 * (size_t)p + 2] = Xn[2];
      s_step += ys[0] * ys[0] + ys[1] * ys[1] + ys[2] * ys[2];
      if (!pt_c) s_xn += X[0] * X[0] + X[1] * X[1] + X[2] * X[2];
      if (G22 != 0.0) {                                  // (constant / unobserved points: G = 0, ys = 0)
        const double z2 = ys[2] / G22, z1 = (ys[1] - G12 * z2) / G11, z0 = (ys[0] - G01 * z1 - G02 * z2) / G00;
        s_mcc += 0.5 * (z0 * z0 + z1 * z1 + z2 * z2) + 0.5 * (pd0 * ys[0] * ys[0] + pd1 * ys[1] * ys[1] + pd2 * ys[2] * ys[2]);
      }
    }
    // candidate residuals.  (Measured and rejected, round 4: two observations of a lane evaluated side by side so that
    //  their dependent fp64 chains interleave -- 0.143 -> 0.150 ms at configs[2]; the sweep is not bound by its chain.)
    for (int o = o0 + sl; o < o1; o += LPP) {
      const int pass = (o - o0) / LPP;
      const int c = f_pf.cam(pass, pb.obs_cam, o);
      const int a = d.shared ? 0 : c;
      const float2 uv = f_pf.uv(pass, pb.obs_uv, o);
      double rc[2];
      if (LDSCAM) obs_residual_R(d.model, lcq + 9 * c, lct + 3 * c, w.cand_intr + 4 * a, Xn, (double)uv.x, (double)uv.y, rc);
      else obs_residual_R(d.model, CamR(w.cand_q + 4 * c).R, w.cand_t + 3 * c, w.cand_intr + 4 * a, Xn, (double)uv.x, (double)uv.y, rc);
      s_cost += loss_rho0(d, rc[0] * rc[0] + rc[1] * rc[1]);
    }
  }
  s_cost = wave_sum(s_cost); s_mcc = wave_sum(s_mcc); s_step = wave_sum(s_step); s_xn = wave_sum(s_xn);
  if (lane == 0) { red[wave][0] = s_cost; red[wave][1] = s_mcc; red[wave][2] = s_step; red[wave][3] = s_xn; }
  __syncthreads();
  if (threadIdx.x < 4) w.part_F[4 * blockIdx.x + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void reduce_step_kernel(Ws w, int nparts) {
  __shared__ double red[4][256];
  if (w.ctl->done) return;
  double s[4] = {0, 0, 0, 0};
  for (int i = threadIdx.x; i < nparts; i += 256)
    for (int k = 0; k < 4; ++k) s[k] += w.part_F[4 * i + k];
  for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = s[k];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x < 4) w.stepsum[threadIdx.x] = red[threadIdx.x][0];
}

// Ceres' trust-region decision (TrustRegionMinimizer::Minimize body + LevenbergMarquardtStrategy)
__global__ __launch_bounds__(64) void control_kernel(Ws w, vgg_ba_options opt, int C) {
  Ctl* c = w.ctl;
  if (c->done) { if (threadIdx.x == 0) c->accept = 0; return; }   // (no-op iterations behind the end must not commit again)
  double step_c = 0, xn_c = 0, mcc_c = 0;
#pragma unroll 4
  for (int i = threadIdx.x; i <= C; i += 64) { step_c += w.cam_part[3 * i]; xn_c += w.cam_part[3 * i + 1]; mcc_c += w.cam_part[3 * i + 2]; }
  step_c = wave_sum(step_c); xn_c = wave_sum(xn_c); mcc_c = wave_sum(mcc_c);
  if (threadIdx.x != 0) return;
  const int it = c->iteration;
  vgg_ba_iteration li;
  li.iteration = it; li.successful = 0; li.cost = c->x_cost; li.cost_change = 0; li.gradient_max_norm = c->gmax;
  li.step_norm = 0; li.relative_decrease = 0; li.radius = c->radius;
  c->accept = 0;
  const double cand_cost = 0.5 * w.stepsum[0];
  const double mcc = w.stepsum[1] + mcc_c;         // (points' share, summed over the ranks) + (cameras' share, replicated)
  const double step_norm = sqrt(w.stepsum[2] + step_c);
  const double x_norm = sqrt(w.stepsum[3] + xn_c);
  c->cand_cost = cand_cost; c->mcc = mcc; c->step_norm = step_norm;
  const bool step_valid = !c->linear_fail && (mcc > 0.0);
  c->linear_fail = 0;
  if (!step_valid) {
    if (++c->invalid_streak >= opt.max_num_consecutive_invalid_steps) { c->done = 1; c->termination = 5; }
    c->radius /= c->decrease_factor; c->decrease_factor *= 2.0;
    c->num_unsucc += 1;
    li.radius = c->radius;
    w.log[it] = li;
    return;
  }
  c->invalid_streak = 0;
  li.step_norm = step_norm;
  li.cost_change = c->x_cost - cand_cost;
  const double ptol = opt.parameter_tolerance * (x_norm + opt.parameter_tolerance);
  if (!(step_norm > ptol)) { c->done = 1; c->termination = 3; w.log[it] = li; return; }
  if (fabs(c->x_cost - cand_cost) <= opt.function_tolerance * c->x_cost) { c->done = 1; c->termination = 2; w.log[it] = li; return; }
  const double rel = (c->x_cost - cand_cost) / mcc;
  c->rel = rel;
  li.relative_decrease = rel;
  if (rel > opt.min_relative_decrease) {
    c->accept = 1;
    c->x_cost = cand_cost;
    const double tmp = 2.0 * rel - 1.0;
    c->radius = c->radius / fmax(1.0 / 3.0, 1.0 - tmp * tmp * tmp);
    c->radius = fmin(opt.max_trust_region_radius, c->radius);
    c->decrease_factor = 2.0;
    c->need_lin = 1;
    c->num_succ += 1;
    li.successful = 1; li.cost = cand_cost;
  } else {
    c->radius /= c->decrease_factor; c->decrease_factor *= 2.0;
    c->num_unsucc += 1;
  }
  li.radius = c->radius;
  w.log[it] = li;
}

__global__ void commit_kernel(Ws w, double* cam_q, double* cam_t, double* intr, double* pts, int C, int NI, int P) {
  if (!w.ctl->accept) return;     // accept is cleared by control_kernel on every live iteration
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)3 * P) pts[i] = w.cand_pts[i];
  if (i < (size_t)4 * C) cam_q[i] = w.cand_q[i];
  if (i < (size_t)3 * C) cam_t[i] = w.cand_t[i];
  if (i < (size_t)4 * NI) intr[i] = w.cand_intr[i];
}

// ---------------------------------------------------------------------------------------------
// host side
// Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg).
enum { kProfLinearize = 0, kProfPointPass, kProfCamRhs, kProfSchurTile, kProfCholesky, kProfPointStep,
       kProfSchurTileDiag, kProfCount };   // kProfSchurTile = the off-diagonal launch
struct Profiler {
  bool on = false;
  int cap = 0;
  hipEvent_t* start[kProfCount] = {};
  hipEvent_t* stop[kProfCount] = {};
  int n[kProfCount] = {};
};
static Profiler g_prof;
struct ProfScope {
  int id; hipStream_t st; bool live;
  ProfScope(int id_, hipStream_t st_) : id(id_), st(st_), live(g_prof.on && g_prof.n[id_] < g_prof.cap) {
    if (live) (void)hipEventRecord(g_prof.start[id][g_prof.n[id]], st);
  }
  ~ProfScope() {
    if (live) { (void)hipEventRecord(g_prof.stop[id][g_prof.n[id]], st); g_prof.n[id]++; }
  }
};

struct Launch {
  Dims d; DevProblem dp; Ws w; vgg_ba_options opt; hipStream_t st; int wgB;
  int lpp;                                      // lanes per point of the point passes: 16, 32 or 64 (lanes_per_point)
  const int32_t* chunk_desc; const int32_t* entries; int num_chunks, num_segments;
  int merged_tile_launch;
  const int32_t* tile_desc; int num_tiles;
  const int32_t* batches; int num_batches;      // HOST table [num_batches][6], see vgg_ba_problem.tile_batches
  int chol_split_a, chol_split_b;               // block-diagonal leading part of the reduced system (0 = none)
  const int32_t* chol_first_blk;                // row envelope of the reduced system in 64-column blocks (device) or NULL
  double *cam_q, *cam_t, *intr, *pts;
};

// Multi-GPU payload of the reduced system: only the lower triangle is ever filled, so the all-reduce carries
// n(n+1)/2 + n doubles instead of n^2 + n (phases 4 / 5, called by the distributed host loop only).
// mode 0: S, rhs -> packed (+ the zero tail up to W equal slices, + this rank's gradient maximum behind its slice buffer);
// mode 1: packed -> S, rhs;  mode 2: the all-gather output -> S, rhs (slice r of ceil(count / W) elements sits at
// r (ceil(count / W) + 1), followed by rank r's gradient maximum) and the maximum over the ranks -> gmax_pts.
__global__ __launch_bounds__(256) void pack_lower_kernel(Ws w, int n, int mode) {
  if (w.ctl->done) return;
  const int i = blockIdx.x;                                   // row i, or row n = the right-hand side
  const size_t base = (i < n) ? (size_t)i * (i + 1) / 2 : (size_t)n * (n + 1) / 2;
  const int len = (i < n) ? i + 1 : n;
  double* full = (i < n) ? w.S + (size_t)i * n : w.rhs;
  const int W = w.ctl->world > 0 ? w.ctl->world : 1;
  const size_t chunk = (w.packed_count + W - 1) / W;
  if (mode == 0) {
    for (int j = threadIdx.x; j < len; j += 256) w.packed[base + j] = full[j];
    if (i == n) {
      for (size_t j = w.packed_count + threadIdx.x; j < (size_t)W * chunk; j += 256) w.packed[j] = 0.0;
      if (threadIdx.x == 0) w.pk_mine[chunk] = w.gmax_pts[0];
    }
  } else if (mode == 1) {
    for (int j = threadIdx.x; j < len; j += 256) full[j] = w.packed[base + j];
  } else {
    for (int j = threadIdx.x; j < len; j += 256) {
      const size_t e = base + j, r = e / chunk;
      full[j] = w.pk_gathered[r * (chunk + 1) + (e - r * chunk)];
    }
    if (i == n && threadIdx.x < 64) {
      double m = 0.0;
      for (int r = threadIdx.x; r < W; r += 64) m = fmax(m, w.pk_gathered[(size_t)r * (chunk + 1) + chunk]);
      m = wave_max(m);
      if (threadIdx.x == 0) w.gmax_pts[0] = m;
    }
  }
}

// SPLIT exchange of the sharded solve (round 6, phases 7..11): the packed lower triangle is cut into the part the OFF-DIAGONAL
// tile launch fills (A: elements whose row and column belong to cameras of different 16-camera groups) and the rest (B: the
// diagonal tiles' blocks, everything assemble_kernel adds, the shared-intrinsics border, the right-hand side), so that the
// reduce-scatter + all-gather of A can run on the communicator's stream while the diagonal tile launch is still computing B.
// Row r of S contributes up to three column ranges to A and two to B, in this order; a part is packed row by row.
struct RowParts { int a0[3], a1[3], b0[2], b1[2]; };   // fixed slots; a range with x1 <= x0 is empty
__device__ __forceinline__ RowParts row_parts(int r, int n, int C, int KD, int shared) {
  RowParts p;
  p.a0[0] = p.a1[0] = p.a0[1] = p.a1[1] = p.a0[2] = p.a1[2] = p.b0[0] = p.b1[0] = p.b0[1] = p.b1[1] = 0;
  const int P6 = 6 * C, GW = 6 * kGroup;
  if (r >= n) { p.b1[0] = n; }                                // the right-hand side
  else if (r < P6) { const int g = r / GW; p.a1[0] = GW * g; p.b0[0] = GW * g; p.b1[0] = r + 1; }
  else if (shared || KD == 0) { p.b1[0] = r + 1; }            // shared-intrinsics border: assemble_kernel's
  else {                                                      // intrinsics row of camera a (per-camera intrinsics)
    const int a = (r - P6) / KD, g = a / kGroup;
    const int pb0 = GW * g, pb1 = min(GW * (g + 1), P6), ib0 = P6 + kGroup * KD * g;
    p.a1[0] = pb0; p.b0[0] = pb0; p.b1[0] = pb1; p.a0[1] = pb1; p.a1[1] = P6; p.a0[2] = P6; p.a1[2] = ib0; p.b0[1] = ib0; p.b1[1] = r + 1;
  }
  return p;
}
__device__ __forceinline__ int parts_count(const RowParts& p, int part) {
  return part == 0 ? max(p.a1[0] - p.a0[0], 0) + max(p.a1[1] - p.a0[1], 0) + max(p.a1[2] - p.a0[2], 0)
                   : max(p.b1[0] - p.b0[0], 0) + max(p.b1[1] - p.b0[1], 0);
}
// mode 0: S, rhs -> the part's region of `packed` (zero tail up to W equal slices; part B: + this rank's gradient maximum behind
// its slice);  mode 2: the part's all-gather output -> S, rhs (part B: + the maximum over the ranks -> gmax_pts).
// Regions: packed = [A: W chunkA | B: W chunkB], pk_mine = [A: chunkA | B: chunkB + 1], pk_gathered = [A: W chunkA | B: W (chunkB + 1)].
// (the rows' offsets inside their part: one workgroup, once per workspace -- phase 12; the first version had every workgroup of
//  pack_split_kernel sum the rows above its own: 10 M row classifications per launch at n = 3200, 120 us)
__global__ __launch_bounds__(256) void split_offsets_kernel(Ws w, int n, int C, int KD, int shared) {
  __shared__ unsigned long long part_sum[2][256];
  const int per = (n + 1 + 255) / 256, r0 = threadIdx.x * per, r1 = min(r0 + per, n + 1);
  unsigned long long sa = 0, sb = 0;
  for (int r = r0; r < r1; ++r) { const RowParts p = row_parts(r, n, C, KD, shared); sa += parts_count(p, 0); sb += parts_count(p, 1); }
  part_sum[0][threadIdx.x] = sa; part_sum[1][threadIdx.x] = sb;
  __syncthreads();
  if (threadIdx.x < 2) {
    unsigned long long run = 0;
    for (int t = 0; t < 256; ++t) { const unsigned long long v = part_sum[threadIdx.x][t]; part_sum[threadIdx.x][t] = run; run += v; }
    w.split_off[(size_t)threadIdx.x * (n + 2) + n + 1] = run;
  }
  __syncthreads();
  sa = part_sum[0][threadIdx.x]; sb = part_sum[1][threadIdx.x];
  for (int r = r0; r < r1; ++r) {
    const RowParts p = row_parts(r, n, C, KD, shared);
    w.split_off[r] = sa; w.split_off[(size_t)(n + 2) + r] = sb;
    sa += parts_count(p, 0); sb += parts_count(p, 1);
  }
}
__global__ __launch_bounds__(256) void pack_split_kernel(Ws w, int n, int C, int KD, int shared, int part, int mode) {
  if (w.ctl->done) return;
  const int i = blockIdx.x;                                   // row i, or row n = the right-hand side
  const unsigned long long before = w.split_off[(size_t)part * (n + 2) + i], total = w.split_off[(size_t)part * (n + 2) + n + 1];
  const unsigned long long total_a = w.split_off[n + 1];
  const size_t W = w.ctl->world > 0 ? w.ctl->world : 1;
  const size_t chunk_a = (total_a + W - 1) / W, chunk_b = (w.packed_count - total_a + W - 1) / W;
  const size_t chunk = part == 0 ? chunk_a : chunk_b, gstride = chunk + (part == 0 ? 0 : 1);
  double* packed = w.packed + (part == 0 ? 0 : W * chunk_a);
  const double* gathered = w.pk_gathered + (part == 0 ? 0 : W * chunk_a);
  double* full = (i < n) ? w.S + (size_t)i * n : w.rhs;
  const RowParts p = row_parts(i, n, C, KD, shared);
  size_t e = before;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (part == 1 && k == 2) break;
    const int x0 = part == 0 ? p.a0[k] : p.b0[k < 2 ? k : 0], x1 = part == 0 ? p.a1[k] : p.b1[k < 2 ? k : 0];
    if (x1 <= x0) continue;
    if (mode == 0) {
      for (int j = x0 + threadIdx.x; j < x1; j += 256) packed[e + (size_t)(j - x0)] = full[j];
    } else {
      // (slice of the first element once per thread, then by increments: no 64-bit division per element)
      size_t ee = e + threadIdx.x, r = ee / chunk, off = ee - r * chunk;
      for (int j = x0 + threadIdx.x; j < x1; j += 256) {
        full[j] = gathered[r * gstride + off];
        off += 256;
        while (off >= chunk) { off -= chunk; ++r; }
      }
    }
    e += (size_t)(x1 - x0);
  }
  if (i == n) {
    if (mode == 0) {
      for (size_t j = total + threadIdx.x; j < W * chunk; j += 256) packed[j] = 0.0;
      if (part == 1 && threadIdx.x == 0) w.pk_mine[chunk_a + chunk_b] = w.gmax_pts[0];
    } else if (part == 1 && threadIdx.x < 64) {
      double m = 0.0;
      for (size_t r = threadIdx.x; r < W; r += 64) m = fmax(m, gathered[r * gstride + chunk]);
      m = wave_max(m);
      if (threadIdx.x == 0) w.gmax_pts[0] = m;
    }
  }
}
// (host: number of elements of part A -- the same sum)
static size_t split_count_a(const Dims& d) {
  size_t c = 0;
  const int P6 = 6 * d.C, GW = 6 * kGroup;
  for (int r = 0; r < d.n_red; ++r) {
    if (r < P6) c += (size_t)GW * (r / GW);
    else if (!(d.shared || d.kd == 0)) {
      const int a = (r - P6) / d.kd, g = a / kGroup;
      const int pb0 = GW * g, pb1 = (GW * (g + 1) < P6) ? GW * (g + 1) : P6;
      c += (size_t)pb0 + (size_t)(P6 - pb1) + (size_t)kGroup * d.kd * g;
    }
  }
  return c;
}

// workgroups per camera of the camera passes: ~1024 workgroups in total for a large problem (c3: 19 observations per
// thread; 2048 workgroups cost 0.117 + 0.131 ms, 1024: 0.099 + 0.128, 512: 0.105 + 0.146), at least ~2048 observations per
// workgroup for a small one (c2: 512 workgroups 0.025 + 0.019 ms, 1024: 0.038 + 0.026).  VGG_CAM_WGS overrides the total.
static inline int cam_split_for(int C, int O) {
  const int target = g_tuning.cam_wgs > 0 ? g_tuning.cam_wgs : min(1024, max(256, O / 2048));
  const int s = target / (C > 0 ? C : 1);
  return s < 1 ? 1 : (s > kCamSplitMax ? kCamSplitMax : s);
}

template <int KD>
static void phase_linearize(const Launch& L) {
  ProfScope ps(kProfLinearize, L.st);
  const int split = cam_split_for(L.d.C, L.d.O);
  cam_pass_kernel<KD, 0><<<dim3(L.d.C, split), 256, 0, L.st>>>(L.dp, L.w);
  cam_reduce_kernel<KD, 0><<<L.d.C, 64, 0, L.st>>>(L.dp, L.w, split, 0);
}

// one batch of Schur tiles: the off-diagonal launch, the diagonal launch, and the ordered sum of their chunks into
// dst (S, or S2 for a batch that runs beside the factorisation)
template <int BD>
static void launch_schur_batch(const Launch& L, int batch, hipStream_t st, double* dst, int which = 0) {
  // which: 0 = the whole batch; 1 = the off-diagonal launch and the sums of its tiles; 2 = the diagonal launch and its sums
  // (1 / 2: the split exchange of the sharded solve; never with the merged launch)
  const int32_t* B = L.batches + 6 * batch;
  const int c0 = B[0], cm = B[1], c1 = B[2], t0 = B[3], t1 = B[4];
  if (L.merged_tile_launch && c1 > c0) {
    ProfScope ps(kProfSchurTile, st);
    schur_tile_merged_kernel<BD><<<c1 - c0, 256, 0, st>>>(L.w, L.chunk_desc, L.entries, c0, L.num_segments);
  } else if (cm > c0 && which != 2) {
    if (BD == 6 && L.w.tile_dma) {                 // round-6 A/B: LDS-DMA staging from the expanded segment image
      {
        ProfScope ps(kProfCamRhs, st);              // (the slot of cam_pass<RHS>, which tile_rhs leaves empty: the expansion, timed apart)
        expand_segments_kernel<<<div_up((L.num_segments + 1) * kGroup, 256), 256, 0, st>>>(L.w, L.num_segments);
      }
      ProfScope ps(kProfSchurTile, st);
      if (L.w.tile_dma == 2) schur_tile_dma_kernel<2><<<cm - c0, 256, 0, st>>>(L.w, L.chunk_desc, L.entries, c0, L.num_segments);
      else schur_tile_dma_kernel<1><<<cm - c0, 256, 0, st>>>(L.w, L.chunk_desc, L.entries, c0, L.num_segments);
    } else {
      ProfScope ps(kProfSchurTile, st);
      schur_tile_kernel<BD, false><<<cm - c0, 256, 0, st>>>(L.w, L.chunk_desc, L.entries, c0, L.num_segments);
    }
  }
  if (!L.merged_tile_launch && c1 > cm && which != 1) {
    ProfScope ps(kProfSchurTileDiag, st);
    schur_tile_kernel<BD, true><<<c1 - cm, 256, 0, st>>>(L.w, L.chunk_desc, L.entries, cm, L.num_segments);
  }
  if (t1 > t0)
    tile_reduce_kernel<BD><<<dim3(div_up(kGroup * BD * kGroup * BD, 256) + (L.w.tile_rhs ? (BD == 6 ? div_up(kGroup * BD * 3, 256) : div_up(2 * kGroup * BD, 256)) : 0),
                                  t1 - t0), 256, 0, st>>>(L.w, L.d.n_red, L.d.C, L.d.kd, L.tile_desc, t0, dst, which == 0 ? -1 : which - 1);
}

template <int KD>
static void launch_schur_batches(const Launch& L, int b0, int b1, hipStream_t st, double* dst, hipEvent_t* done_events,
                                 int32_t* done_flags = nullptr, int which = 0) {
  for (int b = b0; b < b1; ++b) {
    if (L.d.shared || KD == 0) launch_schur_batch<6>(L, b, st, dst, which);
    else launch_schur_batch<6 + KD>(L, b, st, dst, which);
    if (done_events) (void)hipEventRecord(done_events[b], st);
    if (done_flags) dataflow_signal(done_flags + b, st);    // (for the single-launch factorisation, which waits on the device)
  }
}

// Overlap mode (options.overlap_factorization = number of CUs given to the factorisation, a multiple of 32; >= 2
// tile batches): two CU-masked streams split the chip while batches 1.. and the factorisation run side by side.
// Without the masks the two only overlap by accident (the tile launches fill every CU); see DESIGN.md section 6.
struct OverlapCtx {
  int chol_cus = 0;
  hipStream_t st_chol = nullptr, st_rest = nullptr;
  hipEvent_t ready = nullptr, chol_done = nullptr, batch_done[8] = {};
};
static OverlapCtx* overlap_ctx(const Launch& L) {
  static OverlapCtx ctx[16];
  static bool failed[16] = {};
  const int cus = L.opt.overlap_factorization;
  if (cus <= 0 || cus % 32 != 0 || L.num_batches < 2 || L.num_batches > 8 || L.num_chunks <= 0) return nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || failed[dev]) return nullptr;
  OverlapCtx& c = ctx[dev];
  if (c.chol_cus == cus) return &c;
  if (c.chol_cus != 0) return nullptr;             // one split per process
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= cus || prop.multiProcessorCount % 32 != 0) {
    failed[dev] = true; return nullptr;
  }
  const int words = prop.multiProcessorCount / 32, cw = cus / 32;
  uint32_t mask_chol[16] = {}, mask_rest[16] = {};
  if (words > 16) { failed[dev] = true; return nullptr; }
  for (int i = 0; i < words; ++i) { mask_chol[i] = (i < cw) ? 0xFFFFFFFFu : 0u; mask_rest[i] = (i < cw) ? 0u : 0xFFFFFFFFu; }
  bool ok = hipExtStreamCreateWithCUMask(&c.st_chol, words, mask_chol) == hipSuccess &&
            hipExtStreamCreateWithCUMask(&c.st_rest, words, mask_rest) == hipSuccess &&
            hipEventCreateWithFlags(&c.ready, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&c.chol_done, hipEventDisableTiming) == hipSuccess;
  for (int i = 0; ok && i < 8; ++i) ok = hipEventCreateWithFlags(&c.batch_done[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) { (void)hipGetLastError(); failed[dev] = true; return nullptr; }
  c.chol_cus = cus;
  return &c;
}

// which: 0 = the whole phase; 1 = up to the off-diagonal tile launch and the sums of its tiles; 2 = the diagonal launch, its
// sums and assemble (1 / 2: the split exchange of the sharded solve, phases 7 / 9 -- one tile batch, separate launches)
template <int KD>
static void phase_schur(const Launch& L, int which = 0) {
  const Dims& d = L.d;
  if (which == 2) {
    if (L.num_chunks > 0) launch_schur_batches<KD>(L, 0, L.num_batches, L.st, L.w.S, nullptr, nullptr, 2);
    assemble_kernel<KD><<<d.C + 1, 64, 0, L.st>>>(L.dp, L.w, L.tile_desc, L.num_tiles, L.wgB);
    return;
  }
  prep_kernel<KD><<<1, 256, 0, L.st>>>(L.dp, L.w, L.opt);
  {
    ProfScope ps(kProfPointPass, L.st);
    // cameras (q, t, pose scales, constant flags: 14 doubles each) cached in LDS when they fit beside 2 workgroups/CU
    const size_t cam_lds = sizeof(double) * 19 * (size_t)d.C;
    {
      auto launch_cy = [&](auto lpp, auto cy) {
        constexpr int LPP = decltype(lpp)::value;
        constexpr bool CY = decltype(cy)::value;
        const bool longt = long_tracks(LPP, L.d.P, L.d.O);
        // (two workgroups per CU: the compressed variant keeps 25 KB of record staging beside the camera table)
        const size_t cam_cap = (CY ? 53 : 64) * 1024;
        if (longt && LPP <= 32) {
          if (cam_lds <= cam_cap) point_pass_kernel<KD, true, CY, LPP, (LPP <= 32)><<<L.wgB, 256, cam_lds, L.st>>>(L.dp, L.w, L.opt);
          else point_pass_kernel<KD, false, CY, LPP, (LPP <= 32)><<<L.wgB, 256, 0, L.st>>>(L.dp, L.w, L.opt);
        } else if (cam_lds <= cam_cap) point_pass_kernel<KD, true, CY, LPP><<<L.wgB, 256, cam_lds, L.st>>>(L.dp, L.w, L.opt);
        else point_pass_kernel<KD, false, CY, LPP><<<L.wgB, 256, 0, L.st>>>(L.dp, L.w, L.opt);
      };
      // compressed Schur factors with 6 x 6 tile blocks (shared or constant intrinsics), full factors otherwise
      auto launch = [&](auto lpp) {
        if constexpr (KD == 0) launch_cy(lpp, std::true_type{});
        else if (d.shared) launch_cy(lpp, std::true_type{});
        else launch_cy(lpp, std::false_type{});
      };
      if (L.lpp == 8) launch(std::integral_constant<int, 8>{});
      else if (L.lpp == 16) launch(std::integral_constant<int, 16>{});
      else if (L.lpp == 32) launch(std::integral_constant<int, 32>{});
      else launch(std::integral_constant<int, 64>{});
    }
  }
  if (!L.w.tile_rhs) {
    ProfScope ps(kProfCamRhs, L.st);
    const int split = cam_split_for(d.C, d.O);
    cam_pass_kernel<KD, 1><<<dim3(d.C, split), 256, 0, L.st>>>(L.dp, L.w);
    cam_reduce_kernel<KD, 1><<<d.C, 64, 0, L.st>>>(L.dp, L.w, split, L.wgB);
  }
  // (the reduced system was zeroed by cam_pass_kernel<KD, 1>: no fill launch)
  if (which == 1) {
    if (L.num_chunks > 0) launch_schur_batches<KD>(L, 0, L.num_batches, L.st, L.w.S, nullptr, nullptr, 1);
    return;
  }
  if (L.num_chunks > 0) {
    if (overlap_ctx(L)) {
      // only the first batch here; the others are enqueued by phase_step beside the factorisation and land in S2
      (void)hipMemsetAsync(L.w.S2, 0, sizeof(double) * (size_t)d.n_red * d.n_red, L.st);
      launch_schur_batches<KD>(L, 0, 1, L.st, L.w.S, nullptr);
    } else {
      launch_schur_batches<KD>(L, 0, L.num_batches, L.st, L.w.S, nullptr);
    }
  }
  assemble_kernel<KD><<<d.C + 1, 64, 0, L.st>>>(L.dp, L.w, L.tile_desc, L.num_tiles, L.wgB);
}

template <int KD>
static int phase_step(const Launch& L) {
  const Dims& d = L.d;
  size_t chol_flag_count = 0;
  int32_t* chol_flags = cholesky_dataflow_flags(L.w.chol_inv, d.n_red, &chol_flag_count);
  begin_iteration_kernel<<<1, 256, 0, L.st>>>(L.w, L.opt, d.n_red, chol_flags, chol_flags ? (int)chol_flag_count : 0);
  int rc;
  if (OverlapCtx* oc = overlap_ctx(L)) {
    (void)hipMemsetAsync(L.w.batch_flags, 0, 64, L.st);
    (void)hipEventRecord(oc->ready, L.st);
    (void)hipStreamWaitEvent(oc->st_rest, oc->ready, 0);
    (void)hipStreamWaitEvent(oc->st_chol, oc->ready, 0);
    launch_schur_batches<KD>(L, 1, L.num_batches, oc->st_rest, L.w.S2, oc->batch_done, L.w.batch_flags);
    CholOverlap ov;
    ov.dev_flags = L.w.batch_flags + 1;            // flag of wait k = batch k + 1
    ov.S2 = L.w.S2;
    ov.first_col = 6 * kGroup * L.batches[6 * 1 + 5];
    ov.num_waits = L.num_batches - 1;
    for (int b = 1; b < L.num_batches; ++b) {
      ov.wait_col[b - 1] = 6 * kGroup * L.batches[6 * b + 5];
      ov.wait_ev[b - 1] = oc->batch_done[b];
    }
    {
      ProfScope ps(kProfCholesky, oc->st_chol);
      rc = cholesky_solve_enqueue(L.w.S, L.w.rhs, d.n_red, L.w.chol_inv, &L.w.ctl->linear_fail, &L.w.ctl->done, oc->st_chol, &ov,
                                  L.chol_split_a, L.chol_split_b, L.chol_first_blk, chol_flags != nullptr);
    }
    (void)hipEventRecord(oc->chol_done, oc->st_chol);
    (void)hipStreamWaitEvent(L.st, oc->chol_done, 0);
  } else {
    ProfScope ps(kProfCholesky, L.st);
    rc = cholesky_solve_enqueue(L.w.S, L.w.rhs, d.n_red, L.w.chol_inv, &L.w.ctl->linear_fail, &L.w.ctl->done, L.st, nullptr,
                                L.chol_split_a, L.chol_split_b, L.chol_first_blk, chol_flags != nullptr);
  }
  if (rc != VGG_OK) return rc;
  cam_update_kernel<KD><<<div_up(d.C + 1, 64), 64, 0, L.st>>>(L.dp, L.w);
  {
    ProfScope ps(kProfPointStep, L.st);
    const size_t cam_lds = sizeof(double) * 31 * (size_t)d.C;
    auto launch = [&](auto lpp) {
      constexpr int LPP = decltype(lpp)::value;
      const bool longt = long_tracks(LPP, L.d.P, L.d.O);
      auto go = [&](auto fy) {
        constexpr int FY = decltype(fy)::value;
        const size_t lds = FY == 1 ? sizeof(double) * 18 * (size_t)d.C : cam_lds;
        if (longt && LPP <= 32) {
          if (lds <= 64 * 1024) point_step_kernel<KD, true, LPP, (LPP <= 32), FY><<<L.wgB, 256, lds, L.st>>>(L.dp, L.w);
          else point_step_kernel<KD, false, LPP, (LPP <= 32), FY><<<L.wgB, 256, 0, L.st>>>(L.dp, L.w);
        } else if (lds <= 64 * 1024) point_step_kernel<KD, true, LPP, false, FY><<<L.wgB, 256, lds, L.st>>>(L.dp, L.w);
        else point_step_kernel<KD, false, LPP, false, FY><<<L.wgB, 256, 0, L.st>>>(L.dp, L.w);
      };
      if (L.w.step_from_factors == 1) go(std::integral_constant<int, 1>{});
      else if (L.w.step_from_factors == 2) go(std::integral_constant<int, 2>{});
      else go(std::integral_constant<int, 0>{});
    };
    if (L.lpp == 8) launch(std::integral_constant<int, 8>{});
    else if (L.lpp == 16) launch(std::integral_constant<int, 16>{});
    else if (L.lpp == 32) launch(std::integral_constant<int, 32>{});
    else launch(std::integral_constant<int, 64>{});
  }
  reduce_step_kernel<<<1, 256, 0, L.st>>>(L.w, L.wgB);
  return VGG_OK;
}

static void phase_update(const Launch& L) {
  const Dims& d = L.d;
  control_kernel<<<1, 64, 0, L.st>>>(L.w, L.opt, d.C);
  size_t nmax = (size_t)3 * d.P;
  if ((size_t)4 * d.C > nmax) nmax = (size_t)4 * d.C;
  if ((size_t)4 * d.NI > nmax) nmax = (size_t)4 * d.NI;
  commit_kernel<<<div_up((long)nmax, 256), 256, 0, L.st>>>(L.w, L.cam_q, L.cam_t, L.intr, L.pts, d.C, d.NI, d.P);
}

// compute units of the current device (256 on MI355X), looked up once per device
static int device_cus() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

static int make_launch(const vgg_ba_problem* pb, const vgg_ba_options* opt, void* workspace, hipStream_t st, Launch* L) {
  if (!pb || !opt || !workspace) return VGG_ERR_INVALID_ARGUMENT;
  if (pb->num_cams <= 0 || pb->num_pts < 0 || pb->num_obs < 0) return VGG_ERR_INVALID_ARGUMENT;
  if (pb->num_intr != 1 && pb->num_intr != pb->num_cams) return VGG_ERR_UNSUPPORTED;
  if (pb->camera_model != kPinhole && pb->camera_model != kSimpleRadial) return VGG_ERR_UNSUPPORTED;
  L->d = make_dims(pb);
  L->dp = dev_problem(pb, L->d);
  L->w = carve(L->d, opt->max_num_iterations, pb->num_chunks, pb->num_segments, workspace);
  L->opt = *opt;
  L->st = st;
  L->lpp = lanes_per_point(L->d.P, L->d.O);
  // (every workgroup fills its LDS camera cache first: ONE resident round of the point passes -- two workgroups per CU -- is the
  //  cheapest; round 4, configs[2]: 512 workgroups point_pass + point_step 0.352 ms, 768: 0.415, 1024: 0.359, 2048: 0.371)
  L->wgB = min(max(div_up(L->d.P, 4 * (64 / L->lpp)), 1), g_tuning.point_wgs > 0 ? min(g_tuning.point_wgs, kMaxWG) : min(2 * device_cus(), kMaxWG));
  L->chunk_desc = pb->chunk_desc; L->entries = pb->entries; L->num_chunks = pb->num_chunks;
  L->merged_tile_launch = pb->merged_tile_launch;
  L->num_segments = pb->num_segments;
  L->batches = pb->tile_batches; L->num_batches = pb->num_tile_batches;
  L->chol_split_a = pb->chol_split_a; L->chol_split_b = pb->chol_split_b;
  L->chol_first_blk = pb->chol_first_blk;
  if (pb->chol_split_a < 0 || pb->chol_split_b < 0 || pb->chol_split_a % 64 != 0 ||
      pb->chol_split_a + pb->chol_split_b > 6 * pb->num_cams)
    return VGG_ERR_INVALID_ARGUMENT;
  if (pb->num_chunks > 0) {
    if (!pb->tile_batches || pb->num_tile_batches < 1) return VGG_ERR_INVALID_ARGUMENT;
    int prev_c = 0, prev_t = 0, prev_g = 0;
    for (int b = 0; b < pb->num_tile_batches; ++b) {   // consecutive, ordered ranges
      const int32_t* B = pb->tile_batches + 6 * b;
      if (B[0] != prev_c || B[1] < B[0] || B[2] < B[1] || B[3] != prev_t || B[4] < B[3] || B[5] < prev_g) return VGG_ERR_INVALID_ARGUMENT;
      prev_c = B[2]; prev_t = B[4]; prev_g = B[5];
    }
    if (prev_c != pb->num_chunks || prev_t != pb->num_tiles) return VGG_ERR_INVALID_ARGUMENT;
  }
  L->tile_desc = pb->tile_desc; L->num_tiles = pb->num_tiles;
  // reduced right-hand side from the diagonal tile launch instead of cam_pass<RHS>: compressed 6 x 6 tile blocks (shared or
  // constant intrinsics), one tile batch (the overlap mode needs the right-hand side before its later batches have run)
  // (round 6: also with per-camera intrinsics -- 7 x 7 / 8 x 8 blocks, full factors; g_tuning.tile_rhs == 1 keeps those on the
  //  camera pass, 2 = everywhere)
  L->w.tile_rhs = (g_tuning.tile_rhs && (L->d.shared || L->d.kd == 0 || g_tuning.tile_rhs >= 2) && pb->num_chunks > 0 &&
                   pb->num_tile_batches == 1) ? 1 : 0;
  L->w.step_from_factors = (L->d.shared || L->d.kd == 0) ? g_tuning.step_factors : 0;      // (compressed factors in the segment buffer)
  L->w.tile_dma = (L->w.Yx && !pb->merged_tile_launch && pb->num_tile_batches == 1) ? g_tuning.tile_dma : 0;
  L->cam_q = pb->cam_q; L->cam_t = pb->cam_t; L->intr = pb->intr; L->pts = pb->pts;
  return VGG_OK;
}

template <typename Fn0, typename Fn1, typename Fn2>
static int dispatch_kd(int kd, Fn0 f0, Fn1 f1, Fn2 f2) {
  switch (kd) { case 0: return f0(); case 1: return f1(); default: return f2(); }
}

static int run_phase(const Launch& L, int phase) {
  switch (phase) {
    case 0:
      return dispatch_kd(L.d.kd, [&] { phase_linearize<0>(L); return VGG_OK; }, [&] { phase_linearize<1>(L); return VGG_OK; },
                         [&] { phase_linearize<2>(L); return VGG_OK; });
    case 1:
      return dispatch_kd(L.d.kd, [&] { phase_schur<0>(L); return VGG_OK; }, [&] { phase_schur<1>(L); return VGG_OK; },
                         [&] { phase_schur<2>(L); return VGG_OK; });
    case 2:
      return dispatch_kd(L.d.kd, [&] { return phase_step<0>(L); }, [&] { return phase_step<1>(L); }, [&] { return phase_step<2>(L); });
    case 3: phase_update(L); return VGG_OK;
    case 4: pack_lower_kernel<<<L.d.n_red + 1, 256, 0, L.st>>>(L.w, L.d.n_red, 0); return VGG_OK;
    case 5: pack_lower_kernel<<<L.d.n_red + 1, 256, 0, L.st>>>(L.w, L.d.n_red, 1); return VGG_OK;
    case 6: pack_lower_kernel<<<L.d.n_red + 1, 256, 0, L.st>>>(L.w, L.d.n_red, 2); return VGG_OK;
    // split exchange (round 6): 7 = phase 1 up to the sums of the off-diagonal tiles, 8 = pack part A, 9 = the rest of phase 1,
    // 10 = pack part B, 11 = both parts' gathered slices -> S | rhs; 12 = "is the split available for this problem?" + the rows' offsets inside the parts (call it once per workspace)
    case 7: case 9: case 12:
      if (L.num_batches != 1 || L.merged_tile_launch || L.d.C <= kGroup) return VGG_ERR_UNSUPPORTED;   // (one camera group: no off-diagonal part)
      if (phase == 12) {                           // (also prepares the rows' offsets inside the two parts: once per workspace)
        split_offsets_kernel<<<1, 256, 0, L.st>>>(L.w, L.d.n_red, L.d.C, L.d.kd, L.d.shared);
        return VGG_OK;
      }
      return dispatch_kd(L.d.kd, [&] { phase_schur<0>(L, phase == 7 ? 1 : 2); return VGG_OK; },
                         [&] { phase_schur<1>(L, phase == 7 ? 1 : 2); return VGG_OK; },
                         [&] { phase_schur<2>(L, phase == 7 ? 1 : 2); return VGG_OK; });
    case 8: case 10:
      pack_split_kernel<<<L.d.n_red + 1, 256, 0, L.st>>>(L.w, L.d.n_red, L.d.C, L.d.kd, L.d.shared, phase == 8 ? 0 : 1, 0);
      return VGG_OK;
    case 11:
      pack_split_kernel<<<L.d.n_red + 1, 256, 0, L.st>>>(L.w, L.d.n_red, L.d.C, L.d.kd, L.d.shared, 0, 2);
      pack_split_kernel<<<L.d.n_red + 1, 256, 0, L.st>>>(L.w, L.d.n_red, L.d.C, L.d.kd, L.d.shared, 1, 2);
      return VGG_OK;
    default: return VGG_ERR_INVALID_ARGUMENT;
  }
}

static int finish(const Launch& L, int max_iters, vgg_ba_summary* summary, vgg_ba_iteration* log, int log_cap) {
  Ctl h;
  if (hipMemcpyAsync(&h, L.w.ctl, sizeof(Ctl), hipMemcpyDeviceToHost, L.st) != hipSuccess) return VGG_ERR_HIP;
  int ncopy = 0;
  if (log && log_cap > 0) {
    ncopy = (max_iters + 1 < log_cap) ? max_iters + 1 : log_cap;
    if (hipMemcpyAsync(log, L.w.log, sizeof(vgg_ba_iteration) * ncopy, hipMemcpyDeviceToHost, L.st) != hipSuccess) return VGG_ERR_HIP;
  }
  if (hipStreamSynchronize(L.st) != hipSuccess) return VGG_ERR_HIP;
  if (summary) {
    summary->initial_cost = h.initial_cost; summary->final_cost = h.x_cost;
    summary->num_iterations = h.iteration; summary->num_successful_steps = h.num_succ;
    summary->num_unsuccessful_steps = h.num_unsucc; summary->termination = h.termination;
    summary->n_reduced = L.d.n_red;
    summary->num_log = (h.iteration + 1 < ncopy) ? h.iteration + 1 : ncopy;
  }
  return VGG_OK;
}

}  // namespace vgg

using namespace vgg;

extern "C" {

int vgg_ba_tuning(int lanes_per_point, int long_tracks, int cam_workgroups, int point_workgroups) {
  if (!(lanes_per_point == 0 || lanes_per_point == 8 || lanes_per_point == 16 || lanes_per_point == 32 || lanes_per_point == 64))
    return VGG_ERR_INVALID_ARGUMENT;
  vgg::g_tuning = vgg::Tuning{lanes_per_point, long_tracks < 0 ? -1 : (long_tracks ? 1 : 0), cam_workgroups > 0 ? cam_workgroups : 0,
                              point_workgroups > 0 ? point_workgroups : 0, vgg::g_tuning.tile_rhs, vgg::g_tuning.step_factors,
                              vgg::g_tuning.tile_dma};
  return VGG_OK;
}

int vgg_ba_set_tile_rhs(int enable) {
  vgg::g_tuning.tile_rhs = (enable >= 0 && enable <= 2) ? enable : 1;
  return VGG_OK;
}

int vgg_ba_set_tile_dma(int mode) {
  vgg::g_tuning.tile_dma = (mode >= 0 && mode <= 2) ? mode : 0;
  return VGG_OK;
}

int vgg_ba_set_step_from_factors(int enable) {
  vgg::g_tuning.step_factors = (enable == 1 || enable == 2) ? enable : 0;
  return VGG_OK;
}

int vgg_ba_profile(int enable, int max_launches_per_kernel) {
  for (int k = 0; k < kProfCount; ++k) {
    for (int i = 0; i < g_prof.cap; ++i) { (void)hipEventDestroy(g_prof.start[k][i]); (void)hipEventDestroy(g_prof.stop[k][i]); }
    delete[] g_prof.start[k]; delete[] g_prof.stop[k];
    g_prof.start[k] = g_prof.stop[k] = nullptr; g_prof.n[k] = 0;
  }
  g_prof.cap = 0; g_prof.on = false;
  if (!enable) return VGG_OK;
  if (max_launches_per_kernel <= 0) return VGG_ERR_INVALID_ARGUMENT;
  for (int k = 0; k < kProfCount; ++k) {
    g_prof.start[k] = new hipEvent_t[max_launches_per_kernel];
    g_prof.stop[k] = new hipEvent_t[max_launches_per_kernel];
    for (int i = 0; i < max_launches_per_kernel; ++i) {
      VGG_HIP_CHECK(hipEventCreate(&g_prof.start[k][i]));
      VGG_HIP_CHECK(hipEventCreate(&g_prof.stop[k][i]));
    }
  }
  g_prof.cap = max_launches_per_kernel; g_prof.on = true;
  return VGG_OK;
}

int vgg_ba_profile_read(int kernel_id, double* total_ms, int* launches, int reset) {
  if (kernel_id < 0 || kernel_id >= kProfCount || !total_ms || !launches) return VGG_ERR_INVALID_ARGUMENT;
  double tot = 0;
  for (int i = 0; i < g_prof.n[kernel_id]; ++i) {
    VGG_HIP_CHECK(hipEventSynchronize(g_prof.stop[kernel_id][i]));
    float ms = 0;
    VGG_HIP_CHECK(hipEventElapsedTime(&ms, g_prof.start[kernel_id][i], g_prof.stop[kernel_id][i]));
    tot += ms;
  }
  *total_ms = tot; *launches = g_prof.n[kernel_id];
  if (reset) g_prof.n[kernel_id] = 0;
  return VGG_OK;
}

const char* vgg_build_arch(void) { return "gfx950"; }
int vgg_abi_version(void) { return VGG_ABI_VERSION; }
size_t vgg_abi_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(vgg_ba_problem);
    case 1: return sizeof(vgg_ba_options);
    case 2: return sizeof(vgg_ba_iteration);
    case 3: return sizeof(vgg_ba_summary);
    default: return 0;
  }
}

size_t vgg_ba_workspace_bytes(const vgg_ba_problem* problem, const vgg_ba_options* options) {
  if (!problem || !options) return 0;
  const Dims d = make_dims(problem);
  return carve(d, options->max_num_iterations, problem->num_chunks, problem->num_segments, nullptr).total_bytes + 256;
}

int vgg_ba_begin(const vgg_ba_problem* problem, const vgg_ba_options* options, void* workspace, size_t workspace_bytes,
                 int rank, int world_size, void* stream) {
  Launch L;
  int rc = make_launch(problem, options, workspace, (hipStream_t)stream, &L);
  if (rc != VGG_OK) return rc;
  if (workspace_bytes < L.w.total_bytes) return VGG_ERR_WORKSPACE;
  if (world_size < 1 || world_size > kPackPad || rank < 0 || rank >= world_size) return VGG_ERR_INVALID_ARGUMENT;
  VGG_HIP_CHECK(hipMemsetAsync(L.w.Y, 0, L.w.y_bytes, L.st));   // absent slots stay zero for the whole solve
  init_kernel<<<div_up(L.d.n_red > 0 ? L.d.n_red : 1, 256), 256, 0, L.st>>>(L.dp, L.w, L.opt, rank, world_size);
  VGG_LAUNCH_CHECK();
  return VGG_OK;
}

int vgg_ba_phase(const vgg_ba_problem* problem, const vgg_ba_options* options, void* workspace, int phase, void* stream) {
  Launch L;
  int rc = make_launch(problem, options, workspace, (hipStream_t)stream, &L);
  if (rc != VGG_OK) return rc;
  rc = run_phase(L, phase);
  if (rc != VGG_OK) return rc;
  VGG_LAUNCH_CHECK();
  return VGG_OK;
}

int vgg_ba_reduce_buffer(const vgg_ba_problem* problem, const vgg_ba_options* options, void* workspace, int which,
                         double** device_ptr, size_t* count) {
  if (!problem || !options || !workspace || !device_ptr || !count) return VGG_ERR_INVALID_ARGUMENT;
  const Dims d = make_dims(problem);
  Ws w = carve(d, options->max_num_iterations, problem->num_chunks, problem->num_segments, workspace);
  switch (which) {
    case 0: *device_ptr = w.lin; *count = w.lin_count; break;
    case 1: *device_ptr = w.sys; *count = w.sys_count; break;
    case 2: *device_ptr = w.gmax_pts; *count = 1; break;
    case 3: *device_ptr = w.stepsum; *count = 4; break;
    case 4: *device_ptr = w.packed; *count = w.packed_count; break;
    case 5: *device_ptr = w.pk_mine; *count = w.packed_count + 2; break;
    case 6: *device_ptr = w.pk_gathered; *count = w.packed_count + 2 * kPackPad; break;
    case 7: *device_ptr = w.packed; *count = split_count_a(d); break;      // (count = elements of part A of the split exchange)
    default: return VGG_ERR_INVALID_ARGUMENT;
  }
  return VGG_OK;
}

int vgg_ba_finish(const vgg_ba_problem* problem, const vgg_ba_options* options, void* workspace,
                  vgg_ba_summary* summary, vgg_ba_iteration* log, int log_cap, void* stream) {
  Launch L;
  int rc = make_launch(problem, options, workspace, (hipStream_t)stream, &L);
  if (rc != VGG_OK) return rc;
  return finish(L, options->max_num_iterations, summary, log, log_cap);
}

int vgg_ba_poll_done(const vgg_ba_problem* problem, const vgg_ba_options* options, void* workspace, int32_t* done_host,
                     void* stream) {
  if (!done_host) return VGG_ERR_INVALID_ARGUMENT;
  Launch L;
  int rc = make_launch(problem, options, workspace, (hipStream_t)stream, &L);
  if (rc != VGG_OK) return rc;
  int32_t done = 0;
  VGG_HIP_CHECK(hipMemcpyAsync(&done, &L.w.ctl->done, sizeof(done), hipMemcpyDeviceToHost, L.st));
  VGG_HIP_CHECK(hipStreamSynchronize(L.st));
  *done_host = done;
  return VGG_OK;
}

int vgg_ba_solve(const vgg_ba_problem* problem, const vgg_ba_options* options, void* workspace,
                 size_t workspace_bytes, vgg_ba_summary* summary, vgg_ba_iteration* log, int log_cap, void* stream) {
  int rc = vgg_ba_begin(problem, options, workspace, workspace_bytes, 0, 1, stream);
  if (rc != VGG_OK) return rc;
  Launch L;
  rc = make_launch(problem, options, workspace, (hipStream_t)stream, &L);
  if (rc != VGG_OK) return rc;
  // iteration max_num_iterations+1 only runs the start-of-iteration checks (gradient test after the last step).
  // Termination is decided on the device; iterations enqueued after it are no-ops, but ~100 empty launches each.
  // Every kPoll iterations the host therefore reads the `done` flag (one 4-byte copy + stream sync; the queue
  // refills within microseconds) and stops enqueuing once the solve has terminated.
  constexpr int kPoll = 8;
  for (int it = 0; it <= options->max_num_iterations; ++it) {
    for (int phase = 0; phase < 4; ++phase) {
      rc = run_phase(L, phase);
      if (rc != VGG_OK) return rc;
    }
    if (it % kPoll == kPoll - 1 && it < options->max_num_iterations) {
      int32_t done = 0;
      if (hipMemcpyAsync(&done, &L.w.ctl->done, sizeof(done), hipMemcpyDeviceToHost, L.st) != hipSuccess) return VGG_ERR_HIP;
      if (hipStreamSynchronize(L.st) != hipSuccess) return VGG_ERR_HIP;
      if (done) break;
    }
  }
  VGG_LAUNCH_CHECK();
  return finish(L, options->max_num_iterations, summary, log, log_cap);
}

}  // extern "C"

#if VGG_TILE_TRACE
extern "C" int vgg_debug_read_tile_trace(long long* host, size_t count) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(vgg::g_tile_trace), sizeof(long long) * count);
}
#endif
