// Batched absolute-pose refinement: one workgroup runs the whole Levenberg-Marquardt solve of one frame.
//
// Replaces the per-frame Python loops around pycolmap.pose_refinement in the reference
//   init_refine_pose   vggsfm/utils/triangulation.py:482-647 (loop :542-608)
//   refine_pose        vggsfm/utils/triangulation.py:260-479 (loop :341-441)
//   align_next_window  vggsfm/runners/video_runner.py:941-1017
// i.e. COLMAP's RefineAbsolutePose: 3D points constant, CauchyLoss(1), unknowns = pose (quaternion
// manifold + translation) and optionally focal / extra parameter, Ceres trust-region LM with
// gradient_tolerance 1.0, 100 iterations.  The reference pays S Python iterations,
// S device->host copies and S single-threaded Ceres solves; here all frames are solved concurrently,
// observations are read straight from the dense (S,P) track tensor (coalesced rows), the 8x8 normal
// equations live in LDS and nothing leaves the device until the results are ready.
#include "camera_model.hpp"
#include "../../include/vggsfm_amd.h"

namespace vgg {

constexpr int kPN = 8;                       // max unknowns: 6 pose + focal + extra
constexpr int kPH = kPN * (kPN + 1) / 2;     // 36 unique entries of J^T J
constexpr int kPV = kPH + kPN + 1;           // + gradient + cost

struct PoseState { double q[4], t[3], in4[4]; };

// One pass over the observations of frame s at state x: cost, and (LIN) J^T J / J^T r, block-reduced
// into out[kPV] (LDS).  Jacobian columns: 0-2 rotation tangent, 3-5 translation, 6 focal, 7 extra.
template <typename TrackT, bool LIN>
__device__ __forceinline__ void pose_pass(const double* __restrict__ pts, const TrackT* __restrict__ tr,
                                          const uint8_t* __restrict__ mask, int P, int model, int loss, double loss_scale,
                                          const PoseState& x, double* red /* [4][kPV] */, double* out /* [kPV] */) {
  double acc[kPV];
#pragma unroll
  for (int i = 0; i < kPV; ++i) acc[i] = 0.0;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    if (!mask[p]) continue;
    const double X[3] = {pts[3 * p], pts[3 * p + 1], pts[3 * p + 2]};
    const double u = (double)tr[2 * (size_t)p], v = (double)tr[2 * (size_t)p + 1];
    double r[2];
    if (LIN) {
      double Jp[12], Ji[4], Jx[6], F[2 * kPN];
      obs_eval(model, x.q, x.t, x.in4, X, u, v, r, Jp, Ji, Jx);
#pragma unroll
      for (int row = 0; row < 2; ++row) {
#pragma unroll
        for (int k = 0; k < 6; ++k) F[row * kPN + k] = Jp[row * 6 + k];
        F[row * kPN + 6] = Ji[row * 2]; F[row * kPN + 7] = Ji[row * 2 + 1];
      }
      const double s = r[0] * r[0] + r[1] * r[1];
      double rho[3];
      loss_eval(loss, loss_scale, s, rho);
      if (loss != kLossTrivial) {
        Corrector c(s, rho);
        c.jac<kPN>(r, F);
        r[0] *= c.residual_scaling; r[1] *= c.residual_scaling;
      }
      int q = 0;
#pragma unroll
      for (int i = 0; i < kPN; ++i)
#pragma unroll
        for (int k = i; k < kPN; ++k) acc[q++] += F[i] * F[k] + F[kPN + i] * F[kPN + k];
#pragma unroll
      for (int i = 0; i < kPN; ++i) acc[kPH + i] += F[i] * r[0] + F[kPN + i] * r[1];
      acc[kPH + kPN] += rho[0];
    } else {
      obs_residual(model, x.q, x.t, x.in4, X, u, v, r);
      double rho[3];
      loss_eval(loss, loss_scale, r[0] * r[0] + r[1] * r[1], rho);
      acc[kPH + kPN] += rho[0];
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
#pragma unroll
  for (int i = LIN ? 0 : kPH + kPN; i < kPV; ++i) {
    const double sv = wave_sum(acc[i]);
    if (lane == 0) red[wave * kPV + i] = sv;
  }
  __syncthreads();
  if (threadIdx.x < kPV) out[threadIdx.x] = red[threadIdx.x] + red[kPV + threadIdx.x] + red[2 * kPV + threadIdx.x] + red[3 * kPV + threadIdx.x];
  __syncthreads();
}

__device__ __forceinline__ double sym_at(const double* h, int i, int j) {   // packed upper triangle, row-major
  if (i > j) { const int t = i; i = j; j = t; }
  return h[i * kPN - i * (i - 1) / 2 + (j - i)];
}

__device__ __forceinline__ void plus_state(const PoseState& x, const double* d, PoseState& o) {
  quat_plus(x.q, d, o.q);
  for (int k = 0; k < 3; ++k) o.t[k] = x.t[k] + d[3 + k];
  o.in4[0] = x.in4[0] + d[6]; o.in4[1] = x.in4[1]; o.in4[2] = x.in4[2]; o.in4[3] = x.in4[3] + d[7];
}

template <typename TrackT>
__global__ __launch_bounds__(256) void pose_refine_kernel(
    const double* __restrict__ pts, const TrackT* __restrict__ tracks, const uint8_t* __restrict__ inlier, int S, int P,
    const int32_t* __restrict__ frame_ids, double* __restrict__ cam_q, double* __restrict__ cam_t,
    double* __restrict__ intr, int model, const uint8_t* __restrict__ refine_flags, vgg_ba_options opt, int loss,
    double loss_scale, vgg_ba_summary* __restrict__ summaries) {
  __shared__ double red[4 * kPV];
  __shared__ double lin[kPV];        // J^T J (packed), J^T r, cost of the current linearisation point
  __shared__ double cand[kPV];
  __shared__ double sh_delta[kPN];
  __shared__ int sh_flag[4];         // 0 done, 1 accept
  __shared__ PoseState sx, sc;
  const int f = frame_ids[blockIdx.x];
  const TrackT* tr = tracks + (size_t)f * P * 2;
  const uint8_t* mk = inlier + (size_t)f * P;
  const unsigned rf = refine_flags ? refine_flags[f] : 0u;
  bool act[kPN];
  for (int k = 0; k < 6; ++k) act[k] = true;
  act[6] = (rf & 1u) != 0;
  act[7] = (rf & 2u) != 0 && model == kSimpleRadial;
  if (threadIdx.x == 0) {
    for (int k = 0; k < 4; ++k) { sx.q[k] = cam_q[4 * f + k]; sx.in4[k] = intr[4 * f + k]; }
    for (int k = 0; k < 3; ++k) sx.t[k] = cam_t[3 * f + k];
    sh_flag[0] = 0;
  }
  __syncthreads();
  pose_pass<TrackT, true>(pts, tr, mk, P, model, loss, loss_scale, sx, red, lin);

  // ---- thread-0-private LM state (Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy)
  double radius = opt.initial_trust_region_radius, decrease = 2.0, x_cost = 0.0, initial_cost = 0.0;
  double scale[kPN], colsq[kPN];
  int it = 0, invalid_streak = 0, n_succ = 0, n_unsucc = 0, term = 0;
  auto grad_max = [&](const PoseState& x, const double* g) {
    double d[kPN], m = 0.0;
    for (int k = 0; k < kPN; ++k) d[k] = act[k] ? -g[k] : 0.0;
    PoseState o;
    plus_state(x, d, o);
    for (int k = 0; k < 4; ++k) m = fmax(m, fabs(o.q[k] - x.q[k]));
    for (int k = 0; k < 3; ++k) m = fmax(m, fabs(o.t[k] - x.t[k]));
    m = fmax(m, fmax(fabs(o.in4[0] - x.in4[0]), fabs(o.in4[3] - x.in4[3])));
    return m;
  };
  double gmax = 0.0;
  if (threadIdx.x == 0) {
    x_cost = 0.5 * lin[kPH + kPN];
    initial_cost = x_cost;
    for (int k = 0; k < kPN; ++k) {
      colsq[k] = act[k] ? sym_at(lin, k, k) : 0.0;
      scale[k] = opt.jacobi_scaling ? 1.0 / (1.0 + sqrt(colsq[k])) : 1.0;
    }
    gmax = grad_max(sx, lin + kPH);
  }
  bool step_ok = true;   // last step successful (iteration 0 counts as successful)
  for (;;) {
    __syncthreads();                             // everybody has consumed sh_flag / cand of the previous round
    // ---- thread 0: termination tests, damped normal equations, candidate
    if (threadIdx.x == 0) {
      int done = 0;
      if (it >= opt.max_num_iterations) { done = 1; term = 0; }
      else if (step_ok && gmax <= opt.gradient_tolerance) { done = 1; term = 1; }
      else if (radius <= opt.min_trust_region_radius) { done = 1; term = 4; }
      if (!done) {
        ++it;
        // (S J^T J S + D^2) y = S g
        double A[kPN][kPN], b[kPN];
        for (int i = 0; i < kPN; ++i) {
          for (int j = 0; j < kPN; ++j) A[i][j] = (act[i] && act[j]) ? scale[i] * scale[j] * sym_at(lin, i, j) : 0.0;
          double dd = colsq[i] * scale[i] * scale[i];
          dd = fmin(fmax(dd, opt.min_lm_diagonal), opt.max_lm_diagonal);
          A[i][i] = act[i] ? A[i][i] + dd / radius : 1.0;
          b[i] = act[i] ? scale[i] * lin[kPH + i] : 0.0;
        }
        bool bad = false;
        for (int j = 0; j < kPN; ++j) {          // in-place Cholesky + solve
          double d = A[j][j];
          for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k];
          if (!(d > 0.0)) { bad = true; break; }
          d = sqrt(d); A[j][j] = d;
          for (int i = j + 1; i < kPN; ++i) { double s2 = A[i][j]; for (int k = 0; k < j; ++k) s2 -= A[i][k] * A[j][k]; A[i][j] = s2 / d; }
        }
        double y[kPN];
        if (!bad) {
          for (int i = 0; i < kPN; ++i) { double s2 = b[i]; for (int k = 0; k < i; ++k) s2 -= A[i][k] * y[k]; y[i] = s2 / A[i][i]; }
          for (int i = kPN - 1; i >= 0; --i) { double s2 = y[i]; for (int k = i + 1; k < kPN; ++k) s2 -= A[k][i] * y[k]; y[i] = s2 / A[i][i]; }
        }
        // model cost change = -(J d)^T (r + J d / 2) = -d^T g - d^T H d / 2 with d = -S y
        double mcc = 0.0, dlt[kPN];
        if (!bad) {
          for (int i = 0; i < kPN; ++i) { dlt[i] = act[i] ? -y[i] * scale[i] : 0.0; if (!(fabs(dlt[i]) <= 1.7976931348623157e308)) bad = true; }
          double dg = 0.0, dhd = 0.0;
          for (int i = 0; i < kPN; ++i) { dg += dlt[i] * lin[kPH + i]; for (int j = 0; j < kPN; ++j) dhd += dlt[i] * sym_at(lin, i, j) * dlt[j]; }
          mcc = -dg - 0.5 * dhd;
        }
        if (bad || !(mcc > 0.0)) {
          if (++invalid_streak >= opt.max_num_consecutive_invalid_steps) { done = 1; term = 5; }
          radius /= decrease; decrease *= 2.0; step_ok = false; ++n_unsucc;
          sh_flag[1] = -1;                       // no candidate this round
        } else {
          invalid_streak = 0;
          for (int i = 0; i < kPN; ++i) sh_delta[i] = dlt[i];
          plus_state(sx, dlt, sc);
          sh_flag[1] = 1;
          sh_flag[2] = __double2hiint(mcc); sh_flag[3] = __double2loint(mcc);
        }
      }
      sh_flag[0] = done;
    }
    __syncthreads();
    if (sh_flag[0]) break;
    if (sh_flag[1] < 0) continue;                // invalid step: next iteration with a smaller radius
    // ---- all threads: cost + linearisation at the candidate (one pass; discarded if the step is rejected)
    pose_pass<TrackT, true>(pts, tr, mk, P, model, loss, loss_scale, sc, red, cand);
    if (threadIdx.x == 0) {
      const double mcc = __hiloint2double(sh_flag[2], sh_flag[3]);
      const double cand_cost = 0.5 * cand[kPH + kPN];
      double sn = 0.0, xn = 0.0;
      for (int k = 0; k < 4; ++k) { const double d = sc.q[k] - sx.q[k]; sn += d * d; xn += sx.q[k] * sx.q[k]; }
      for (int k = 0; k < 3; ++k) { const double d = sc.t[k] - sx.t[k]; sn += d * d; xn += sx.t[k] * sx.t[k]; }
      if (act[6] || act[7]) {                    // the camera block is only part of the problem when refined
        const int np = (model == kSimpleRadial) ? 4 : 3;
        for (int k = 0; k < np; ++k) { const double d = sc.in4[k] - sx.in4[k]; sn += d * d; xn += sx.in4[k] * sx.in4[k]; }
      }
      const double step_norm = sqrt(sn), x_norm = sqrt(xn);
      int done = 0;
      if (!(step_norm > opt.parameter_tolerance * (x_norm + opt.parameter_tolerance))) { done = 1; term = 3; }
      else if (fabs(x_cost - cand_cost) <= opt.function_tolerance * x_cost) { done = 1; term = 2; }
      else {
        const double rel = (x_cost - cand_cost) / mcc;
        if (rel > opt.min_relative_decrease) {
          sx = sc;
          x_cost = cand_cost;
          for (int i = 0; i < kPV; ++i) lin[i] = cand[i];
          for (int k = 0; k < kPN; ++k) colsq[k] = act[k] ? sym_at(lin, k, k) : 0.0;
          gmax = grad_max(sx, lin + kPH);
          const double tmp = 2.0 * rel - 1.0;
          radius = fmin(opt.max_trust_region_radius, radius / fmax(1.0 / 3.0, 1.0 - tmp * tmp * tmp));
          decrease = 2.0; step_ok = true; ++n_succ;
        } else {
          radius /= decrease; decrease *= 2.0; step_ok = false; ++n_unsucc;
        }
      }
      sh_flag[0] = done;
    }
    __syncthreads();
    if (sh_flag[0]) break;
  }
  if (threadIdx.x == 0) {
    for (int k = 0; k < 4; ++k) { cam_q[4 * f + k] = sx.q[k]; intr[4 * f + k] = sx.in4[k]; }
    for (int k = 0; k < 3; ++k) cam_t[3 * f + k] = sx.t[k];
    if (summaries) {
      vgg_ba_summary sm;
      sm.initial_cost = initial_cost; sm.final_cost = x_cost; sm.num_iterations = it; sm.num_successful_steps = n_succ;
      sm.num_unsuccessful_steps = n_unsucc; sm.termination = term; sm.n_reduced = 6 + (act[6] ? 1 : 0) + (act[7] ? 1 : 0);
      sm.num_log = 0;
      summaries[blockIdx.x] = sm;
    }
  }
}

}  // namespace vgg

using namespace vgg;

extern "C" int vgg_pose_refine(const double* points3D, const void* tracks, int tracks_are_f64, const uint8_t* inlier_mask,
                               int S, int P, const int32_t* frame_ids, int num_frames, double* cam_q, double* cam_t,
                               double* intr, int camera_model, const uint8_t* refine_flags, const vgg_ba_options* options,
                               int loss, double loss_scale, vgg_ba_summary* summaries, void* stream) {
  if (!points3D || !tracks || !inlier_mask || !frame_ids || !cam_q || !cam_t || !intr || !options || S <= 0 || P < 0)
    return VGG_ERR_INVALID_ARGUMENT;
  if (camera_model != kPinhole && camera_model != kSimpleRadial) return VGG_ERR_UNSUPPORTED;
  if (num_frames <= 0) return VGG_OK;
  hipStream_t st = (hipStream_t)stream;
  if (tracks_are_f64)
    pose_refine_kernel<double><<<num_frames, 256, 0, st>>>(points3D, (const double*)tracks, inlier_mask, S, P, frame_ids, cam_q,
                                                           cam_t, intr, camera_model, refine_flags, *options, loss, loss_scale,
                                                           summaries);
  else
    pose_refine_kernel<float><<<num_frames, 256, 0, st>>>(points3D, (const float*)tracks, inlier_mask, S, P, frame_ids, cam_q,
                                                          cam_t, intr, camera_model, refine_flags, *options, loss, loss_scale,
                                                          summaries);
  VGG_LAUNCH_CHECK();
  return VGG_OK;
}
