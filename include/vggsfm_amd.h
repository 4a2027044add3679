/*
 * vggsfm_amd -- C-ABI of the MI355X (gfx950) geometry hot path of VGGSfM.
 *
 * This is the drop-in boundary: plain pointers + sizes, no torch / pycolmap types.  Every pointer
 * named *device* is HBM memory of the current HIP device; `stream` is a hipStream_t passed as void*.
 * All entry points are asynchronous on `stream` unless stated, never allocate, and return VGG_OK (0)
 * or a negative VGG_ERR_* code.  Each one names the reference interface it replaces (paths under
 * /root/reference); the Python host side in vggsfm_amd/ mirrors the reference signatures on top.
 *
 * Layout conventions (the reference's own): extrinsics (S,3,4) row-major
 * [R|t] with x_cam = R X + t (OpenCV frame); intrinsics (S,3,3) row-major; tracks (S,P,2) pixels
 * (float32 as produced by the tracker, or float64); masks uint8 0/1; all geometry float64.
 */
#ifndef VGGSFM_AMD_H
#define VGGSFM_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VGG_OK 0
#define VGG_ERR_INVALID_ARGUMENT (-1)
#define VGG_ERR_HIP (-2)
#define VGG_ERR_WORKSPACE (-3)
#define VGG_ERR_UNSUPPORTED (-4)

/* library / build identification ("gfx950"), for loud failure when the wrong object is loaded */
const char* vgg_build_arch(void);
int vgg_abi_version(void);          /* VGG_ABI_VERSION of the build: bumped whenever a struct below changes layout */
#define VGG_ABI_VERSION 2
/* sizeof of the by-pointer structs as the library was compiled (which: 0 vgg_ba_problem, 1 vgg_ba_options,
 * 2 vgg_ba_iteration, 3 vgg_ba_summary; 0 for anything else): a binding checks its own layout against these and
 * refuses to drive a library built from another revision of this header (vggsfm_amd/_lib.py does). */
size_t vgg_abi_sizeof(int which);

/* ------------------------------------------------------------------------------------------
 * project_3D_points + img_from_cam       vggsfm/utils/triangulation_helpers.py:311-395
 * out_uv (S,P,2) and/or out_cam (S,3,P) may be NULL; intrinsics NULL == only_points_cam.
 * num_extra in {0,1,2,4} (SIMPLE_RADIAL / RADIAL / OPENCV distortion, distortion.py:102-159). */
int vgg_project_points(const double* points3D, int P, const double* extrinsics, const double* intrinsics,
                       const double* extra_params, int num_extra, int S, double* out_uv, double* out_cam,
                       void* stream);

/* filter_all_points3D(_single_chunk)      vggsfm/utils/triangulation_helpers.py:133-307
 * reprojection error^2 <= max^2 with depth > 0 (else `behind_value`: 1e6 there, 1e9 in refine_pose,
 * triangulation.py:308-312), >= 2 inlier views, |X| <= hard_max (<=0 disables), optional "some inlier
 * pair subtends >= min_tri_angle degrees".  out_mask (P); out_detail (S,P) or NULL.
 * workspace: vgg_filter_points_workspace_bytes(S) device bytes (only read when check_triangle). */
size_t vgg_filter_points_workspace_bytes(int S);
int vgg_filter_points(const double* points3D, int P, const void* tracks, int tracks_are_f64, const double* extrinsics,
                      const double* intrinsics, const double* extra_params, int num_extra, int S,
                      double max_reproj_error, double min_tri_angle, int check_triangle, double hard_max,
                      double behind_value, uint8_t* out_mask, uint8_t* out_detail, void* workspace, void* stream);

/* cam_from_img + iterative_undistortion   vggsfm/utils/triangulation_helpers.py:398-428,
 *                                         vggsfm/utils/distortion.py:27-99
 * out (S,P,2) f64.  With num_extra > 0 runs the reference's Newton iteration incl. its tensor-global
 * stopping rule; this SYNCHRONISES the stream every 8 iterations; *iterations_run (host) may be NULL. */
size_t vgg_cam_from_img_workspace_bytes(int S, int P, int max_iterations);
int vgg_cam_from_img(const void* tracks, int tracks_are_f64, const double* intrinsics, const double* extra_params,
                     int num_extra, int S, int P, double* out, int max_iterations, double max_step_norm,
                     double rel_step_size, double eps, void* workspace, int* iterations_run, void* stream);

/* ------------------------------------------------------------------------------------------
 * triangulate_tracks_single_chunk         vggsfm/utils/triangulation.py:776-956
 * LO-RANSAC multi-view DLT for the N tracks of one chunk, one wavefront per track.
 *   tracks_t (N,S,2) f64: normalised rays, TRACK-major (the reference transposes too, :798)
 *   invalid_vis_conf_t (N,S) uint8 = (vis <= 0.05) | (score <= 0.5)              (:867-874)
 *   pairs (H,2) int32 device: hypothesis view pairs; the caller reproduces the reference's host-side
 *   torch.randperm draw (:804-813).  H <= 256, lo_num <= 64 (second round: min(10, lo_num), :902-904).
 * Hypotheses are ranked by inlier count in STABLE descending order (tie-break contract).
 * *threshold_io (host, in/out): calculate_residual_indicator (vggsfm/two_view_geo/utils.py:63-87)
 * normalises by the chunk-global max mean inlier error + 1e-6.  Pass 2*pi + 1e-6 (exact whenever some
 * hypothesis of the chunk has no inlier); on return it holds the measured value -- if it differs,
 * call again with the returned value.  Synchronises the stream once.
 * Outputs: points (N,3) f64, inlier_num (N) int64, inlier_mask (N,S) uint8. */
/* triangulate_tracks over ALL reference chunks of a call in one launch (vggsfm/utils/triangulation.py:712-758 splits
 * the track axis into ceil(S*N/819200) chunks, each with its own torch.randperm draw and its own chunk-global
 * residual-indicator threshold).  pairs [num_chunks,H,2] int32: the hypothesis pairs of every chunk, drawn by the
 * caller in chunk order; chunk c = tracks [c*chunk_size, min(N,(c+1)*chunk_size)).  thresholds_io: host array
 * [num_chunks], semantics of threshold_io above, per chunk.  One stream synchronisation.
 * Limits (both entries): 1 <= H <= 256 hypotheses per track, 1 <= lo_num <= 64 local-optimisation candidates;
 * VGG_ERR_INVALID_ARGUMENT beyond (the reference calls with H = 256 or 128 and lo_num = 50). */
size_t vgg_triangulate_chunks_workspace_bytes(int S, int num_chunks);
int vgg_triangulate_tracks_chunks(const double* extrinsics, const double* tracks_t, const uint8_t* invalid_vis_conf_t,
                                  const int32_t* pairs, int S, int N, int H, int num_chunks, int chunk_size, int lo_num,
                                  double max_angular_error_deg, double min_tri_angle_deg, double* out_points,
                                  int64_t* out_inlier_num, uint8_t* out_inlier_mask, double* thresholds_io,
                                  void* workspace, void* stream);
/* Asynchronous form of the same launch (round 4): nothing is copied back, nothing is waited for.  `num_chunks` consecutive
 * chunks starting at tracks_t[0]; thresholds_dev [num_chunks] and gmax_dev [num_chunks] (zeroed by the caller) are DEVICE
 * arrays -- gmax_dev receives the bit pattern of the largest mean inlier error of every chunk (compare with the threshold
 * semantics above); centers_dev [S][3] is scratch.  Lets the caller enqueue a call group of chunks by group of chunks while
 * it is still drawing the later groups' hypothesis pairs on the host. */
int vgg_triangulate_tracks_chunks_enqueue(const double* extrinsics, const double* tracks_t, const uint8_t* invalid_vis_conf_t,
                                          const int32_t* pairs, int S, int N, int H, int num_chunks, int chunk_size, int lo_num,
                                          double max_angular_error_deg, double min_tri_angle_deg, double* out_points,
                                          int64_t* out_inlier_num, uint8_t* out_inlier_mask, const double* thresholds_dev,
                                          unsigned long long* gmax_dev, double* centers_dev, void* stream);

/* triangulate_by_pair (vggsfm/utils/triangulation.py:45-135): the S-1 two-view DLT point clouds between frame 0 and
 * every other frame.  extrinsics [S,3,4] f64, tracks_normalized [S,N,2] f64 (frame-major, camera rays) ->
 * out_points [S-1,N,3] f64.  Cheirality and triangulation angle are elementwise follow-ups on the host side. */
int vgg_triangulate_by_pair(const double* extrinsics, const double* tracks_normalized, int S, int N, double* out_points,
                            void* stream);

size_t vgg_triangulate_workspace_bytes(int S, int N, int H, int lo_num);
int vgg_triangulate_tracks(const double* extrinsics, const double* tracks_t, const uint8_t* invalid_vis_conf_t,
                           const int32_t* pairs, int S, int N, int H, int lo_num, double max_angular_error_deg,
                           double min_tri_angle_deg, double* out_points, int64_t* out_inlier_num,
                           uint8_t* out_inlier_mask, double* threshold_io, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * Bundle adjustment / pose refinement = what the reference obtains from
 *   pycolmap.bundle_adjustment   vggsfm/utils/triangulation.py:213,1050,1142; runners/video_runner.py:508
 *   pycolmap.pose_refinement     vggsfm/utils/triangulation.py:387,590; runners/video_runner.py:1001
 * (COLMAP BundleAdjuster on Ceres' Levenberg-Marquardt, Schur elimination of the points).
 *
 * State and structure live on the device in flat arrays; nothing is marshalled through host objects.
 * Observations are given twice: point-major CSR (row_ptr/obs_*) and camera-major CSR (col_ptr/cobs_*),
 * both sorted ascending in the minor index.  `tile_*` is the block-sparse Schur work list built by the
 * host side (vggsfm_amd/ba.py: build_schur_tiles): cameras are grouped 16 at a time, a "segment" is the
 * run of one point's observations inside one camera group, an "entry" pairs two segments of one point. */
typedef struct {
  int32_t num_cams, num_pts, num_obs, num_intr;   /* num_intr == 1 (shared camera) or == num_cams */
  int32_t camera_model;                           /* 0 SIMPLE_PINHOLE  1 SIMPLE_RADIAL */
  int32_t refine_focal, refine_extra;             /* principal point is always constant */
  int32_t loss;                                   /* 0 trivial 1 Cauchy 2 Huber 3 SoftL1 */
  double loss_scale;
  /* state, in/out (device) */
  double* cam_q;              /* [num_cams,4] unit quaternion x,y,z,w */
  double* cam_t;              /* [num_cams,3] */
  double* intr;               /* [num_intr,4] f,cx,cy,k */
  double* pts;                /* [num_pts,3] */
  /* structure (device) */
  const int32_t* row_ptr;     /* [num_pts+1] */
  const int32_t* obs_cam;     /* [num_obs] */
  const float* obs_uv;        /* [num_obs,2] */
  const int32_t* col_ptr;     /* [num_cams+1] */
  const int32_t* cobs_pt;     /* [num_obs] */
  const float* cobs_uv;       /* [num_obs,2] */
  const uint8_t* cam_const;   /* [num_cams] bit0: pose constant; bit1..3: t_x,t_y,t_z constant; or NULL */
  const uint8_t* intr_const;  /* [num_intr] or NULL */
  const uint8_t* pt_const;    /* [num_pts] or NULL */
  /* Schur tile work list (device) */
  int32_t num_chunks;
  int32_t num_tile_batches;   /* >= 1: the tiles are cut into batches of consecutive camera groups groupI (tile_batches) */
  const int32_t* chunk_desc;  /* [num_chunks,6] = groupI, groupJ, tile_entry_begin, tile_entry_end, j, J:
                                 workgroup j of the J of its tile takes the 32-entry sub-chunks j, j+J, ... */
  const int32_t* entries;     /* [num_entries,4] = point index (row of pts), segment_A, segment_B, maskA | maskB<<16 of the
                                 QUAD: the four consecutive entries, aligned to tile_entry_begin, that the kernel packs
                                 along K (bit l of a mask: camera group*16+l observes one of the four points; 16-row
                                 blocks of the tile without any camera are skipped for the quad) */
  int32_t num_segments;       /* segments: runs of one point's observations inside one camera group */
  const int32_t* obs_slot;    /* [num_obs] = segment*16 + (camera % 16): where the point-major observation's
                                 Schur factor lives in the zero-padded segment buffer */
  int32_t num_tiles;
  const int32_t* tile_desc;   /* [num_tiles,4] = groupI, groupJ, chunk_begin, chunk_end (chunks of one tile
                                 are consecutive; their partial sums are reduced in this order) */
  const int32_t* tile_batches;/* HOST array [num_tile_batches,6] = chunk_begin, first_diagonal_chunk, chunk_end, tile_begin,
                                 tile_end, first_camera_group.  Batch-major order of chunk_desc / tile_desc: inside a batch
                                 the chunks of the off-diagonal tiles (groupI < groupJ) come first, then the diagonal
                                 ones (separate launches).  Column block g of the reduced system only receives sums
                                 from tiles with groupI <= g: it is complete once the batches up to that of group g
                                 are done, so the factorisation can start while later batches are being computed. */
  int32_t chol_split_a;       /* Block-diagonal leading part of the reduced camera system, 0 = none.  If the cameras are */
  int32_t chol_split_b;       /* ordered so that no point is seen both by one of the first chol_split_a / 6 cameras and by
                                 one of the next chol_split_b / 6 (sliding-window / video-like visibility: the host side
                                 checks this and orders the cameras, ba.py), the pose columns [0, a) and [a, a + b) of S
                                 do not couple and their factorisations advance side by side in shared launches
                                 (chol.hip); a must be a multiple of 64.  The caller guarantees the structure -- with
                                 several GPUs for the SUM over ranks. */
  const int32_t* chol_first_blk; /* device, optional: ROW ENVELOPE of the reduced system in 64-column blocks.  Entry r (r = 0 ..
                                 ceil(n / 64) - 1) = the first block column in which block row r (rows 64 r .. 64 r + 63 of S) can
                                 hold a non-zero; the Cholesky factor keeps the row envelope of the matrix, so the dataflow
                                 factorisation (chol.hip) skips the tiles left of it and drops their dependences: camera
                                 blocks that do not couple factor concurrently, whatever their number (ba.py:
                                 find_camera_order -- nested-dissection order of sliding-window / video visibility).  NULL =
                                 dense (then chol_split_a / b, if set, describe a two-block leading part).  The caller
                                 guarantees the structure, with several ranks for the SUM of their systems. */
  int32_t merged_tile_launch;    /* 1: the off-diagonal and the diagonal chunks of a batch run in ONE launch (the host sized them for
                                 one shared resident round) -- small problems, where the second launch costs more than the
                                 lower occupancy of the merged kernel */
} vgg_ba_problem;

typedef struct {
  int32_t max_num_iterations;
  int32_t max_num_consecutive_invalid_steps;
  int32_t jacobi_scaling;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius;
  double min_lm_diagonal, max_lm_diagonal, min_relative_decrease;
  int32_t overlap_factorization; /* 0 = off; otherwise the number of CUs (a multiple of 32) given to the factorisation
                                    while it overlaps the tile batches 1.. (>= 2 batches; they then run on the other
                                    CUs, both on CU-masked side streams joined back into `stream`).  Single GPU only:
                                    the multi-GPU loop all-reduces the complete system between phases 1 and 2 */
} vgg_ba_options;

typedef struct {
  int32_t iteration;
  int32_t successful;
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, radius;
} vgg_ba_iteration;

typedef struct {
  double initial_cost, final_cost;
  int32_t num_iterations, num_successful_steps, num_unsuccessful_steps;
  int32_t termination;        /* 0 iteration cap, 1 gradient, 2 function, 3 parameter, 4 radius, 5 failure */
  int32_t n_reduced, num_log;
} vgg_ba_summary;

/* device bytes vgg_ba_solve needs in `workspace` (256-byte aligned base) */
size_t vgg_ba_workspace_bytes(const vgg_ba_problem* problem, const vgg_ba_options* options);

/* Runs the whole LM loop on `stream` (device-side control flow: the host enqueues
 * max_num_iterations iterations, a device flag turns the rest into no-ops after termination),
 * then synchronises ONCE to copy summary + per-iteration log (log_cap entries, may be NULL). */
int vgg_ba_solve(const vgg_ba_problem* problem, const vgg_ba_options* options, void* workspace,
                 size_t workspace_bytes, vgg_ba_summary* summary, vgg_ba_iteration* log, int log_cap, void* stream);

/* Multi-GPU (points sharded across ranks, cameras replicated): the same loop split into phases so
 * that the host can interleave RCCL all-reduces on the same stream.
 *   vgg_ba_begin    : init control block
 *   phase 0 LINEARIZE: camera-side J^T J / J^T r / cost of the local points  -> reduce buffer 0 (SUM)
 *   phase 1 SCHUR    : point blocks, reduced camera system of the local points -> reduce buffer 1 (SUM),
 *                      gradient max of local points                            -> reduce buffer 2 (MAX)
 *   phase 2 STEP     : Cholesky + back-substitution + candidate cost         -> reduce buffer 3 (SUM)
 *   phase 3 UPDATE   : trust-region decision, commit
 *   phase 4 / 5      : pack / unpack the lower triangle of the reduced system + rhs into / from reduce buffer 4
 *                      (n(n+1)/2 + n doubles): all-reducing buffer 4 between them replaces the all-reduce of
 *                      buffer 1 (n^2 + n doubles, upper triangle all zero) at half the payload
 *   phase 6          : unpack from the OUTPUT of a reduce-scatter + all-gather of buffer 4, read in place: with W ranks
 *                      and c = ceil(count / W), phase 4 also zeroes buffer 4 up to W c doubles (it is carved with that
 *                      padding: the reduce-scatter input, no staging copy) and puts the rank's gradient maximum at
 *                      element c of buffer 5 (c + 1 doubles: reduce-scatter output = all-gather input); buffer 6
 *                      (W (c + 1) doubles) is the all-gather output; phase 6 rebuilds S | rhs from its W slices and
 *                      takes the maximum of the W riding gradient norms (replaces the MAX reduce of buffer 2).  W <= 1024
 *   phases 7..12     : the SPLIT exchange (round 6) -- phase 1 and the exchange of the system cut in two so that the first
 *                      half of the payload travels while the diagonal tile launch computes the second:
 *                      7 = phase 1 up to the sums of the OFF-DIAGONAL tiles; 8 = pack part A (the elements of the lower
 *                      triangle whose row and column belong to cameras of different 16-camera groups -- all the off-diagonal
 *                      launch fills) into buffer 4; 9 = the rest of phase 1 (diagonal launch, its sums, assemble); 10 = pack
 *                      part B (everything else + rhs) behind it; 11 = S | rhs from the two parts' all-gather outputs; 12 =
 *                      query, to be called ONCE per workspace before phases 7..11: VGG_OK if the split is available for this problem (one tile
 *                      batch, separate tile launches, more than 16 cameras) -- it then also writes the rows' offsets inside the two parts into the
 *                      workspace --, VGG_ERR_UNSUPPORTED otherwise (nothing is launched).  With a = vgg_ba_reduce_buffer(7) elements in
 *                      part A, b = count(4) - a, ca = ceil(a / W), cb = ceil(b / W): buffer 4 = [A: W ca | B: W cb] (reduce-
 *                      scatter inputs), buffer 5 = [A: ca | B: cb + 1] (the rank's slices; the gradient maximum rides behind
 *                      B's), buffer 6 = [A: W ca | B: W (cb + 1)] (all-gather outputs)
 * vgg_ba_reduce_buffer returns the device address / element count (doubles) of each reduce buffer (which = 7: buffer 4's address
 * and the element count of part A). */
int vgg_ba_begin(const vgg_ba_problem* problem, const vgg_ba_options* options, void* workspace, size_t workspace_bytes,
                 int rank, int world_size, void* stream);
int vgg_ba_phase(const vgg_ba_problem* problem, const vgg_ba_options* options, void* workspace, int phase,
                 void* stream);
int vgg_ba_reduce_buffer(const vgg_ba_problem* problem, const vgg_ba_options* options, void* workspace, int which,
                         double** device_ptr, size_t* count);
int vgg_ba_finish(const vgg_ba_problem* problem, const vgg_ba_options* options, void* workspace,
                  vgg_ba_summary* summary, vgg_ba_iteration* log, int log_cap, void* stream);
/* Has the solve terminated?  Copies the 4-byte device flag and synchronises the stream (what vgg_ba_solve does every
 * 8 iterations): a phase-driven caller uses it to stop enqueuing iterations -- and, with several ranks, their
 * collectives -- once the (rank-identical) device-side decision has ended the solve. */
int vgg_ba_poll_done(const vgg_ba_problem* problem, const vgg_ba_options* options, void* workspace, int32_t* done_host,
                     void* stream);

/* pycolmap.pose_refinement for a batch of frames   vggsfm/utils/triangulation.py:387,590 (loops :341-441, :542-608)
 * = COLMAP RefineAbsolutePose: points constant, robust loss (reference: Cauchy, scale 1), unknowns pose
 * (+ focal if refine_flags bit0, + extra param if bit1).  One workgroup solves one frame; frames are
 * independent.  frame_ids (F) int32 device: frames to refine; inlier_mask (S,P) uint8: observations used.
 * cam_q (S,4) / cam_t (S,3) / intr (S,4 = f,cx,cy,k PER FRAME) are updated in place for those frames.
 * options: Ceres defaults for this call site are max_num_iterations 100, function_tolerance 1e-6,
 * gradient_tolerance 1.0, parameter_tolerance 1e-8.  summaries (F) device or NULL. */
int vgg_pose_refine(const double* points3D, const void* tracks, int tracks_are_f64, const uint8_t* inlier_mask, int S,
                    int P, const int32_t* frame_ids, int num_frames, double* cam_q, double* cam_t, double* intr,
                    int camera_model, const uint8_t* refine_flags, const vgg_ba_options* options, int loss,
                    double loss_scale, vgg_ba_summary* summaries, void* stream);

/* pycolmap.absolute_pose_estimation, the RANSAC part   vggsfm/utils/triangulation.py:324-326,400-432 (fallback of
 * refine_pose), vggsfm/runners/video_runner.py:987-998 (align_next_window, use_pnp)
 * = COLMAP EstimateAbsolutePose restated (third-party, absent from the reference tree; PARITY UNPINNED, see
 * oracle/p3p.py): P3P minimal samples, squared reprojection error on the normalised image plane, points with
 * depth <= 0 never inliers, support = (most inliers, then smallest inlier residual sum, then lowest index).
 * Deviations: the caller draws a fixed number of samples (no adaptive trial count), no EPnP step inside the loop;
 * the caller runs vgg_pose_refine on the returned inliers afterwards, as COLMAP does.
 * All num_frames (virtual) frames run concurrently.  A virtual frame is (frame, focal length factor): COLMAP's
 * estimate_focal_length runs one RANSAC per factor -- the caller passes the points normalised with each scaled
 * camera, frames_per_sample_set consecutive virtual frames share one sample set.
 *   points2D_normalized [num_frames][num_points][2] f64, points3D [num_points][3] f64,
 *   candidate_mask [num_frames][num_points] uint8 or NULL (matches that take part), samples
 *   [num_frames / frames_per_sample_set][num_hypotheses][3] int32 indices into num_points (distinct, candidates),
 *   max_error_sq [num_frames] f64 (squared threshold on the normalised plane = (max_error / focal)^2).
 * Outputs per virtual frame: out_pose [12] row-major [R|t] (zeros if nothing found), out_num_inliers (0 if nothing
 * found), out_residual_sum, out_best (hypothesis * 4 + solution, -1 if none), out_inlier_mask [num_points]. */
size_t vgg_p3p_ransac_workspace_bytes(int num_frames, int num_hypotheses);
int vgg_p3p_ransac(const double* points2D_normalized, const double* points3D, const uint8_t* candidate_mask,
                   const int32_t* samples, int num_frames, int frames_per_sample_set, int num_points, int num_hypotheses,
                   const double* max_error_sq, double* out_pose, int32_t* out_num_inliers, double* out_residual_sum,
                   int32_t* out_best, uint8_t* out_inlier_mask, void* workspace, void* stream);

/* Two-view stage in front of the path: estimate_fundamental = 7-point RANSAC + 8-point local
 * optimisation for all (query frame, other frame) pairs at once   vggsfm/two_view_geo/fundamental.py:43-183,
 * two_view_geo/utils.py:63-298; caller estimate_preliminary.py:103-152 (-> fmat_inlier_mask).  PARITY PARTLY PINNED
 * (oracle/fundamental.py: Sampson distance, 8-point fit and winner selection checked against the reference's own functions;
 * the 7-point solver and the RNG are not -- kornia is absent); float64 here.
 * points1 / points2 [num_pairs][num_points][2] f64 pixels; valid_mask [num_pairs][num_points] uint8 or NULL
 * (matches with vis >= 0.05 and score >= 0.5, estimate_preliminary.py:118-126); matrices are 9 doubles row-major,
 * unit Frobenius norm, x2^T F x1 = 0.
 *  vgg_fmat_seven_point  samples [num_samples][7] int32 shared by all pairs (fundamental.py:67-75) ->
 *                        out_fmat [num_pairs][num_samples][3][9], out_valid [..][3] (1..3 real roots of the cubic)
 *  vgg_fmat_score        squared Sampson distance (utils.py:90-172) of every match under every hypothesis; inliers
 *                        <= max_error_sq and valid; out_counts [num_pairs][num_hypotheses] (-1 for an invalid
 *                        hypothesis), out_residual_sums = sum of the inlier residuals (-> the mean that breaks ties,
 *                        utils.py:63-87)
 *  vgg_fmat_eight_point  for each selected [num_pairs][num_selected] index into src_fmat [num_pairs][num_src][9]: the
 *                        8-point fit (masked normalisation, smallest eigenvector of X^T X, rank 2; fundamental.py:261-334)
 *                        on the inliers of that hypothesis, recomputed on the fly; selected entries whose src_counts is
 *                        negative give an invalid result
 *  vgg_fmat_residuals    fmat [num_pairs][9] -> out_residuals [num_pairs][num_points] (1e6 for invalid matches)
 * The caller sorts the counts (stable, descending) to select the lo_num best hypotheses between the calls. */
int vgg_fmat_seven_point(const double* points1, const double* points2, const int32_t* samples, int num_pairs, int num_points,
                         int num_samples, double* out_fmat, uint8_t* out_valid, void* stream);
int vgg_fmat_score(const double* points1, const double* points2, const uint8_t* valid_mask, const double* fmat,
                   const uint8_t* fmat_valid, int num_pairs, int num_points, int num_hypotheses, double max_error_sq,
                   int32_t* out_counts, double* out_residual_sums, void* stream);
int vgg_fmat_eight_point(const double* points1, const double* points2, const uint8_t* valid_mask, const double* src_fmat,
                         const int32_t* src_counts, const int32_t* selected, int num_pairs, int num_points, int num_src,
                         int num_selected, double max_error_sq, double* out_fmat, uint8_t* out_valid, void* stream);
int vgg_fmat_residuals(const double* points1, const double* points2, const uint8_t* valid_mask, const double* fmat, int num_pairs,
                       int num_points, double* out_residuals, void* stream);

/* Optional per-kernel timing (HIP events recorded on the launch stream around the dominant kernels).
 * kernel_id: 0 cam_pass<linearize> 1 point_pass 2 cam_pass<rhs> 3 schur_tile<off-diagonal tiles> 4 cholesky (all
 * launches of one solve) 5 point_step 6 schur_tile<diagonal tiles>.  vgg_ba_profile_read synchronises on the
 * recorded events. */
/* Launch choices the solver otherwise makes from the problem's shape (process-wide; for tests and measurements):
 * lanes_per_point 8 | 16 | 32 | 64 (0 = from the mean track length) -- lanes of a wavefront that share one point in the point
 * passes; long_tracks 1 / 0 (-1 = automatic) -- the variants that prefetch four observations per lane and keep no cached
 * Jacobians; cam_workgroups / point_workgroups (0 = automatic) -- total workgroups of the camera / point passes.  Every
 * combination computes the same iteration up to the order of its sums. */
int vgg_ba_tuning(int lanes_per_point, int long_tracks, int cam_workgroups, int point_workgroups);
/* Where F^T (r - E h) (and, with a shared camera, the intrinsics border of the reduced system) come from: 2 (default;
 * VGG_TILE_RHS in the environment seeds it) = from the diagonal Schur tile launch for every block shape -- 6 x 6 blocks
 * (shared or constant intrinsics, compressed factors): two wavefronts of a diagonal-tile workgroup multiply the staged segments
 * by a 3 x 3 per-point block on the side, + per-point sums of the point pass; 7 x 7 / 8 x 8 blocks (per-camera intrinsics, full
 * factors; round 6): every thread adds one tile row's products for two of a batch's entries -- no second camera-major evaluation
 * of the projections per iteration; 1 = the tile launch for 6 x 6 blocks only, the camera pass cam_pass<RHS> for the others (the
 * round-4/5 behaviour); 0 = the camera pass always.  All compute the same system up to the order of its sums.  Requires
 * entries[e][0] = the entry's point. */
int vgg_ba_set_tile_rhs(int enable);
/* Where the back-substitution of the points (point_step_kernel) takes E^T F dy from, again with 6 x 6 tile blocks: 0 (default;
 * VGG_STEP_FACTORS seeds it) = a second evaluation of every projection and its Jacobians; 1 = from the compressed Schur
 * factors the point pass left in the segment buffer (one 96-byte record per observation; the model cost change's camera
 * share from the cameras' J^T J, J^T r); 2 = every other observation of a lane from its factor.  Same step up to rounding;
 * 1 and 2 measured slower on gfx950 (ba.hip, point_step_kernel) and are kept for measurements. */
int vgg_ba_set_step_from_factors(int enable);
/* Round-6 measurement switch for the off-diagonal Schur tile launch with 6 x 6 blocks (two launches per batch only): 0 (default;
 * VGG_TILE_DMA seeds it) = segments staged through registers from the compressed records; 1 = LDS-DMA (global_load_lds_dwordx4)
 * from an EXPANDED image of the segments, which a kernel of its own derives from the compressed records in front of the launch
 * (the workspace grows by 2304 bytes per segment: query vgg_ba_workspace_bytes AFTER setting the mode); 2 = 1 + a touch of the
 * lines of the batch after next.  Same tile sums up to the rounding of the rebuilt rows.  Measured, not adopted: ba.hip,
 * schur_tile_dma_kernel. */
int vgg_ba_set_tile_dma(int mode);
int vgg_ba_profile(int enable, int max_launches_per_kernel);
int vgg_ba_profile_read(int kernel_id, double* total_ms, int* launches, int reset);

/* Dense symmetric positive-definite solve of the reduced camera system (exposed for tests):
 * A (n,n) row-major lower triangle is overwritten by its Cholesky factor, b by the solution
 * (if b == A + n*n the forward substitution is fused into the factorisation).
 * workspace: vgg_cholesky_workspace_bytes(n) device bytes (inverse diagonal blocks).
 * *device_fail (int32, device) is set non-zero on a non-positive pivot (2: a hand-off of the single-launch form timed out).
 * n >= 128 with b behind A runs as ONE launch (a workgroup per 64 x 64 tile, hand-offs through per-tile flags); the sums of
 * a tile are taken in a fixed order that depends on the structure passed (split / envelope) only: the factor is reproducible
 * run to run, and equal to the plain entry's up to the rounding of those sums (VGG_CHOL_CHAIN / VGG_CHOL_LEGACY in the
 * environment select the other forms for measurements). */
size_t vgg_cholesky_workspace_bytes(int n);
int vgg_cholesky_solve(double* A, double* b, int n, void* workspace, int32_t* device_fail, void* stream);
/* the same with a block-diagonal leading part: A[i][j] = 0 for split_a <= i < split_a + split_b, j < split_a (exposed
 * for tests; see vgg_ba_problem.chol_split_a) */
int vgg_cholesky_solve_split(double* A, double* b, int n, int split_a, int split_b, void* workspace, int32_t* device_fail,
                             void* stream);
/* the same with a row envelope: first_blk[r] (device, ceil(n / 64) entries) = first 64-column block in which rows
 * 64 r .. 64 r + 63 of A can be non-zero (see vgg_ba_problem.chol_first_blk); b must be stored directly behind A
 * (b == A + n * n) and n >= 128 for the envelope to be used (exposed for tests).  With an envelope the tiles are launched
 * in the order of their dependency depth and those outside it are not launched at all (one small set-up launch per call;
 * VGG_CHOL_TILE_MAP=0 in the environment keeps the (column, row) order for measurements) */
int vgg_cholesky_solve_envelope(double* A, double* b, int n, const int32_t* first_blk, void* workspace, int32_t* device_fail,
                                void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VGGSFM_AMD_H */
