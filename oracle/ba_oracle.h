/* ORACLE (test infrastructure only) -- C interface of the Ceres/COLMAP-faithful CPU
 * restatement of the bundle adjustment.  See ba_oracle.c for provenance. PARITY UNPINNED. */
#ifndef BA_ORACLE_H
#define BA_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { BAO_LOSS_TRIVIAL = 0, BAO_LOSS_CAUCHY = 1, BAO_LOSS_HUBER = 2, BAO_LOSS_SOFT_L1 = 3 };
enum {
  BAO_NO_CONVERGENCE = 0,        /* iteration cap */
  BAO_CONVERGENCE_GRADIENT = 1,
  BAO_CONVERGENCE_FUNCTION = 2,
  BAO_CONVERGENCE_PARAMETER = 3,
  BAO_CONVERGENCE_RADIUS = 4,
  BAO_FAILURE = 5
};

typedef struct {
  int32_t num_cams, num_pts, num_obs, num_intr;
  int32_t camera_model;          /* 0 SIMPLE_PINHOLE (f,cx,cy)  1 SIMPLE_RADIAL (f,cx,cy,k) */
  int32_t refine_focal, refine_extra;
  int32_t loss;                  /* BAO_LOSS_* */
  double loss_scale;
  const int32_t* cam_intr;       /* [num_cams] intrinsics block of each camera */
  const int32_t* row_ptr;        /* [num_pts+1] CSR by point */
  const int32_t* obs_cam;        /* [num_obs] */
  const double* obs_uv;          /* [num_obs*2] pixels */
  const uint8_t* cam_const;      /* [num_cams] bit0 whole pose; bit1..3 t_x,t_y,t_z; may be NULL */
  const uint8_t* intr_const;     /* [num_intr] or NULL */
  const uint8_t* pt_const;       /* [num_pts] or NULL */
} bao_problem_t;

typedef struct {
  int32_t max_num_iterations;
  int32_t max_num_consecutive_invalid_steps;
  int32_t jacobi_scaling;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius;
  double min_lm_diagonal, max_lm_diagonal, min_relative_decrease;
} bao_options_t;

typedef struct {
  int32_t iteration;
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, radius;
  int32_t successful;
} bao_iter_t;

typedef struct {
  double initial_cost, final_cost;
  int32_t num_iterations, num_successful_steps, num_unsuccessful_steps, termination, n_reduced, num_log;
} bao_summary_t;

/* cam_q [C*4] (x,y,z,w), cam_t [C*3], intr [num_intr*4] (f,cx,cy,k), pts [P*3]: in/out */
int bao_solve(const bao_problem_t* pb, const bao_options_t* opt, double* cam_q, double* cam_t, double* intr,
              double* pts, bao_summary_t* summary, bao_iter_t* log, int log_cap);
int bao_num_threads(void);
void bao_obs_eval(int model, const double* q, const double* t, const double* intr, const double* X, double u, double v,
                  double* r, double* Jp, double* Ji, double* Jx);
void bao_quat_plus(const double* x, const double* d, double* out);
/* test hook: copy the first reduced camera system (lhs n*n full, rhs n) of the next bao_solve call */
void bao_debug_dump_system(double* lhs, double* rhs);
#ifdef __cplusplus
}
#endif
#endif
