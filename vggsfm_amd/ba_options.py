"""Option structs with the pycolmap 3.10 field names the reference touches
(vggsfm/utils/triangulation_helpers.py:626-635, vggsfm/models/triangulator.py:254-260,
vggsfm/utils/triangulation.py:324-331).  Defaults are COLMAP 3.10's / Ceres 2.x's."""
from dataclasses import dataclass, field


@dataclass
class SolverOptions:
    function_tolerance: float = 0.0
    gradient_tolerance: float = 1e-4
    parameter_tolerance: float = 0.0
    max_num_iterations: int = 100
    max_linear_solver_iterations: int = 200          # only meaningful for iterative solvers; kept for API parity
    max_num_consecutive_invalid_steps: int = 10
    jacobi_scaling: bool = True
    initial_trust_region_radius: float = 1e4
    max_trust_region_radius: float = 1e16
    min_trust_region_radius: float = 1e-32
    min_lm_diagonal: float = 1e-6
    max_lm_diagonal: float = 1e32
    min_relative_decrease: float = 1e-3


@dataclass
class BundleAdjustmentOptions:
    solver_options: SolverOptions = field(default_factory=SolverOptions)
    refine_focal_length: bool = True
    refine_principal_point: bool = False             # the kernels keep the principal point constant
    refine_extra_params: bool = True
    refine_extrinsics: bool = True
    print_summary: bool = False
    loss_function_type: str = "TRIVIAL"              # TRIVIAL | CAUCHY | HUBER | SOFT_L1
    loss_function_scale: float = 1.0


@dataclass
class AbsolutePoseRefinementOptions:
    gradient_tolerance: float = 1.0
    max_num_iterations: int = 100
    loss_function_scale: float = 1.0
    refine_focal_length: bool = False
    refine_extra_params: bool = False
    print_summary: bool = False


@dataclass
class RANSACOptions:
    max_error: float = 12.0                           # pixels (COLMAP AbsolutePoseEstimationOptions.ransac.max_error default)
    num_hypotheses: int = 1024                        # device path: fixed number of minimal samples (COLMAP: adaptive, <= 10000)


@dataclass
class AbsolutePoseEstimationOptions:
    """pycolmap.AbsolutePoseEstimationOptions as the reference sets it (triangulation.py:324-326, video_runner.py:988-989)."""
    estimate_focal_length: bool = False
    num_focal_length_samples: int = 30
    min_focal_length_ratio: float = 0.2
    max_focal_length_ratio: float = 5.0
    ransac: RANSACOptions = field(default_factory=RANSACOptions)


LOSS_ID = {"TRIVIAL": 0, "CAUCHY": 1, "HUBER": 2, "SOFT_L1": 3}
TERMINATION = {0: "NO_CONVERGENCE (iteration cap)", 1: "CONVERGENCE (gradient tolerance)",
               2: "CONVERGENCE (function tolerance)", 3: "CONVERGENCE (parameter tolerance)",
               4: "CONVERGENCE (trust region radius)", 5: "FAILURE (consecutive invalid steps)"}
