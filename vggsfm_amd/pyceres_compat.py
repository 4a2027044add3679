"""The two `pyceres` names the reference touches (vggsfm/runners/video_runner.py:1329-1330): ``SolverSummary`` and
``solve(options, problem, summary)``.  `problem` is what ``vggsfm_amd.pycolmap_compat.BundleAdjuster.problem`` holds;
the solve runs on the GPU (``vgg_ba_solve``) and fills the summary fields ``log_ba_summary`` reads
(video_runner.py:1300-1318)."""
from .ba_options import TERMINATION


class SolverSummary:
    def __init__(self):
        self.num_residuals_reduced = 0
        self.num_effective_parameters_reduced = 0
        self.num_successful_steps = 0
        self.num_unsuccessful_steps = 0
        self.total_time_in_seconds = 0.0
        self.initial_cost = 0.0
        self.final_cost = 0.0
        self.termination_type = -1
        self.message = ""
        self.iterations = []

    def BriefReport(self):
        return (f"Ceres-style LM on MI355X: iterations {self.num_successful_steps + self.num_unsuccessful_steps}, "
                f"initial cost {self.initial_cost:.6e}, final cost {self.final_cost:.6e}, termination: {self.message}")

    FullReport = BriefReport


def solve(options, problem, summary):
    """pyceres.solve: run the bundle adjuster's problem with `options` (a ``solver_options`` struct)."""
    out = problem.adjuster._run(options)
    summary.num_residuals_reduced = out["num_residuals"]
    summary.num_effective_parameters_reduced = out["n_reduced"]
    summary.num_successful_steps = out["num_successful_steps"]
    summary.num_unsuccessful_steps = out["num_unsuccessful_steps"]
    summary.total_time_in_seconds = out["total_time_in_seconds"]
    summary.initial_cost, summary.final_cost = out["initial_cost"], out["final_cost"]
    summary.termination_type = out["termination"]
    summary.message = TERMINATION.get(out["termination"], "?")
    summary.iterations = out["iterations"]
    return summary
