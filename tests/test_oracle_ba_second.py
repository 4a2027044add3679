"""Pins oracle/ba_oracle.c (analytic Jacobians, Schur elimination, C) against oracle/ba_autograd.py (autograd
Jacobians, dense normal equations, numpy): two derivations that share no arithmetic must walk the same
Levenberg-Marquardt trajectory.  VERDICT r2 "What's missing" 1 / "Next round" 1c.  Neither is pycolmap -- the BA half
of the oracle stays PARITY UNPINNED vs the reference's third-party solver -- but a sign, a factor 2 of the half-angle
tangent, a wrong corrector, a wrong damping or a wrong acceptance rule in either would show up here."""
import numpy as np
import pytest

from oracle import ba as OB
from oracle import ba_autograd as OA
from vggsfm_amd.scene import make_scene, perturb_for_ba


def _both(monkeypatch, run):
    """Run `run()` (which ends in OB.solve_csr) with a wrapper that solves the SAME raw problem with both oracles."""
    got = {}
    c_solve = OB.solve_csr

    def wrapper(cam_q, cam_t, intr, pts, cam_intr, row_ptr, obs_cam, obs_uv, camera_model, options, **kw):
        kw.pop("log_cap", None)
        a = [x.copy() for x in (cam_q, cam_t, intr, pts)]
        got["second"] = OA.solve_csr(*a, cam_intr, row_ptr, obs_cam, obs_uv, camera_model, options, **kw)
        got["second_x"] = a
        got["first"] = c_solve(cam_q, cam_t, intr, pts, cam_intr, row_ptr, obs_cam, obs_uv, camera_model, options, **kw)
        got["first_x"] = [x.copy() for x in (cam_q, cam_t, intr, pts)]
        return got["first"]
    monkeypatch.setattr(OB, "solve_csr", wrapper)
    run()
    return got


def _compare(got, min_compared=4, x_tol=1e-6):
    a, b = got["first"], got["second"]
    assert a["n_reduced"] == b["n_reduced"]
    assert abs(a["initial_cost"] - b["initial_cost"]) <= 1e-12 * b["initial_cost"]
    compared, g0 = 0, b["iterations"][0]["gradient_max_norm"]
    for u, v in zip(a["iterations"], b["iterations"]):
        # (below |cost change| = 1e-9 cost the step quality is rounding noise on both sides)
        if v["iteration"] > 0 and abs(v["cost_change"]) < 1e-9 * v["cost"]:
            break
        assert u["iteration"] == v["iteration"] and u["successful"] == v["successful"], (u, v)
        assert abs(u["cost"] - v["cost"]) <= 1e-10 * v["cost"], (u, v)
        assert abs(u["radius"] - v["radius"]) <= 1e-6 * v["radius"], (u, v)
        # (the gradient near the optimum is a difference of terms ~g0: absolute floor 1e-11 g0; the step quality is
        #  cost change / model change and the cost change carries ~1e-15 cost of rounding)
        assert abs(u["gradient_max_norm"] - v["gradient_max_norm"]) <= 1e-7 * v["gradient_max_norm"] + 1e-11 * g0, (u, v)
        assert abs(u["step_norm"] - v["step_norm"]) <= 1e-6 * v["step_norm"] + 1e-12, (u, v)
        if v["iteration"] > 0 and v["cost_change"] != 0:
            tol = 1e-6 * max(1.0, abs(v["relative_decrease"])) + 1e-14 * v["cost"] / abs(v["cost_change"])
            assert abs(u["relative_decrease"] - v["relative_decrease"]) <= tol, (u, v)
        compared += 1
    assert compared >= min(min_compared, b["num_iterations"] + 1), (compared, b["num_iterations"])
    if compared == len(b["iterations"]) == len(a["iterations"]):      # (no noise-floor cut: the exits must agree too)
        assert a["termination"] == b["termination"] and a["num_iterations"] == b["num_iterations"]
    assert abs(a["final_cost"] - b["final_cost"]) <= 1e-8 * b["final_cost"]
    for x, y in zip(got["first_x"], got["second_x"]):
        np.testing.assert_allclose(x, y, rtol=0, atol=x_tol * max(1.0, np.abs(y).max()))


@pytest.mark.parametrize("S,N,cam,shared,kind", [
    (6, 60, "SIMPLE_PINHOLE", False, "prep"),
    (7, 120, "SIMPLE_RADIAL", True, "prep"),
    (9, 90, "SIMPLE_RADIAL", False, "default"),
    (5, 100, "SIMPLE_PINHOLE", True, "default"),
    (2, 80, "SIMPLE_PINHOLE", False, "prep"),           # init_BA shape
])
def test_bundle_adjustment_trajectories_agree(monkeypatch, S, N, cam, shared, kind):
    sc = make_scene(S, N, cam, shared_camera=shared, seed=S + N, full_visibility=(S <= 3))
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=S + N)
    opts = OB.prepare_ba_options() if kind == "prep" else OB.ceres_options()
    got = _both(monkeypatch, lambda: OB.bundle_adjustment(pts0, ext0, K0, sc.tracks, sc.mask, extra0, shared, cam, opts))
    _compare(got)


@pytest.mark.parametrize("loss,scale", [(1, 2.0), (2, 1.0), (3, 1.5)])     # Cauchy, Huber, SoftL1
def test_robust_loss_trajectories_agree(monkeypatch, loss, scale):
    sc = make_scene(7, 150, "SIMPLE_RADIAL", shared_camera=True, seed=17, outlier_frac=0.08)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=17)
    got = _both(monkeypatch, lambda: OB.bundle_adjustment(pts0, ext0, K0, sc.tracks, sc.mask, extra0, True, "SIMPLE_RADIAL",
                                                          OB.ceres_options(30), loss=loss, loss_scale=scale))
    _compare(got)


@pytest.mark.parametrize("shared", [True, False])
@pytest.mark.parametrize("rf,rk", [(True, False), (False, True), (False, False)])
def test_partial_intrinsics_trajectories_agree(monkeypatch, shared, rf, rk):
    sc = make_scene(6, 100, "SIMPLE_RADIAL", shared_camera=shared, seed=19, outlier_frac=0.0)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=19)
    got = _both(monkeypatch, lambda: OB.bundle_adjustment(pts0, ext0, K0, sc.tracks, sc.mask, extra0, shared, "SIMPLE_RADIAL",
                                                          OB.ceres_options(25), refine_focal=rf, refine_extra=rk))
    _compare(got)


def test_window_ba_constant_blocks_agree(monkeypatch):
    """video_runner.py:800-838: frame 0 constant, carried-over points constant, intrinsics fixed, no depth filter."""
    sc = make_scene(9, 160, "SIMPLE_RADIAL", shared_camera=True, seed=12, outlier_frac=0.0)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=12)
    ext0[0] = sc.extrinsics[0]
    valid = sc.mask.sum(0) >= 2
    constant = valid & (np.cumsum(valid) - 1 < 70)
    pts0[constant] = sc.points3D[constant]
    got = _both(monkeypatch, lambda: OB.bundle_adjustment(pts0, ext0, K0, sc.tracks, sc.mask, extra0, True, "SIMPLE_RADIAL",
                                                          options=OB.ceres_options(100), constant_points=constant,
                                                          constant_pose_frames=[0], filter_negative_depth=False,
                                                          refine_focal=False, refine_extra=False))
    _compare(got)


@pytest.mark.parametrize("cam,rf", [("SIMPLE_RADIAL", True), ("SIMPLE_PINHOLE", True), ("SIMPLE_RADIAL", False)])
def test_pose_refinement_trajectories_agree(monkeypatch, cam, rf):
    """pycolmap.pose_refinement (triangulation.py:387,590): one camera, constant points, CauchyLoss(1), Ceres default
    tolerances + gradient_tolerance 1."""
    sc = make_scene(4, 200, cam, shared_camera=True, seed=3, full_visibility=True, outlier_frac=0.1)
    ext0, K0, extra0, _ = perturb_for_ba(sc, seed=3)
    intr = np.array([K0[2, 0, 0], K0[2, 0, 2], K0[2, 1, 2], 0.0 if extra0 is None else extra0[2, 0]])
    inl = np.ones(sc.tracks.shape[1], bool)
    inl[::7] = False
    got = _both(monkeypatch, lambda: OB.pose_refinement(ext0[2], sc.tracks[2].astype(np.float64), sc.points3D, inl,
                                                        intr[:4 if cam == "SIMPLE_RADIAL" else 3], cam, rf, rf))
    _compare(got, min_compared=3)
