"""Cut line B2 on the device: ``vggsfm_amd.pycolmap_compat``'s solver entry points -- ``bundle_adjustment``,
``BundleAdjuster`` + ``pyceres.solve``, ``pose_refinement``, ``absolute_pose_estimation``, ``ObservationManager`` -- run on
the GPU kernels and are compared with the CPU oracle on reconstructions built through the pycolmap API the way the
reference's loops build them (vggsfm/utils/tensor_to_pycolmap.py:62-158, vggsfm/runners/video_runner.py:543-605,
800-838, 961-1011).  (The reference tree is absent on the GPU box; the reference's unmodified code on top of the same
object model runs in tests/test_pycolmap_compat.py.)"""
import copy

import numpy as np
import pytest
import torch

from oracle import ba as OB
from tests import cpu_backend
from tests.test_pycolmap_compat import _api_build
from vggsfm_amd import pyceres_compat as pyceres
from vggsfm_amd import pycolmap_compat as pc
from vggsfm_amd.scene import make_scene, perturb_for_ba

pytestmark = pytest.mark.gpu


def _problem(S=12, N=300, cam="SIMPLE_RADIAL", shared=True, seed=11):
    sc = make_scene(S, N, cam, shared_camera=shared, seed=seed, outlier_frac=0.0)
    ext0, K0, xp0, pts0 = perturb_for_ba(sc, seed=seed)
    rec = _api_build(pts0, ext0, K0, sc.tracks, sc.mask, [1024, 1024], shared, cam, xp0)
    return sc, rec, (pts0, ext0, K0, xp0)


def _state(rec):
    ids = sorted(rec.images)
    ext = np.stack([rec.images[i].cam_from_world.matrix() for i in ids])
    prm = np.stack([rec.cameras[rec.images[i].camera_id].params for i in ids])
    return ext, prm, rec._xyz[:rec._n].copy(), rec._alive[:rec._n].copy()


@pytest.mark.parametrize("cam,shared", [("SIMPLE_PINHOLE", False), ("SIMPLE_RADIAL", True)])
def test_bundle_adjustment_on_device_matches_oracle(cam, shared, monkeypatch):
    sc, rec, _ = _problem(cam=cam, shared=shared)
    rec_cpu = copy.deepcopy(rec)
    opts = pc.BundleAdjustmentOptions()
    opts.solver_options.max_num_iterations = 4           # (iteration-capped: the iteration on which the gradient tolerance
    opts.solver_options.gradient_tolerance = 0.0         #  is met is a knife edge between two implementations)
    summ = pc.bundle_adjustment(rec, opts)                               # GPU
    with monkeypatch.context() as m:
        cpu_backend.patch(m)
        summ_cpu = pc.bundle_adjustment(rec_cpu, opts)                   # same entry, oracle arithmetic
    assert summ["num_iterations"] == summ_cpu["num_iterations"] and summ["termination"] == summ_cpu["termination"]
    assert abs(summ["final_cost"] - summ_cpu["final_cost"]) <= 1e-9 * summ_cpu["final_cost"]
    for a, b in zip(_state(rec), _state(rec_cpu)):
        np.testing.assert_allclose(a, b, rtol=1e-7, atol=1e-8)
    assert summ["final_cost"] < 0.2 * summ["initial_cost"]


def test_bundle_adjuster_config_and_pyceres_solve(monkeypatch):
    """The window BA of the video runner (video_runner.py:813-836, 1321-1331): every image in the config, the first one's
    pose constant, the first 100 points constant, intrinsics not refined."""
    sc, rec, (pts0, ext0, K0, xp0) = _problem(S=9, N=260)
    rec_cpu = copy.deepcopy(rec)

    def run(r):
        o = pc.BundleAdjustmentOptions()
        o.solver_options.max_num_iterations = 4                # (well before convergence: see above)
        o.solver_options.gradient_tolerance = 0.0
        o.refine_focal_length = False
        o.refine_extra_params = False
        cfg = pc.BundleAdjustmentConfig()
        for i in r.reg_image_ids():
            cfg.add_image(i)
        cfg.set_constant_cam_pose(r.reg_image_ids()[0])
        for p in r.point3D_ids():
            (cfg.add_constant_point if p < 101 else cfg.add_variable_point)(p)
        adj = pc.BundleAdjuster(o, cfg)
        adj.set_up_problem(r, o.create_loss_function())
        so = adj.set_up_solver_options(adj.problem, o.solver_options)
        summary = pyceres.SolverSummary()
        pyceres.solve(so, adj.problem, summary)
        return summary

    s_gpu = run(rec)
    with monkeypatch.context() as m:
        cpu_backend.patch(m)
        s_cpu = run(rec_cpu)
    assert s_gpu.num_residuals_reduced == s_cpu.num_residuals_reduced > 0
    assert s_gpu.num_successful_steps + s_gpu.num_unsuccessful_steps == s_cpu.num_successful_steps + s_cpu.num_unsuccessful_steps
    assert abs(s_gpu.final_cost - s_cpu.final_cost) <= 1e-9 * s_cpu.final_cost
    e, prm, xyz, _ = _state(rec)
    np.testing.assert_array_equal(e[0], ext0[0])                          # constant pose
    np.testing.assert_array_equal(xyz[:100], rec_cpu._xyz[:100])          # constant points untouched on both sides
    valid = np.nonzero(sc.mask.sum(0) >= 2)[0]
    np.testing.assert_array_equal(xyz[:100], pts0[valid][:100])
    assert prm[0, 0] == K0[0, 0, 0]                                       # intrinsics not refined
    for a, b in zip(_state(rec), _state(rec_cpu)):
        np.testing.assert_allclose(a, b, rtol=1e-7, atol=1e-8)


def test_pose_refinement_and_absolute_pose_estimation():
    sc = make_scene(6, 400, "SIMPLE_RADIAL", shared_camera=True, seed=21, outlier_frac=0.0)
    ext0, K0, xp0, _ = perturb_for_ba(sc, seed=21, rot_deg=1.0, trans=0.05)
    f = 3
    cam = pc.Camera("SIMPLE_RADIAL", 1024, 1024, [K0[f, 0, 0], 512.0, 512.0, xp0[f, 0]], 0)
    cam_cpu = copy.deepcopy(cam)
    ro = pc.AbsolutePoseRefinementOptions()
    ro.refine_focal_length, ro.refine_extra_params = True, True
    T0 = pc.Rigid3d(pc.Rotation3d(ext0[f][:3, :3]), ext0[f][:3, 3])
    ans = pc.pose_refinement(T0, sc.tracks[f].astype(np.float64), sc.points3D, sc.mask[f], cam, ro)
    e_ref, p_ref, _ = OB.pose_refinement(ext0[f], sc.tracks[f].astype(np.float64), sc.points3D, sc.mask[f], cam_cpu.params,
                                         "SIMPLE_RADIAL", True, True)
    np.testing.assert_allclose(ans["cam_from_world"].matrix(), e_ref, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(cam.params, p_ref[:4], rtol=1e-7)
    np.testing.assert_allclose(ans["cam_from_world"].matrix(), sc.extrinsics[f], atol=5e-3)
    # P3P RANSAC + refinement from scratch, on the visible matches with a few gross outliers
    rng = np.random.default_rng(0)
    vis = np.nonzero(sc.mask[f])[0]
    p2 = sc.tracks[f][vis].astype(np.float64)
    bad = rng.choice(len(vis), len(vis) // 10, replace=False)
    p2[bad] += rng.uniform(50, 200, (len(bad), 2))
    cam2 = pc.Camera("SIMPLE_RADIAL", 1024, 1024, [sc.intrinsics[f, 0, 0], 512.0, 512.0, sc.extra_params[f, 0]], 0)
    eo = pc.AbsolutePoseEstimationOptions()
    eo.ransac.max_error = 12
    est = pc.absolute_pose_estimation(p2, sc.points3D[vis], cam2, eo, pc.AbsolutePoseRefinementOptions())
    assert est is not None and est["num_inliers"] >= 0.85 * len(vis) and not est["inliers"][bad].any()
    np.testing.assert_allclose(est["cam_from_world"].matrix(), sc.extrinsics[f], atol=5e-3)
    assert pc.absolute_pose_estimation(p2[:2], sc.points3D[vis][:2], cam2, eo) is None


def test_observation_manager_matches_oracle_filters(monkeypatch):
    sc, rec, _ = _problem(S=10, N=400, seed=5)
    # spoil some observations and push one point behind the cameras
    rng = np.random.default_rng(1)
    for i in (2, 5, 7):
        p2 = rec.images[i].points2D
        k = rng.choice(len(p2), 25, replace=False)
        p2._xy[k] += rng.uniform(20, 60, (25, 2))
    rec.points3D[9].xyz = [0.0, 0.0, -50.0]
    rec_cpu = copy.deepcopy(rec)
    om = pc.ObservationManager(rec)
    n1 = om.filter_all_points3D(2.0, 1.5)
    n2 = om.filter_observations_with_negative_depth()
    with monkeypatch.context() as m:
        cpu_backend.patch(m)
        omc = pc.ObservationManager(rec_cpu)
        c1 = omc.filter_all_points3D(2.0, 1.5)
        c2 = omc.filter_observations_with_negative_depth()
    assert (n1, n2) == (c1, c2) and n1 >= 75 and 9 not in rec.points3D
    assert rec.point3D_ids() == rec_cpu.point3D_ids()
    for i in rec.images:
        assert np.array_equal(rec.images[i].points2D._pid, rec_cpu.images[i].points2D._pid)


def test_dropin_reconstruction_is_a_pycolmap_object(tmp_path):
    """What ``vggsfm_amd.models.Triangulator`` returns can be edited and saved the way the reference's runner does
    (runner.py:555-559, 575, 592-611, 1020-1052, 911) -- here through the same calls, on the GPU box."""
    import glob
    import os
    import types
    from vggsfm_amd.models import Triangulator
    g = np.load(sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "triangulator_pinhole_s10.npz")))[0])
    dev = "cuda"
    W = int(g["W"])
    t = torch.from_numpy(g["images_small"])
    r = W // t.shape[-1]
    images = t.repeat_interleave(r, dim=-1).repeat_interleave(r, dim=-2).to(dev)
    cams = types.SimpleNamespace(R=torch.from_numpy(g["R"]).to(dev), T=torch.from_numpy(g["T"]).to(dev),
                                 focal_length=torch.from_numpy(np.stack([g["focal_ndc"]] * 2, -1)).to(dev))
    torch.manual_seed(0)
    out = Triangulator()(cams, torch.from_numpy(g["tracks"])[None].to(dev), torch.from_numpy(g["vis"])[None].to(dev), images,
                         {"fmat_inlier_mask": torch.from_numpy(g["fmat_inlier"])[None].to(dev)},
                         pred_score=torch.from_numpy(g["score"])[None].to(dev), BA_iters=2, robust_refine=2)
    rec = out[5]
    # (the P3P fallback of refine_pose runs here, unlike in the golden comparison: the counts need not be the golden ones)
    assert isinstance(rec, pc.Reconstruction) and rec.num_points3D() == len(out[3]) > 0.8 * g["tracks"].shape[1]
    n = rec.num_points3D()
    pid = rec.add_point3D(np.array([0.1, 0.2, 0.3]), pc.Track(), np.array([1, 2, 3]))
    assert pid == n + 1
    rec.deregister_image(9)
    for i in rec.images:
        im, cam = rec.images[i], rec.cameras[rec.images[i].camera_id]
        im.name = f"frame_{i}.png"
        prm = copy.deepcopy(cam.params)
        prm[0] *= 2.0
        prm[1:3] = [1024, 768]
        cam.params, cam.width, cam.height = prm, torch.tensor(2048.0), torch.tensor(1536.0)
        for p2 in im.points2D:
            p2.xy = (p2.xy - np.array([0.0, 128.0])) * 2.0
    assert rec.cameras[3].calibration_matrix()[1, 2] == 768.0
    el = rec.points3D[1].track.elements
    assert len(el) >= 2 and rec.images[el[0].image_id].points2D[el[0].point2D_idx].point3D_id == 1
    rec.write(str(tmp_path))
    back = pc.Reconstruction(str(tmp_path))
    assert back.num_reg_images() == 9 and back.num_points3D() == rec.num_points3D()
    assert back.images[0].name == "frame_0.png" and back.cameras[0].width == 2048
    assert np.array_equal(back.images[4].points2D._xy, rec.images[4].points2D._xy)


@pytest.mark.parametrize("cam,shared", [("SIMPLE_PINHOLE", False), ("SIMPLE_RADIAL", True)])
def test_batch_matrix_to_pycolmap_device_selection_equals_host_construction(cam, shared):
    """``batch_matrix_to_pycolmap`` on GPU tensors selects the observations on the device and ships the lists
    (``Reconstruction.from_frame_lists``); same model as the host construction ``Reconstruction.from_arrays`` (which
    tests/test_pycolmap_compat.py pins to the reference's own loop), including the two exclusion rules: tracks with fewer
    than two observations and points with a coordinate >= max_points3D_val."""
    from vggsfm_amd.utils.tensor_to_pycolmap import batch_matrix_to_pycolmap
    sc = make_scene(14, 700, cam, shared_camera=shared, seed=21)
    ext0, K0, xp0, pts0 = perturb_for_ba(sc, seed=21)
    mask = sc.mask.copy()
    mask[1:, 5] = False                                # a track with one observation
    pts0[9, 1] = 4000.0                                # a point beyond the cap: kept, without observations
    D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
    size = torch.tensor([1024, 1024])
    a = batch_matrix_to_pycolmap(D(pts0), D(ext0), D(K0), D(sc.tracks), D(mask), size, shared_camera=shared, camera_type=cam,
                                 extra_params=D(xp0))
    b = pc.Reconstruction.from_arrays(pts0, ext0, K0, sc.tracks, mask, size.numpy(), 3000, shared, cam, xp0)
    assert a.num_points3D() == b.num_points3D() and np.array_equal(a.valid_idx, b.valid_idx)
    assert np.array_equal(a._xyz[:a._n], b._xyz[:b._n]) and sorted(a.cameras) == sorted(b.cameras)
    for i in sorted(b.images):
        assert np.array_equal(a.images[i].points2D._pid, b.images[i].points2D._pid)
        assert np.array_equal(a.images[i].points2D._xy, b.images[i].points2D._xy)
        assert np.array_equal(a.images[i].cam_from_world.matrix(), b.images[i].cam_from_world.matrix())
        assert a.images[i].camera_id == b.images[i].camera_id
        assert np.array_equal(a.cameras[a.images[i].camera_id].params, b.cameras[b.images[i].camera_id].params)
    pa, ia, xa = a._track_csr()
    pb, ib, xb = b._track_csr()
    assert np.array_equal(pa, pb) and np.array_equal(ia, ib) and np.array_equal(xa, xb)
