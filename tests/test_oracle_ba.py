"""Checks on the C restatement of COLMAP/Ceres BA (oracle/ba_oracle.c).  PARITY UNPINNED vs
pycolmap (absent here); these tests pin the mathematics: analytic Jacobians vs finite
differences, the LM optimum vs scipy.optimize.least_squares, Ceres control-flow invariants."""
import ctypes

import numpy as np
import pytest
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation

from oracle import ba as OB
from vggsfm_amd.scene import make_scene, perturb_for_ba, project


def _c(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _obs_eval(model, q, t, intr, X, uv):
    L = OB.lib()
    r = np.zeros(2)
    Jp = np.zeros(12)
    Ji = np.zeros(4)
    Jx = np.zeros(6)
    L.bao_obs_eval(ctypes.c_int(model), _c(q), _c(t), _c(intr), _c(X), ctypes.c_double(uv[0]),
                   ctypes.c_double(uv[1]), _c(r), _c(Jp), _c(Ji), _c(Jx))
    return r, Jp.reshape(2, 6), Ji.reshape(2, 2), Jx.reshape(2, 3)


def _quat_plus(q, d):
    out = np.zeros(4)
    OB.lib().bao_quat_plus(_c(q), _c(np.ascontiguousarray(d)), _c(out))
    return out


@pytest.mark.parametrize("model", [0, 1])
def test_jacobians_match_finite_differences(model):
    rng = np.random.default_rng(0)
    for _ in range(20):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        t = rng.normal(size=3) * 0.3 + np.array([0, 0, 4.0])
        intr = np.array([900.0 + 200 * rng.uniform(), 512.0, 512.0, 0.08 * rng.normal()])
        X = rng.uniform(-1, 1, size=3)
        uv = rng.uniform(0, 1024, size=2)
        r0, Jp, Ji, Jx = _obs_eval(model, q, t, intr, X, uv)
        h = 1e-6
        for k in range(3):
            d = np.zeros(3)
            d[k] = h
            rp = _obs_eval(model, _quat_plus(q, d), t, intr, X, uv)[0]
            rm = _obs_eval(model, _quat_plus(q, -d), t, intr, X, uv)[0]
            np.testing.assert_allclose((rp - rm) / (2 * h), Jp[:, k], rtol=1e-6, atol=1e-4)
            rp = _obs_eval(model, q, t + d, intr, X, uv)[0]
            rm = _obs_eval(model, q, t - d, intr, X, uv)[0]
            np.testing.assert_allclose((rp - rm) / (2 * h), Jp[:, 3 + k], rtol=1e-6, atol=1e-4)
            rp = _obs_eval(model, q, t, intr, X + d, uv)[0]
            rm = _obs_eval(model, q, t, intr, X - d, uv)[0]
            np.testing.assert_allclose((rp - rm) / (2 * h), Jx[:, k], rtol=1e-6, atol=1e-4)
        for k, idx in enumerate([0, 3] if model == 1 else [0]):
            d = np.zeros(4)
            d[idx] = h
            rp = _obs_eval(model, q, t, intr + d, X, uv)[0]
            rm = _obs_eval(model, q, t, intr - d, X, uv)[0]
            np.testing.assert_allclose((rp - rm) / (2 * h), Ji[:, k], rtol=1e-6, atol=1e-4)


def test_quaternion_roundtrip():
    R = Rotation.random(50, random_state=2).as_matrix()
    q = OB.rotmat_to_quat(R)
    np.testing.assert_allclose(OB.quat_to_rotmat(q), R, atol=1e-14)
    qs = Rotation.from_matrix(R).as_quat()        # scipy is (x,y,z,w) too; sign is free
    s = np.sign((q * qs).sum(-1))[:, None]
    np.testing.assert_allclose(q * s, qs, atol=1e-14)


def _scipy_ba(sc, ext0, K0, extra0, pts0, masks):
    """Independent solve of the same gauge-fixed problem (image 0 fixed, t_x of image 1 fixed)."""
    S, N = masks.shape
    radial = extra0 is not None
    shared = sc.shared_camera
    nf = 1 if shared else S

    def unpack(x):
        i = 0
        ext = ext0.copy()
        for s in range(1, S):
            w = x[i:i + 3]
            i += 3
            ext[s, :, :3] = Rotation.from_rotvec(w).as_matrix() @ ext0[s, :, :3]
            if s == 1:
                ext[s, 1:, 3] = ext0[s, 1:, 3] + x[i:i + 2]
                i += 2
            else:
                ext[s, :, 3] = ext0[s, :, 3] + x[i:i + 3]
                i += 3
        f = x[i:i + nf]
        i += nf
        K = K0.copy()
        K[:, 0, 0] = K[:, 1, 1] = K0[:, 0, 0] * (f[0] if shared else f)
        extra = None
        if radial:
            kk = x[i:i + nf]
            i += nf
            extra = (np.full((S, 1), kk[0]) if shared else kk[:, None])
        pts = pts0 + x[i:].reshape(-1, 3)
        return ext, K, extra, pts

    def fun(x):
        ext, K, extra, pts = unpack(x)
        uv, _ = project(pts, ext, K, extra)
        return (uv - sc.tracks.astype(np.float64))[masks].ravel()

    nx = 6 * (S - 1) - 1 + nf * (2 if radial else 1) + 3 * N
    x0 = np.zeros(nx)
    off = 6 * (S - 1) - 1
    x0[off:off + nf] = 1.0
    if radial:
        x0[off + nf:off + 2 * nf] = (extra0[0, 0] if shared else extra0[:, 0])
    sol = least_squares(fun, x0, method="trf", x_scale="jac", xtol=1e-15, ftol=1e-15, gtol=1e-13, max_nfev=300)
    ext, K, extra, pts = unpack(sol.x)
    return ext, K, extra, pts, 0.5 * float((sol.fun ** 2).sum())


@pytest.mark.parametrize("cam_type,shared", [("SIMPLE_PINHOLE", False), ("SIMPLE_RADIAL", True)])
def test_ba_optimum_matches_scipy(cam_type, shared):
    sc = make_scene(6, 60, cam_type, shared_camera=shared, seed=7, outlier_frac=0.0)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=7)
    masks = sc.mask
    assert (masks.sum(0) >= 2).all()
    opt = OB.ceres_options(200, 0.0, 1e-10, 0.0)
    pts, ext, K, extra, summ = OB.bundle_adjustment(pts0, ext0, K0, sc.tracks, masks, extra0, shared, cam_type, opt)
    e2, K2, x2, p2, cost2 = _scipy_ba(sc, ext0, K0, extra0, pts0, masks)
    assert summ["final_cost"] <= summ["initial_cost"]
    np.testing.assert_allclose(summ["final_cost"], cost2, rtol=1e-9)
    np.testing.assert_allclose(ext, e2, atol=2e-6)
    np.testing.assert_allclose(K[:, 0, 0], K2[:, 0, 0], rtol=1e-6)
    np.testing.assert_allclose(pts, p2, atol=5e-6)
    if extra is not None:
        np.testing.assert_allclose(extra, x2, atol=1e-6)
    # gauge: image 0 untouched (up to the R->q->R round trip), t_x of image 1 untouched
    np.testing.assert_allclose(ext[0], ext0[0], atol=1e-15)
    assert ext[1, 0, 3] == ext0[1, 0, 3]


def test_lm_control_flow_invariants():
    sc = make_scene(8, 120, "SIMPLE_PINHOLE", seed=9)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=9)
    pts, ext, K, extra, summ = OB.bundle_adjustment(pts0, ext0, K0, sc.tracks, sc.mask, None, False,
                                                    "SIMPLE_PINHOLE", OB.prepare_ba_options())
    its = summ["iterations"]
    assert its[0]["iteration"] == 0 and its[0]["radius"] == 1e4
    assert summ["num_iterations"] <= 50
    assert summ["num_successful_steps"] + summ["num_unsuccessful_steps"] == summ["num_iterations"]
    prev = its[0]
    for it in its[1:]:
        if it["successful"]:
            assert it["cost"] < prev["cost"]
            assert it["relative_decrease"] > 1e-3
            assert it["radius"] <= min(1e16, prev["radius"] * 3 * (1 + 1e-12))   # StepAccepted grows <= 3x
        else:
            assert it["cost"] == prev["cost"]
            assert it["radius"] < prev["radius"]
        prev = it


def test_negative_depth_filter_and_deleted_points():
    sc = make_scene(5, 40, "SIMPLE_PINHOLE", seed=4, full_visibility=True, outlier_frac=0.0)
    ext0, K0, _, pts0 = perturb_for_ba(sc, seed=4)
    masks = sc.mask.copy()
    masks[2:, 0] = False               # track 0 has exactly two observations ...
    pts0[0] = [0.0, 0.0, -3.0]         # ... and is behind camera 0 -> whole point deleted
    pts0[1, 2] = 3500.0                # >= 3000: no observations at all
    pts, ext, K, extra, summ = OB.bundle_adjustment(pts0, ext0, K0, sc.tracks, masks, None, False,
                                                    "SIMPLE_PINHOLE", OB.prepare_ba_options())
    assert summ["deleted"][0] and (pts[0] == 0).all()
    assert (pts[1] == pts0[1]).all()    # untouched: not part of the problem
    assert np.isfinite(pts).all() and summ["final_cost"] < summ["initial_cost"]


def test_pose_refinement_recovers_pose():
    sc = make_scene(4, 300, "SIMPLE_RADIAL", shared_camera=True, seed=6, full_visibility=True, outlier_frac=0.05)
    ext0, K0, extra0, _ = perturb_for_ba(sc, seed=6, rot_deg=1.0, trans=0.05)
    intr = np.array([K0[2, 0, 0], 512.0, 512.0, extra0[2, 0]])
    ext, intr_out, summ = OB.pose_refinement(ext0[2], sc.tracks[2], sc.points3D, sc.mask[2], intr, "SIMPLE_RADIAL")
    assert summ["final_cost"] < summ["initial_cost"]
    # robust loss: the 5 % outliers do not pull the pose away from the ground truth
    assert np.abs(ext - sc.extrinsics[2]).max() < 5e-3
    assert abs(intr_out[0] - 1000.0) < 5.0


def test_pycolmap_shim_round_trip_equals_dense_oracle():
    """oracle/pycolmap_shim.py (the B2 harness behind tests/golden/triangulator_*.npz): a Reconstruction filled the
    way vggsfm/utils/tensor_to_pycolmap.py:60-158 fills it must give the same BA result as the dense oracle entry."""
    from oracle import pycolmap_shim as pc
    from vggsfm_amd.scene import make_scene, perturb_for_ba

    sc = make_scene(6, 80, "SIMPLE_RADIAL", shared_camera=True, seed=5)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=5)
    valid = np.nonzero(sc.mask.sum(0) >= 2)[0]
    rec = pc.Reconstruction()
    for v in valid:
        rec.add_point3D(pts0[v], pc.Track(), np.zeros(3))
    cam = pc.Camera(model="SIMPLE_RADIAL", width=1024, height=1024,
                    params=[K0[0, 0, 0], K0[0, 0, 2], K0[0, 1, 2], extra0[0, 0]], camera_id=0)
    rec.add_camera(cam)
    for f in range(6):
        im = pc.Image(id=f, name=f"image_{f}", camera_id=0,
                      cam_from_world=pc.Rigid3d(pc.Rotation3d(ext0[f, :, :3]), ext0[f, :, 3]))
        p2 = []
        for pid in range(1, len(valid) + 1):
            if sc.mask[f, valid[pid - 1]]:
                p2.append(pc.Point2D(sc.tracks[f, valid[pid - 1]], pid))
                rec.points3D[pid].track.add_element(f, len(p2) - 1)
        im.points2D = pc.ListPoint2D(p2)
        im.registered = True
        rec.add_image(im)
    opts = pc.BundleAdjustmentOptions()
    opts.solver_options.max_num_iterations = 20
    pc.bundle_adjustment(rec, opts)
    p_ref, e_ref, K_ref, x_ref, summ = OB.bundle_adjustment(pts0, ext0, K0, sc.tracks, sc.mask, extra0, True,
                                                            "SIMPLE_RADIAL", options=OB.ceres_options(20))
    e = np.stack([rec.images[f].cam_from_world.matrix() for f in range(6)])
    # same CSR problem on both sides; the OpenMP reductions of the oracle are not run-to-run bit-stable
    assert np.abs(e - e_ref).max() < 1e-6
    assert abs(cam.params[0] / K_ref[0, 0, 0] - 1) < 1e-6 and abs(cam.params[3] - x_ref[0, 0]) < 1e-6
    for k, pid in enumerate(range(1, len(valid) + 1)):
        if not summ["deleted"][k]:
            assert np.abs(rec.points3D[pid].xyz - p_ref[k]).max() < 1e-6


_REPRO_SNIPPET = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, {root!r})
from oracle import ba as OB
from vggsfm_amd.scene import make_scene, perturb_for_ba
sc = make_scene(24, 3000, "SIMPLE_RADIAL", shared_camera=True, seed=5)
ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=5)
pts, ext, K, extra, summ = OB.bundle_adjustment(pts0, ext0, K0, sc.tracks, sc.mask, extra0, True, "SIMPLE_RADIAL",
                                                OB.prepare_ba_options())
h = hashlib.sha256()
for it in summ["iterations"]:
    h.update(np.array([it[k] for k in ("cost", "cost_change", "gradient_max_norm", "step_norm",
                                      "relative_decrease", "radius")], np.float64).tobytes())
for a in (pts, ext, K, extra):
    h.update(np.ascontiguousarray(a).tobytes())
print(json.dumps(dict(digest=h.hexdigest(), its=summ["num_iterations"], threads=OB.lib().bao_num_threads())))
"""


def test_oracle_is_bit_reproducible():
    """the checker's own trajectory must not depend on the thread count or on scheduling.
    The same solve in fresh processes with 1, 3 and all threads (twice): iteration log and result bit for bit."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    OB.build()
    outs = []
    for threads in ("1", "3", None, None):
        env = dict(os.environ)
        env.pop("OMP_NUM_THREADS", None)
        if threads:
            env["OMP_NUM_THREADS"] = threads
        r = subprocess.run([sys.executable, "-c", _REPRO_SNIPPET.format(root=root)], env=env, capture_output=True,
                           text=True, check=True)
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0]["threads"] == 1 and outs[1]["threads"] == 3
    assert len({o["digest"] for o in outs}) == 1, outs
    assert outs[0]["its"] > 3
