"""GPU: the two-view kernels (vgg_fmat_* through the C-ABI) against oracle/fundamental.py on the SAME samples -- the
two mirror each other operation by operation (float64, no FMA contraction, fixed-order sums), so hypotheses, counts,
winners and inlier masks must be identical -- plus the behaviour of ``estimate_preliminary_cameras`` on a synthetic
sequence.  Parity with the reference's float32 / kornia implementation is unpinned (oracle/fundamental.py header)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import fundamental as Fd
from tests.test_oracle_fundamental import two_view
from vggsfm_amd import _lib
from vggsfm_amd.scene import make_scene
from vggsfm_amd.two_view_geo import estimate_fundamental, estimate_preliminary_cameras

pytestmark = pytest.mark.gpu


def D(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _pairs(B, N, seed):
    xs = [two_view(N, seed + b, outliers=0.2 + 0.05 * b) for b in range(B)]
    x1, x2 = np.stack([x[0] for x in xs]), np.stack([x[1] for x in xs])
    rng = np.random.default_rng(seed)
    vm = rng.random((B, N)) < 0.92
    return x1, x2, vm, xs


def test_seven_point_and_score_match_oracle_bitwise():
    B, N, H = 3, 900, 200
    x1, x2, vm, _ = _pairs(B, N, 10)
    rng = np.random.default_rng(1)
    smp = np.stack([rng.choice(N, 7, replace=False) for _ in range(H)]).astype(np.int32)
    L = _lib.lib()
    d1, d2, ds, dm = D(x1), D(x2), D(smp), D(vm.astype(np.uint8))        # (kept alive across the raw-pointer calls)
    F7 = torch.empty((B, H, 3, 9), dtype=torch.float64, device="cuda")
    v7 = torch.empty((B, H, 3), dtype=torch.uint8, device="cuda")
    _lib.check(L.vgg_fmat_seven_point(_lib.ptr(d1), _lib.ptr(d2), _lib.ptr(ds), B, N, H, _lib.ptr(F7), _lib.ptr(v7),
                                      _lib.stream_ptr()), "seven")
    cnt = torch.empty((B, 3 * H), dtype=torch.int32, device="cuda")
    rs = torch.empty((B, 3 * H), dtype=torch.float64, device="cuda")
    _lib.check(L.vgg_fmat_score(_lib.ptr(d1), _lib.ptr(d2), _lib.ptr(dm), _lib.ptr(F7), _lib.ptr(v7), B, N,
                                3 * H, ctypes.c_double(1.0), _lib.ptr(cnt), _lib.ptr(rs), _lib.stream_ptr()), "score")
    for b in range(B):
        Fo, vo = Fd.seven_point(x1[b][smp], x2[b][smp])
        np.testing.assert_array_equal(v7[b].cpu().numpy().astype(bool), vo)
        np.testing.assert_array_equal(F7[b].cpu().numpy().reshape(H, 3, 3, 3), Fo)          # bit for bit
        co, ro, _ = Fd.score(Fo.reshape(-1, 3, 3), vo.reshape(-1), x1[b], x2[b], vm[b], 1.0)
        np.testing.assert_array_equal(cnt[b].cpu().numpy(), co)
        np.testing.assert_array_equal(rs[b].cpu().numpy(), ro)


@pytest.mark.parametrize("B,N,H,lo", [(2, 700, 128, 24), (1, 3000, 300, 50)])
def test_estimate_fundamental_matches_oracle(B, N, H, lo):
    x1, x2, vm, xs = _pairs(B, N, 20 + N)
    rng = np.random.default_rng(2)
    smp = np.stack([rng.choice(N, 7, replace=False) for _ in range(H)]).astype(np.int32)
    F, num, mask, res = estimate_fundamental(D(x1), D(x2), max_error=1.0, lo_num=lo, valid_mask=D(vm), return_residuals=True,
                                             samples=smp)
    for b in range(B):
        o = Fd.estimate_fundamental_pair(x1[b], x2[b], vm[b], smp, 1.0, lo_num=lo)
        assert int(num[b]) == o["inlier_num"]
        np.testing.assert_array_equal(mask[b].cpu().numpy(), o["inlier_mask"])
        np.testing.assert_array_equal(F[b].cpu().numpy(), o["fmat"])
        np.testing.assert_array_equal(res[b].cpu().numpy(), o["residuals"])
        good = ~xs[b][2] & vm[b]
        assert (o["inlier_mask"] & good).sum() >= 0.98 * good.sum()


def test_estimate_preliminary_cameras_on_a_scene():
    # the query frame against 7 others: clean matches are inliers, gross outliers are not, invisible matches never
    S, N = 8, 2500
    sc = make_scene(S, N, "SIMPLE_PINHOLE", seed=33, full_visibility=True, outlier_frac=0.15, noise_px=0.4)
    vis = sc.vis.copy()
    vis[:, ::17] = 0.0
    np.random.seed(0)
    cams, pre = estimate_preliminary_cameras(D(sc.tracks)[None], D(vis)[None], 1024, 1024, tracks_score=D(sc.score)[None],
                                             max_error=2.0, max_ransac_iters=512, lo_num=60)
    m = pre["fmat_inlier_mask"][0].cpu().numpy()
    # preliminary cameras (estimate_preliminary.py:153-241): E = K^T F K with the DEFAULT intrinsics (focal = 1024 where the
    # scene has ~1000), decomposed, cheirality resolved: rotation close to the true relative rotation, translation
    # direction within a few degrees, camera 0 the identity; pred_cameras = the same in the PyTorch3D convention
    Ro, to = pre["R_opencv"][0].cpu().numpy(), pre["t_opencv"][0].cpu().numpy()
    assert Ro.shape == (S, 3, 3) and np.array_equal(Ro[0], np.eye(3)) and not to[0].any()
    E0 = sc.extrinsics
    for s_ in range(1, S):
        Rrel = E0[s_, :, :3] @ E0[0, :, :3].T
        trel = E0[s_, :, 3] - Rrel @ E0[0, :, 3]
        ang = np.degrees(np.arccos(np.clip((np.trace(Ro[s_] @ Rrel.T) - 1) / 2, -1, 1)))
        cosd = float(to[s_] @ trel / (np.linalg.norm(to[s_]) * np.linalg.norm(trel)))
        assert ang < 3.0 and cosd > 0.98, (s_, ang, cosd)
    Rp, Tp = cams.R.cpu().numpy(), cams.T.cpu().numpy()
    flipxy = np.array([-1.0, -1.0, 1.0])
    assert np.allclose(Rp, np.transpose(Ro, (0, 2, 1)) * flipxy[None, None, :]) and np.allclose(Tp, to * flipxy[None])
    assert pre["default_intri"].shape == (1, S - 1, 3, 3) and float(pre["default_intri"][0, 0, 0, 0]) == 1024.0
    assert pre["emat_fromf"].shape == (S - 1, 3, 3)
    assert m.shape == (S - 1, N) and pre["fmat"].shape == (1, S - 1, 3, 3) and pre["fmat_residuals"].shape == (1, S - 1, N)
    assert not m[:, ::17].any()
    clean = ~(sc.outlier[1:] | sc.outlier[0:1]) & (vis[1:] > 0)
    assert (m & clean).sum() >= 0.97 * clean.sum()
    gross = (sc.outlier[1:] | sc.outlier[0:1]) & (vis[1:] > 0)
    assert (m & gross).sum() <= 0.15 * gross.sum()
    # x2^T F x1 = 0 on the inliers, F scaled to F[2,2] = 1, rank 2
    F = pre["fmat"][0].cpu().numpy()
    assert np.allclose(F[:, 2, 2], 1.0) and np.abs(np.linalg.det(F / np.linalg.norm(F, axis=(1, 2), keepdims=True))).max() < 1e-10
    with pytest.raises(ValueError):
        estimate_fundamental(D(sc.tracks[0:1, :5].astype(np.float64)), D(sc.tracks[1:2, :5].astype(np.float64)))


def test_two_view_stage_feeds_the_triangulator():
    """Everything after the tracker on the device: estimate_preliminary_cameras -> Triangulator.forward
    (vggsfm/runners/runner.py:467-515) on a synthetic scene with a noisy camera prior."""
    import types

    from vggsfm_amd.models import Triangulator
    from vggsfm_amd.scene import perturb_for_ba, project
    S, N, W = 10, 3000, 1024
    sc = make_scene(S, N, "SIMPLE_PINHOLE", shared_camera=False, seed=77, full_visibility=True, outlier_frac=0.08)
    ext0, K0, _, _ = perturb_for_ba(sc, seed=77, rot_deg=0.5, trans=0.02, focal_rel=0.02)
    tracks, vis, score = D(sc.tracks)[None], D(sc.vis)[None], D(sc.score)[None]
    np.random.seed(1)
    _, prelim = estimate_preliminary_cameras(tracks, vis, W, W, tracks_score=score, max_error=4.0, max_ransac_iters=1024,
                                             lo_num=100)
    cams = types.SimpleNamespace(R=torch.from_numpy(ext0[:, :, :3]).float().cuda(), T=torch.from_numpy(ext0[:, :, 3]).float().cuda())
    f = torch.from_numpy(K0[:, 0, 0] / (W / 2.0)).float().cuda()
    cams.focal_length = torch.stack([f, f], -1)
    torch.manual_seed(0)
    out = Triangulator()(cams, tracks, vis, torch.rand(1, S, 3, W, W, device="cuda"), prelim, pred_score=score,
                         shared_camera=False, camera_type="SIMPLE_PINHOLE", BA_iters=2, robust_refine=2)
    ext, K, extra, pts, rgb, rec, valid_frames, valid_2D, valid_tracks = out
    assert bool(valid_frames.all()) and float(valid_tracks.float().mean()) > 0.85
    uv, _ = project(pts.cpu().numpy(), ext.cpu().numpy(), K.cpu().numpy(), None)
    m = valid_2D[:, valid_tracks].cpu().numpy()
    err = np.linalg.norm(uv - sc.tracks[:, valid_tracks.cpu().numpy()], axis=-1)[m]
    assert np.median(err) < 1.0 and np.percentile(err, 95) < 2.5
    # the outliers the two-view stage rejected never vote in the initial pair selection; the final model rejects them too
    bad = sc.outlier[:, valid_tracks.cpu().numpy()]
    assert (m & bad).sum() < 0.1 * bad.sum()


@pytest.mark.gpu
def test_sampson_distance_and_inlier_mask_match_oracle():
    """two_view_geo.utils.sampson_epipolar_distance_batched / inlier_by_fundamental (vgg_fmat_residuals) against the numpy
    restatement of the reference's formula (oracle/fundamental.py: sampson_sq), float64: 1e-12 relative."""
    from oracle import fundamental as OF
    from vggsfm_amd.two_view_geo.utils import inlier_by_fundamental, sampson_epipolar_distance_batched
    rng = np.random.default_rng(11)
    B, K, N = 3, 4, 500
    p1 = rng.uniform(0, 1000, size=(B, N, 2))
    p2 = p1 + rng.normal(0, 5, size=(B, N, 2))
    Fm = rng.normal(size=(B, K, 3, 3))
    dev = torch.device("cuda:0")
    T = lambda x: torch.from_numpy(x).to(dev)
    got = sampson_epipolar_distance_batched(T(p1), T(p2), T(Fm)).cpu().numpy()
    for b in range(B):
        np.testing.assert_allclose(got[b], OF.sampson_sq(Fm[b], p1[b], p2[b]), rtol=1e-11, atol=1e-300)
    unsq = sampson_epipolar_distance_batched(T(p1), T(p2), T(Fm), squared=False).cpu().numpy()
    np.testing.assert_allclose(unsq, np.sqrt(got + 1e-8), rtol=1e-14)
    # inlier_by_fundamental: tracks (1,S,N,2), one F per pair (frame 0, frame s)
    S = B + 1
    tracks = np.concatenate([p1[0:1], p2[0:1], p1[1:2] * 0 + p1[0:1] + rng.normal(0, 3, size=(1, N, 2)),
                             p1[0:1] + rng.normal(0, 3, size=(1, N, 2))])[None]
    F = rng.normal(size=(1, S - 1, 3, 3))
    mask = inlier_by_fundamental(T(F), T(tracks), max_error=40.0).cpu().numpy()
    for s_ in range(1, S):
        ref = OF.sampson_sq(F[0, s_ - 1][None], tracks[0, 0], tracks[0, s_])[0] <= 40.0 ** 2
        assert np.array_equal(mask[0, s_ - 1], ref)


@pytest.mark.gpu
def test_cheirality_helpers_on_a_known_pose():
    """two_view_geo.utils.check_cheirality_batch / triangulate_point_batch (pair-triangulation kernel): with the true
    relative pose every match triangulates in front of both cameras and lands on the ground-truth point; with the
    pose mirrored (t -> -t) none does."""
    from scipy.spatial.transform import Rotation
    from vggsfm_amd.two_view_geo.utils import check_cheirality_batch
    rng = np.random.default_rng(4)
    N = 300
    X = np.array([0.0, 0.0, 5.0]) + rng.uniform(-1, 1, size=(N, 3))
    R = Rotation.from_rotvec([0.02, -0.1, 0.03]).as_matrix()
    t = np.array([0.5, 0.05, 0.1])
    x1 = X[:, :2] / X[:, 2:3]
    Y = X @ R.T + t
    x2 = Y[:, :2] / Y[:, 2:3]
    dev = torch.device("cuda:0")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    nums, pts = check_cheirality_batch(T(np.stack([R, R])), T(np.stack([t, -t])), T(np.stack([x1, x1])), T(np.stack([x2, x2])))
    assert nums.tolist() == [N, 0]
    np.testing.assert_allclose(pts[0].cpu().numpy(), X, rtol=0, atol=1e-9)
