"""GPU: P3P RANSAC (vgg_p3p_ransac through the C-ABI) vs oracle/p3p.py on the SAME samples -- the kernel and the oracle
mirror each other operation by operation (no FMA contraction), so counts, winners and inlier masks must be identical
and poses equal to rounding -- and the callers built on it (absolute_pose_estimation_batch, refine_pose's
force_estimate fallback, align_next_window(use_pnp)).  Parity with pycolmap itself is unpinned (oracle/p3p.py header)."""
import numpy as np
import pytest
import torch

from oracle import p3p as P
from vggsfm_amd import video as V
from vggsfm_amd.ba_options import AbsolutePoseEstimationOptions, AbsolutePoseRefinementOptions
from vggsfm_amd.pose import absolute_pose_estimation_batch, draw_minimal_samples, p3p_ransac
from vggsfm_amd.scene import make_scene, perturb_for_ba
from vggsfm_amd.utils import triangulation as T

pytestmark = pytest.mark.gpu


def D(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _normalized(sc, K=None):
    K = sc.intrinsics if K is None else K
    f, cx, cy = K[:, 0, 0], K[:, 0, 2], K[:, 1, 2]
    x = sc.tracks.astype(np.float64).copy()
    x[..., 0] = (x[..., 0] - cx[:, None]) / f[:, None]
    x[..., 1] = (x[..., 1] - cy[:, None]) / f[:, None]
    return x


@pytest.mark.parametrize("S,N,H,group", [(5, 700, 256, 1), (6, 300, 64, 3), (2, 64, 1000, 1)])
def test_p3p_ransac_matches_oracle_on_same_samples(S, N, H, group):
    sc = make_scene(S, N, "SIMPLE_PINHOLE", seed=40 + S, full_visibility=True, outlier_frac=0.3, noise_px=0.5)
    x = _normalized(sc)
    if group > 1:                                   # virtual frames: (frame, focal factor), sharing the frame's samples
        fac = np.array([0.8, 1.0, 1.3])[:group]
        x = (x[: S // group, None] / fac[None, :, None, None]).reshape(-1, N, 2)
    F = x.shape[0]
    rng = np.random.default_rng(S)
    mask = rng.random((F, N)) < 0.9
    thr = np.full(F, (2.0 / 1000.0) ** 2) * rng.uniform(0.5, 2.0, size=F)
    g = torch.Generator(device="cuda").manual_seed(7)
    samples = draw_minimal_samples(D(mask)[::group].contiguous(), H, g)
    pose, num, rsum, best, inl = p3p_ransac(D(x), D(sc.points3D), D(mask), samples, D(thr), group)
    smp = samples.cpu().numpy()
    for f in range(F):
        o = P.absolute_pose_ransac(x[f], sc.points3D, mask[f], smp[f // group], thr[f])
        assert int(num[f]) == o["num_inliers"] and int(best[f]) == o["best"], (f, int(num[f]), o["num_inliers"])
        np.testing.assert_array_equal(inl[f].cpu().numpy(), o["inliers"])
        np.testing.assert_allclose(pose[f].cpu().numpy(), o["pose"], rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(float(rsum[f]), o["residual_sum"], rtol=1e-10)
    # the winner explains the clean matches of the true pose (30 % outliers)
    if group == 1:
        assert int(num.min()) > 0.5 * N


def test_p3p_ransac_degenerate_inputs():
    # all samples identical / collinear points / no candidates: no pose, zero inliers, empty mask -- and no NaNs
    N = 50
    X = np.zeros((N, 3))
    X[:, 0] = np.linspace(-1, 1, N)                 # collinear world points
    X[:, 2] = 4.0
    x = X[None, :, :2] / X[None, :, 2:]
    smp = torch.zeros((1, 16, 3), dtype=torch.int32, device="cuda")
    smp[0, :, 1], smp[0, :, 2] = 1, 2
    pose, num, rsum, best, inl = p3p_ransac(D(x), D(X), None, smp, D(np.array([1e-4])))
    assert int(num[0]) == 0 and int(best[0]) == -1 and not bool(inl.any()) and bool(torch.isfinite(pose).all())
    with pytest.raises(RuntimeError):
        p3p_ransac(D(x[:, :2]), D(X[:2]), None, smp, D(np.array([1e-4])))      # fewer than 3 points


@pytest.mark.parametrize("cam,estimate_focal", [("SIMPLE_PINHOLE", False), ("SIMPLE_RADIAL", True)])
def test_absolute_pose_estimation_recovers_lost_frames(cam, estimate_focal):
    S, N = 6, 1500
    sc = make_scene(S, N, cam, shared_camera=False, seed=51, full_visibility=True, outlier_frac=0.25)
    params = np.zeros((S, 4))
    params[:, 0], params[:, 1], params[:, 2] = sc.intrinsics[:, 0, 0], sc.intrinsics[:, 0, 2], sc.intrinsics[:, 1, 2]
    if sc.extra_params is not None:
        params[:, 3] = sc.extra_params[:, 0]
    if estimate_focal:
        params[:, 0] *= 1.6                          # a wrong prior focal: the factor search has to find ~1/1.6
    ext0 = sc.extrinsics.copy()
    ext0[1:, :, :3] = np.eye(3)                      # lost frames: identity pose
    ext0[1:, :, 3] = 0.0
    est = AbsolutePoseEstimationOptions(estimate_focal_length=estimate_focal)
    est.ransac.max_error = 4.0
    ref = AbsolutePoseRefinementOptions(refine_focal_length=estimate_focal, refine_extra_params=False)
    flags = torch.full((S,), 1 if estimate_focal else 0, dtype=torch.uint8)
    cand = sc.mask.copy()
    cand[:, ::11] = False
    g = torch.Generator(device="cuda").manual_seed(1)
    ext, prm, ok, num, inl = absolute_pose_estimation_batch(D(ext0), D(params), D(sc.tracks), D(sc.points3D), D(cand),
                                                            list(range(1, S)), cam, flags, est, ref, generator=g)
    assert not bool(ok[0]) and bool(ok[1:].all())
    np.testing.assert_array_equal(ext[0].cpu().numpy(), ext0[0])
    assert not bool((inl & ~D(cand)).any())
    clean = D(cand & ~sc.outlier)
    assert float((inl[1:] & clean[1:]).sum()) > 0.9 * float(clean[1:].sum())
    e = ext.cpu().numpy()
    assert np.abs(e[1:, :, :3] - sc.extrinsics[1:, :, :3]).max() < 2e-2
    assert np.abs(e[1:, :, 3] - sc.extrinsics[1:, :, 3]).max() < 8e-2
    if estimate_focal:
        np.testing.assert_allclose(prm[1:, 0].cpu().numpy(), sc.intrinsics[1:, 0, 0], rtol=2e-2)


def test_refine_pose_force_estimate_fallback():
    # frame 3 starts so far off that (almost) nothing reprojects within 12 px: without force_estimate it keeps its
    # pose (reference: "only has ... geo_vis inliers"), with it the P3P fallback + refinement recovers it
    S, N = 8, 2500
    sc = make_scene(S, N, "SIMPLE_PINHOLE", shared_camera=False, seed=61, full_visibility=True, outlier_frac=0.05)
    ext0, K0, _, _ = perturb_for_ba(sc, seed=61, rot_deg=0.2, trans=0.01, focal_rel=0.005)
    ext0[3, :, :3] = np.eye(3)
    ext0[3, :, 3] = [0.5, -0.5, 1.0]
    size = torch.tensor([1024.0, 1024.0], device="cuda")
    args = (D(ext0), D(K0), None, D(sc.mask), D(sc.points3D), D(sc.tracks), torch.ones(N, dtype=torch.bool, device="cuda"), size)
    e1, K1, _, v1 = T.refine_pose(*args, camera_type="SIMPLE_PINHOLE", force_estimate=False)
    np.testing.assert_allclose(e1[3].cpu().numpy(), ext0[3], atol=1e-12)
    torch.manual_seed(5)
    e2, K2, _, v2 = T.refine_pose(*args, camera_type="SIMPLE_PINHOLE", force_estimate=True)
    assert bool(v2.all())
    assert np.abs(e2[3].cpu().numpy() - sc.extrinsics[3]).max() < 3e-2
    np.testing.assert_allclose(float(K2[3, 0, 0]), 1000.0, rtol=5e-2)     # (focal / depth trade off over a shallow scene)
    # the other frames are untouched by the fallback
    keep = [i for i in range(S) if i != 3]
    np.testing.assert_array_equal(e2[keep].cpu().numpy(), e1[keep].cpu().numpy())


def test_align_next_window_use_pnp():
    S, N = 7, 1200
    sc = make_scene(S, N, "SIMPLE_RADIAL", shared_camera=True, seed=71, full_visibility=True, outlier_frac=0.1)
    ext0 = sc.extrinsics.copy()
    ext0[1:, :, :3] = np.eye(3)                      # no usable prior for the new frames
    ext0[1:, :, 3] = [0.0, 0.0, 1.0]
    inl = sc.mask & ~sc.outlier
    g = torch.Generator(device="cuda").manual_seed(2)
    ext = V.align_next_window(D(ext0), D(sc.tracks), D(inl), D(sc.points3D), D(sc.intrinsics[:1]), D(sc.extra_params[:1]),
                              "SIMPLE_RADIAL", use_pnp=True, generator=g)
    np.testing.assert_array_equal(ext[0].cpu().numpy(), ext0[0])
    assert np.abs(ext.cpu().numpy() - sc.extrinsics).max() < 1e-2
