"""CPU: the P3P RANSAC oracle (oracle/p3p.py; PARITY UNPINNED -- pycolmap is absent, see its header) checked against
what can be known without the reference: exact recovery on noise-free triplets, polynomial roots, RANSAC behaviour on a
synthetic scene, COLMAP's focal length factors; plus the host-side sample drawing of the product path."""
import numpy as np
import torch

from oracle import p3p as P


def _rand_rot(rng):
    q = rng.normal(size=4)
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def test_quartic_real_roots():
    rng = np.random.default_rng(0)
    roots = np.sort(rng.uniform(-3, 3, size=(500, 4)), axis=1)
    roots = roots[np.diff(roots, axis=1).min(1) > 0.05]             # well separated
    lead = rng.uniform(0.5, 2.0, size=len(roots))
    coef = np.stack([np.poly(r) * a for r, a in zip(roots, lead)])  # k4..k0
    v, ok = P.solve_quartic(*(coef[:, i] for i in range(5)))
    assert ok.all()
    np.testing.assert_allclose(np.sort(v, axis=1), roots, atol=1e-9)
    # two real + two complex roots: exactly two reported
    coef = np.stack([np.polymul(np.poly([a, b]), [1.0, 0.0, c]) for a, b, c in rng.uniform(0.5, 2.0, size=(200, 3))])
    v, ok = P.solve_quartic(*(coef[:, i] for i in range(5)))
    assert (ok.sum(1) == 2).all()


def test_p3p_exact_recovery_and_valid_rotations():
    rng = np.random.default_rng(1)
    H = 3000
    Rs = np.stack([_rand_rot(rng) for _ in range(H)])
    ts = rng.normal(size=(H, 3))
    Y = np.concatenate([rng.uniform(-1, 1, size=(H, 3, 2)), rng.uniform(2, 6, size=(H, 3, 1))], -1)
    Y[..., :2] *= Y[..., 2:]
    X = np.einsum("hji,hnj->hni", Rs, Y - ts[:, None])
    x = Y[..., :2] / Y[..., 2:]
    poses, ok = P.p3p_solve(x, X)
    gt = np.concatenate([Rs, ts[..., None]], -1)
    err = np.abs(poses - gt[:, None]).max((-1, -2))
    err[~ok] = np.inf
    assert (err.min(1) < 1e-6).mean() > 0.99                       # the true pose is among the <= 4 solutions
    R = poses[ok][:, :, :3]
    np.testing.assert_allclose(R @ np.swapaxes(R, -1, -2), np.broadcast_to(np.eye(3), R.shape), atol=1e-9)
    np.testing.assert_allclose(np.linalg.det(R), 1.0, atol=1e-9)
    # every reported solution reproduces its three image points
    p = np.einsum("hsij,hnj->hsni", poses[..., :3], X) + poses[..., None, :, 3]
    with np.errstate(all="ignore"):
        rep = np.abs(p[..., :2] / p[..., 2:] - x[:, None]).max((-1, -2))
    assert np.median(rep[ok]) < 1e-12 and (rep[ok] < 1e-5).all()


def test_ransac_scene_with_outliers():
    rng = np.random.default_rng(2)
    N = 600
    Rg, tg = _rand_rot(rng), np.array([0.1, -0.2, 4.0])
    X = rng.uniform(-1, 1, size=(N, 3))
    p = X @ Rg.T + tg
    x = p[:, :2] / p[:, 2:] + rng.normal(size=(N, 2)) * 5e-4
    out = rng.random(N) < 0.35
    x[out] += rng.uniform(-0.5, 0.5, size=(int(out.sum()), 2))
    mask = np.ones(N, bool)
    mask[::13] = False
    samples = np.stack([rng.choice(np.nonzero(mask)[0], 3, replace=False) for _ in range(300)])
    r = P.absolute_pose_ransac(x, X, mask, samples, (3e-3) ** 2)
    good = ~out & mask
    assert r["num_inliers"] >= 0.97 * good.sum() and not r["inliers"][~mask].any()
    assert (r["inliers"] & good).sum() >= 0.97 * good.sum()
    assert np.abs(r["pose"][:, :3] - Rg).max() < 5e-3 and np.abs(r["pose"][:, 3] - tg).max() < 2e-2
    # nothing usable -> no result
    r0 = P.absolute_pose_ransac(x, X, mask, samples[:4] * 0, 1e-6)
    assert r0["num_inliers"] == 0 and r0["best"] == -1


def test_focal_length_factors_follow_colmap():
    f = P.focal_length_factors(True)
    assert len(f) == 30 and f[0] == 0.2 and abs(f[15] - (0.2 + 4.8 * 0.25)) < 1e-12 and f[-1] < 5.0
    assert list(P.focal_length_factors(False)) == [1.0]
    from vggsfm_amd.ba_options import AbsolutePoseEstimationOptions
    from vggsfm_amd.pose import focal_length_factors
    np.testing.assert_allclose(focal_length_factors(AbsolutePoseEstimationOptions(estimate_focal_length=True)), f, rtol=1e-15)


def test_draw_minimal_samples_distinct_candidates():
    from vggsfm_amd.pose import draw_minimal_samples
    g = torch.Generator().manual_seed(3)
    m = torch.rand(5, 40, generator=g) < 0.3
    m[1] = False
    m[1, [4, 9, 30]] = True                                         # exactly three candidates
    m[2] = False                                                    # none: zeros
    s = draw_minimal_samples(m, 200, g).numpy()
    assert s.shape == (5, 200, 3) and s.dtype == np.int32
    for f in (0, 1, 3, 4):
        assert m[f].numpy()[s[f]].all()
        assert (s[f][:, 0] != s[f][:, 1]).all() and (s[f][:, 0] != s[f][:, 2]).all() and (s[f][:, 1] != s[f][:, 2]).all()
    assert set(np.unique(s[1])) == {4, 9, 30} and (s[2] == 0).all()
    # uniform over the candidates (loose chi-square-free check)
    cnt = np.bincount(s[0].ravel(), minlength=40)[m[0].numpy()]
    assert cnt.min() > 0.4 * cnt.mean() and cnt.max() < 1.8 * cnt.mean()
