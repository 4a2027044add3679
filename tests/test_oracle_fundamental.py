"""CPU: the two-view oracle (oracle/fundamental.py; PARITY UNPINNED, see its header) against what can be known without
the reference: its small solvers against numpy, and the whole LO-RANSAC on a synthetic image pair."""
import numpy as np

from oracle import fundamental as Fd


def _rot(w):
    th = np.linalg.norm(w)
    k = w / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def two_view(N, seed, outliers=0.3, noise=0.3):
    rng = np.random.default_rng(seed)
    R, t = _rot(np.array([0.05, 0.2, -0.03])), np.array([0.5, 0.05, 0.1])
    X = np.stack([rng.uniform(-1.5, 1.5, N), rng.uniform(-1.5, 1.5, N), rng.uniform(3, 7, N)], 1)
    proj = lambda R_, t_: (lambda p: p[:, :2] / p[:, 2:] * 1000 + 512)(X @ R_.T + t_)
    x1 = proj(np.eye(3), np.zeros(3)) + rng.normal(0, noise, (N, 2))
    x2 = proj(R, t) + rng.normal(0, noise, (N, 2))
    out = rng.random(N) < outliers
    x2[out] += rng.uniform(-80, 80, (int(out.sum()), 2))
    Kc = np.array([[1000, 0, 512], [0, 1000, 512], [0, 0, 1.0]])
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ft = np.linalg.inv(Kc).T @ (tx @ R) @ np.linalg.inv(Kc)
    return x1, x2, out, Ft / np.linalg.norm(Ft)


def test_cubic_roots_match_numpy():
    rng = np.random.default_rng(0)
    c = rng.normal(size=(3000, 4))
    c[:300, 0] *= 1e-6                       # huge root
    c[300:600, 3] *= 1e-6                    # tiny root
    c[600:650, 0] = 0.0                      # quadratic
    r, v = Fd.cubic_real_roots(c[:, 0], c[:, 1], c[:, 2], c[:, 3])
    pv = ((c[:, 0:1] * r + c[:, 1:2]) * r + c[:, 2:3]) * r + c[:, 3:4]
    scale = np.abs(c[:, 0:1] * r ** 3) + np.abs(c[:, 1:2] * r ** 2) + np.abs(c[:, 2:3] * r) + np.abs(c[:, 3:4])
    assert (np.abs(pv)[v] <= 1e-12 * scale[v]).all()
    for i in list(range(0, 3000, 37)):
        roots = np.roots(c[i][np.argmax(c[i] != 0):])
        assert int((np.abs(roots.imag) < 1e-9).sum()) == int(v[i].sum())


def test_null_space_and_jacobi():
    rng = np.random.default_rng(1)
    A = rng.normal(size=(200, 7, 9))
    f1, f2, ok = Fd.null_space_7x9(A)
    assert ok.all()
    assert np.abs(np.einsum("mij,mj->mi", A, f1)).max() < 1e-12 and np.abs(np.einsum("mij,mj->mi", A, f2)).max() < 1e-12
    assert (np.abs(np.einsum("mi,mi->m", f1, f2)) < 1.0 + 1e9).all() and np.linalg.matrix_rank(np.stack([f1[0], f2[0]])) == 2
    S = rng.normal(size=(50, 9, 12))
    S = S @ np.swapaxes(S, 1, 2)
    d, V = Fd.jacobi_eigh(S, Fd.SWEEPS9)
    np.testing.assert_allclose(V @ (d[:, :, None] * np.swapaxes(V, 1, 2)), S, atol=1e-11)
    np.testing.assert_allclose(np.sort(d, 1), np.linalg.eigvalsh(S), rtol=1e-10, atol=1e-10)


def test_seven_point_contains_true_matrix_on_exact_data():
    x1, x2, _, Ft = two_view(400, 2, outliers=0.0, noise=0.0)
    rng = np.random.default_rng(3)
    smp = np.stack([rng.choice(400, 7, replace=False) for _ in range(200)])
    F, v = Fd.seven_point(x1[smp], x2[smp])
    assert v.any(1).all()
    err = np.minimum(np.abs(F - Ft).max((-1, -2)), np.abs(F + Ft).max((-1, -2)))
    err[~v] = np.inf
    assert (err.min(1) < 1e-7).mean() > 0.97
    # every reported matrix is singular and satisfies its 7 epipolar constraints
    assert np.abs(np.linalg.det(F[v])).max() < 1e-9
    h1 = np.concatenate([x1[smp], np.ones((200, 7, 1))], -1)
    h2 = np.concatenate([x2[smp], np.ones((200, 7, 1))], -1)
    epi = np.einsum("hni,hsij,hnj->hsn", h2, F, h1)
    assert np.abs(epi[v]).max() < 1e-6


def test_lo_ransac_on_synthetic_pair():
    x1, x2, out, Ft = two_view(1500, 0)
    vm = np.ones(1500, bool)
    vm[::9] = False
    rng = np.random.default_rng(5)
    smp = np.stack([rng.choice(1500, 7, replace=False) for _ in range(256)])
    o = Fd.estimate_fundamental_pair(x1, x2, vm, smp, 1.0, lo_num=40)
    good = ~out & vm
    assert not o["inlier_mask"][~vm].any()
    assert (o["inlier_mask"] & good).sum() >= 0.985 * good.sum()
    assert (o["inlier_mask"] & ~good).sum() <= 0.08 * (~good & vm).sum()          # outliers that happen to lie on the line
    assert min(np.abs(o["fmat_unit"] - Ft).max(), np.abs(o["fmat_unit"] + Ft).max()) < 3e-3
    assert np.linalg.svd(o["fmat_unit"])[1][2] < 1e-12 and abs(o["fmat"][2, 2] - 1) < 1e-12
    assert o["best"] >= 3 * 256                                                     # a locally optimised matrix wins
