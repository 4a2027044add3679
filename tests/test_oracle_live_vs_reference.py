"""CPU, build container only (needs /root/reference; skipped on the GPU box): the numpy oracle (oracle/geometry.py) against
the REFERENCE's own torch functions run LIVE on freshly seeded inputs -- shapes and seeds that no committed golden vector
has seen -- for every function of the torch half of the path: project_3D_points, cam_from_img (undistortion),
filter_all_points3D, triangulate_by_pair and triangulate_tracks (LO-RANSAC; the torch.randperm draws the reference makes
are recorded and replayed).  Same comparisons and tolerances as tests/test_oracle_golden.py; what this adds is that the
oracle is not fitted to the goldens."""
import numpy as np
import pytest
import torch

from oracle import geometry as G
from oracle import ref_harness
from oracle.gen_golden import T, _RecordRandperm, _StableSort, tc
from vggsfm_amd.scene import make_scene, perturb_for_ba

pytestmark = pytest.mark.skipif(not ref_harness.available(), reason="reference tree not present (GPU box)")

CASES = [  # seed, S, N, camera, shared
    (101, 7, 90, "SIMPLE_PINHOLE", False),
    (102, 11, 120, "SIMPLE_RADIAL", True),
    (103, 16, 70, "SIMPLE_RADIAL", False),
]


def _scene(seed, S, N, cam, shared):
    sc = make_scene(S, N, cam, shared_camera=shared, seed=seed, outlier_frac=0.08)
    ext, K, extra, pts = perturb_for_ba(sc, seed=seed, rot_deg=0.2, trans=0.01, focal_rel=0.004)
    return sc, ext, K, extra, pts


@pytest.mark.parametrize("seed,S,N,cam,shared", CASES)
def test_projection_normalisation_and_filter_live(seed, S, N, cam, shared):
    _, helpers, _ = ref_harness.load()
    sc, ext, K, extra, pts = _scene(seed, S, N, cam, shared)
    pts = pts.copy()
    pts[0] = [0.0, 0.0, -2.0]                                   # behind the cameras
    pts[1] = [400.0, 1.0, 4.0]                                  # beyond hard_max
    p2r, pcr = helpers.project_3D_points(T(pts), T(ext), T(K), T(extra), return_points_cam=True)
    p2, pc = G.project_3D_points(pts, ext, K, extra, return_points_cam=True)
    np.testing.assert_allclose(pc, pcr.numpy(), rtol=1e-13, atol=1e-13)
    fin = np.isfinite(p2r.numpy()) & (np.abs(p2r.numpy()) < 1e300)
    np.testing.assert_allclose(p2[fin], p2r.numpy()[fin], rtol=1e-11, atol=1e-9)
    tnr = helpers.cam_from_img(T(sc.tracks), T(K), T(extra)).numpy()
    tn = G.cam_from_img(sc.tracks.astype(np.float64), K, extra)
    np.testing.assert_allclose(tn, tnr, rtol=1e-9, atol=1e-12)
    for chk in (False, True):
        for thr in (4, 1.5):
            mr, dr = helpers.filter_all_points3D(T(pts), T(sc.tracks), T(ext), T(K), T(extra), max_reproj_error=thr,
                                                 check_triangle=chk, return_detail=True)
            m, d = G.filter_all_points3D(pts, sc.tracks.astype(np.float64), ext, K, extra, max_reproj_error=thr,
                                         check_triangle=chk, return_detail=True)
            assert np.array_equal(m, mr.numpy()) and np.array_equal(d, dr.numpy().astype(bool)), (chk, thr)


@pytest.mark.parametrize("seed,S,N,cam,shared", CASES)
def test_triangulate_by_pair_live(seed, S, N, cam, shared):
    tri, helpers, _ = ref_harness.load()
    sc, ext, K, extra, _ = _scene(seed, S, N, cam, shared)
    tn = helpers.cam_from_img(T(sc.tracks), T(K), T(extra))
    pr, cher, angr = tri.triangulate_by_pair(T(ext)[None], tn[None])
    p, che, ang = G.triangulate_by_pair(ext, tn.numpy())
    assert np.array_equal(che, cher.numpy())
    ok = np.isfinite(pr.numpy()).all(-1)
    np.testing.assert_allclose(p[ok], pr.numpy()[ok], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(ang[ok], angr.numpy()[ok], rtol=1e-7, atol=1e-9)


@pytest.mark.parametrize("seed,S,N,cam,iters", [(111, 9, 80, "SIMPLE_PINHOLE", 256), (112, 26, 60, "SIMPLE_RADIAL", 256),
                                               (113, 25, 50, "SIMPLE_PINHOLE", 128)])
def test_triangulate_tracks_live(seed, S, N, cam, iters):
    tri, helpers, _ = ref_harness.load()
    sc, ext, K, extra, _ = _scene(seed, S, N, cam, cam == "SIMPLE_RADIAL")
    vis, score = sc.vis.copy(), sc.score.copy()
    score[:, ::11] = 0.45                                        # whole tracks below the score threshold
    vis[1, ::7] = 0.04                                           # entries below the visibility threshold
    tn = helpers.cam_from_img(T(sc.tracks), T(K), T(extra))
    torch.manual_seed(seed)
    with _StableSort(), _RecordRandperm() as rec:
        pr, numr, mskr = tri.triangulate_tracks(T(ext), tc(tn), max_ransac_iters=iters, track_vis=tc(T(vis)),
                                                track_score=tc(T(score)))
    comb = G.generate_combinations(S)
    if iters > len(comb):
        assert not rec.draws
        pairs = comb
    else:
        assert len(rec.draws) == 1                               # one chunk, one draw
        pairs = comb[rec.draws[0][:iters]]
    p, num, msk = G.triangulate_tracks_chunk(ext, tn.numpy(), pairs, lo_num=min(50, len(pairs)), track_vis=vis,
                                             track_score=score)
    assert np.array_equal(num, numr.numpy())
    assert np.array_equal(msk, mskr.numpy())
    ok = numr.numpy() >= 2
    np.testing.assert_allclose(p[ok], pr.numpy()[ok], rtol=1e-7, atol=1e-8)


@pytest.mark.parametrize("bad", [float("inf"), float("nan")])
def test_reference_rejects_nonfinite_rays(bad):
    """ADVICE r3 (a NaN / inf normalised coordinate in a view the track is NOT visible in): the reference does not get as far
    as its all-view mean -- the view pairs of its RANSAC stage include the invisible views, the DLT matrix of such a pair is
    non-finite and ``torch.linalg.eigh`` raises (triangulation_helpers.py:87).  Finite normalised tracks are therefore a
    PRECONDITION of ``triangulate_tracks``; the drop-in raises the same exception type up front
    (vggsfm_amd/utils/triangulation.py), and the kernel's behaviour on such input is defined anyway (every RANSAC hypothesis of
    the track poisoned: tests/test_gpu_triangulation.py::test_nonfinite_ray_in_invisible_view)."""
    tri, helpers, _ = ref_harness.load()
    seed, S, N = 117, 12, 90
    sc, ext, K, extra, _ = _scene(seed, S, N, "SIMPLE_PINHOLE", False)
    vis = sc.vis.copy()
    tn = helpers.cam_from_img(T(sc.tracks), T(K), T(extra)).numpy().copy()
    vis[S - 1, 3] = 0.0
    tn[S - 1, 3, 0] = bad
    with pytest.raises(torch.linalg.LinAlgError):
        tri.triangulate_tracks(T(ext), tc(T(tn)), max_ransac_iters=256, track_vis=tc(T(vis)), track_score=tc(T(sc.score)))
