"""Video-window geometry (vggsfm_amd/video.py) against the oracle: reference call sites
vggsfm/runners/video_runner.py:494-541,800-838,905-1017,1189-1262 and vggsfm/utils/align.py:145-252."""
import numpy as np
import pytest
import torch

from oracle import ba as OB
from oracle import geometry as G
from vggsfm_amd import video as V
from vggsfm_amd.scene import make_scene, perturb_for_ba


def D(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _rand_rot(n, gen):
    q = torch.randn(n, 4, generator=gen, dtype=torch.float64)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                        2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                        2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], 1).reshape(n, 3, 3)


def test_align_camera_extrinsics_known_answer():
    """The reference's own self-check (vggsfm/utils/align.py:255-297): recover a random similarity applied to 10
    random cameras, atol 1e-3 -- here in float64 with 200 trials and a 1e-9 bound."""
    gen = torch.Generator().manual_seed(0)
    for _ in range(200):
        src = torch.cat([_rand_rot(10, gen), torch.randn(10, 3, 1, generator=gen, dtype=torch.float64)], -1)
        Rt, Tt = _rand_rot(1, gen), torch.randn(1, 3, generator=gen, dtype=torch.float64)
        s = torch.rand(1, generator=gen, dtype=torch.float64) * 2 + 0.1
        R_tgt, T_tgt = V.apply_transformation(src, Rt, Tt, s, return_extri=False)
        aR, aT, a_s = V.align_camera_extrinsics(src, torch.cat([R_tgt, T_tgt[..., None]], -1))
        R_v, T_v = V.apply_transformation(src, aR, aT, a_s, return_extri=False)
        assert torch.allclose(R_tgt, R_v, atol=1e-9) and torch.allclose(T_tgt, T_v, atol=1e-9)
        assert abs(float(a_s) - float(s)) < 1e-9


@pytest.mark.gpu
def test_filter_points_and_compute_masks_matches_oracle():
    S, N = 17, 3000
    sc = make_scene(S, N, "SIMPLE_RADIAL", shared_camera=True, seed=41, outlier_frac=0.05)
    pts = sc.points3D + np.random.default_rng(0).normal(0, 0.01, sc.points3D.shape)
    K1, x1 = sc.intrinsics[0:1], sc.extra_params[0:1]
    fp, ft, fm, valid = V.filter_points_and_compute_masks(D(pts), D(sc.tracks), D(sc.extrinsics), D(K1), D(x1))
    _, detail = G.filter_all_points3D(pts, sc.tracks.astype(np.float64), sc.extrinsics, sc.intrinsics, sc.extra_params,
                                      max_reproj_error=4, return_detail=True, hard_max=-1, check_triangle=True)
    v_ref = detail.sum(0) >= 3
    assert np.array_equal(valid.cpu().numpy(), v_ref)                       # bit-exact masks
    assert np.array_equal(fm.cpu().numpy(), detail[:, v_ref])
    assert np.array_equal(fp.cpu().numpy(), pts[v_ref]) and ft.shape == (S, int(v_ref.sum()), 2)


@pytest.mark.gpu
def test_align_next_window_matches_oracle():
    S, N = 9, 800
    sc = make_scene(S, N, "SIMPLE_RADIAL", shared_camera=True, seed=42, full_visibility=True, outlier_frac=0.02)
    ext0, _, _, _ = perturb_for_ba(sc, seed=42, rot_deg=1.0, trans=0.05)
    inl = sc.mask.copy()
    inl[3, 40:] = False                       # frame 3 has only 40 inliers -> uses all points
    ext = V.align_next_window(D(ext0), D(sc.tracks), D(inl), D(sc.points3D), D(sc.intrinsics[0:1]),
                              D(sc.extra_params[0:1]), "SIMPLE_RADIAL").cpu().numpy()
    assert np.array_equal(ext[0], ext0[0])
    params = np.array([sc.intrinsics[0, 0, 0], sc.intrinsics[0, 0, 2], sc.intrinsics[0, 1, 2], sc.extra_params[0, 0]])
    for f in range(1, S):
        m = inl[f] if inl[f].sum() > 50 else np.ones(N, bool)
        e_ref, p_ref, _ = OB.pose_refinement(ext0[f], sc.tracks[f], sc.points3D, m, params, "SIMPLE_RADIAL",
                                             refine_focal_length=False, refine_extra_params=False)
        np.testing.assert_allclose(ext[f], e_ref, atol=1e-7)
        assert np.array_equal(p_ref, params)
    # the refined poses are close to the ground truth (0.5 px noise, 800 points)
    assert np.abs(ext - sc.extrinsics).max() < 5e-3


@pytest.mark.gpu
def test_triangulate_window_tracks_matches_oracle_composition():
    S, N = 17, 1500
    sc = make_scene(S, N, "SIMPLE_RADIAL", shared_camera=True, seed=43, outlier_frac=0.05)
    K1, x1 = sc.intrinsics[0:1], sc.extra_params[0:1]
    fp, ft, fm, fv = V.triangulate_window_tracks(D(sc.tracks), D(sc.vis), D(sc.score), D(sc.extrinsics), D(K1), D(x1))
    tn = G.cam_from_img(sc.tracks, sc.intrinsics, sc.extra_params)
    pts_ref, _, _ = G.triangulate_tracks_chunk(sc.extrinsics, tn, G.generate_combinations(S), track_vis=sc.vis,
                                               track_score=sc.score)
    _, detail = G.filter_all_points3D(pts_ref, sc.tracks.astype(np.float64), sc.extrinsics, sc.intrinsics,
                                      sc.extra_params, max_reproj_error=4, return_detail=True, hard_max=-1)
    v_ref = detail.sum(0) >= 3
    assert fp.shape[0] == int(v_ref.sum()) and fv.shape == (S, int(v_ref.sum()))
    np.testing.assert_allclose(fp.cpu().numpy(), pts_ref[v_ref], rtol=1e-9, atol=1e-9)
    assert np.array_equal(fm.cpu().numpy(), detail[:, v_ref])
    assert np.array_equal(ft.cpu().numpy(), sc.tracks[:, v_ref])


@pytest.mark.gpu
def test_joint_bundle_adjustment_properties_and_oracle():
    S, N = 30, 2500
    sc = make_scene(S, N, "SIMPLE_RADIAL", shared_camera=True, seed=44, outlier_frac=0.03)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=44)
    pts, ext, K, extra, inl, keep, summ = V.joint_bundle_adjustment(D(pts0), D(ext0), D(K0[0:1]), D(sc.tracks), D(sc.mask),
                                                                     D(extra0[0:1]), "SIMPLE_RADIAL")
    assert summ["final_cost"] < summ["initial_cost"] and K.shape == (1, 3, 3) and extra.shape == (1, 1)
    # oracle: normalize -> BA -> (same filters in numpy) -> normalize
    e_n, p_n = OB.normalize_reconstruction(ext0, pts0)
    po, eo, Ko, xo, so = OB.bundle_adjustment(p_n, e_n, K0, sc.tracks, sc.mask, extra0, True, "SIMPLE_RADIAL",
                                              options=OB.ceres_options())
    vi = so["valid_idx"]
    tr = sc.tracks[:, vi].astype(np.float64)
    _, detail = G.filter_all_points3D(po, tr, eo, Ko, xo, max_reproj_error=2.0, check_triangle=False, return_detail=True,
                                      hard_max=-1)
    inl_ref = sc.mask[:, vi] & detail
    tr_far = np.where(inl_ref[..., None], tr, 1e9)
    keep_ref, _ = G.filter_all_points3D(po, tr_far, eo, Ko, xo, max_reproj_error=2.0, min_tri_angle=1.5,
                                        check_triangle=True, hard_max=-1)
    keep_ref = keep_ref & (inl_ref.sum(0) >= 2) & ~so["deleted"]
    ham = int((keep.cpu().numpy() != keep_ref).sum())
    print("joint BA keep Hamming", ham, "of", len(keep_ref), "kept", int(keep_ref.sum()))
    assert ham == 0
    assert np.array_equal(inl.cpu().numpy(), inl_ref & keep_ref[None])
    e_ref, p_ref = OB.normalize_reconstruction(eo, po, keep_ref)
    np.testing.assert_allclose(ext.cpu().numpy(), e_ref, atol=2e-6)
    np.testing.assert_allclose(pts.cpu().numpy()[keep_ref], p_ref[keep_ref], atol=2e-5)
    assert abs(float(K[0, 0, 0]) / Ko[0, 0, 0] - 1) < 1e-6


@pytest.mark.gpu
def test_triangulate_extra_points_matches_oracle_composition():
    """runner.py:696-736 (dense extra points of a frame neighbourhood)."""
    from vggsfm_amd.utils.triangulation import triangulate_extra_points
    S, N = 9, 2000
    sc = make_scene(S, N, "SIMPLE_RADIAL", shared_camera=False, seed=45, outlier_frac=0.05)
    pts, valid = triangulate_extra_points(D(sc.tracks), D(sc.vis), D(sc.score), D(sc.extrinsics), D(sc.intrinsics),
                                          D(sc.extra_params), max_reproj_error=4)
    tn = G.cam_from_img(sc.tracks, sc.intrinsics, sc.extra_params)
    p_ref, n_ref, _ = G.triangulate_tracks_chunk(sc.extrinsics, tn, G.generate_combinations(S), track_vis=sc.vis,
                                                 track_score=sc.score)
    v_ref, _ = G.filter_all_points3D(p_ref, sc.tracks.astype(np.float64), sc.extrinsics, sc.intrinsics, sc.extra_params,
                                     max_reproj_error=4)
    v_ref = v_ref & (n_ref > 3)
    assert np.array_equal(valid.cpu().numpy(), v_ref)
    np.testing.assert_allclose(pts.cpu().numpy()[v_ref], p_ref[v_ref], rtol=1e-9, atol=1e-9)
    assert v_ref.mean() > 0.3          # (9 frames, visibility windows of 3-4 frames: many tracks have <= 3 inliers)
