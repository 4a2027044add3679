"""``vggsfm_amd.runners.GeometryRunner.sparse_reconstruct_from_tracks`` = everything ``VGGSfMRunner.sparse_reconstruct``
does after the tracker (vggsfm/runners/runner.py:467-625), on a synthetic planar scene so that the dense extra-point
pass has a simulated tracker (pixel -> plane -> other frames)."""
import types

import numpy as np
import pytest
import torch

from vggsfm_amd.runners import GeometryConfig, GeometryRunner, generate_grid_samples, sample_subrange
from vggsfm_amd.scene import make_cameras, perturb_for_ba, project


def test_grid_and_subrange_helpers():
    g = generate_grid_samples(torch.tensor([[10.0, 20.0, 110.0, 70.0]]), pixel_interval=10)
    assert g.shape == (10 * 5, 2) and g[0].tolist() == [10.0, 20.0] and g[-1].tolist() == [110.0, 70.0]
    assert g[1].tolist() == [10.0, 32.5]                                          # x-major order, y runs fastest
    g2 = generate_grid_samples(torch.tensor([[0.0, 0.0, 200.0, 100.0]]), N=200)
    assert g2.shape == (20 * 10, 2)
    assert sample_subrange(20, 0, 6) == (0, 6) and sample_subrange(20, 19, 6) == (14, 20) and sample_subrange(20, 10, 6) == (7, 13)
    assert sample_subrange(4, 2, 6) == (0, 4)


@pytest.mark.gpu
def test_sparse_reconstruct_from_tracks_with_extra_points(tmp_path):
    S, N, W = 8, 2500, 1024
    dev = "cuda"
    D = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    ext, K, _ = make_cameras(S, "SIMPLE_PINHOLE", False, 5)
    rng = np.random.default_rng(5)
    # a smooth, non-planar surface in front of the cameras (a plane would make the per-frame focal lengths ambiguous)
    surf = lambda x, y: 4.0 + 0.5 * np.sin(1.5 * x) + 0.4 * np.cos(1.2 * y) + 0.2 * x
    xy = rng.uniform(-1.3, 1.3, size=(N, 2))
    pts = np.concatenate([xy, surf(xy[:, :1], xy[:, 1:])], 1)
    uv, depth = project(pts, ext, K, None)
    vis = ((uv > 4) & (uv < W - 4)).all(-1) & (depth > 0)
    tracks = (uv + rng.normal(0, 0.4, uv.shape)).astype(np.float32)
    out = (rng.random(vis.shape) < 0.05) & vis
    tracks[out] += rng.uniform(-60, 60, (int(out.sum()), 2)).astype(np.float32)
    sc = types.SimpleNamespace(extrinsics=ext, intrinsics=K, extra_params=None, points3D=pts, camera_type="SIMPLE_PINHOLE", S=S,
                               shared_camera=False)
    ext0, K0, _, _ = perturb_for_ba(sc, seed=5, rot_deg=0.5, trans=0.02, focal_rel=0.02)
    cams = types.SimpleNamespace(R=D(ext0[:, :, :3]).float(), T=D(ext0[:, :, 3]).float())
    f = D(K0[:, 0, 0] / (W / 2.0)).float()
    cams.focal_length = torch.stack([f, f], -1)
    images = torch.rand(1, S, 3, W, W, device=dev)
    names = [f"img_{s:02d}.png" for s in range(S)]
    crop = torch.zeros(1, S, 8, device=dev)
    crop[0, :, 0], crop[0, :, 1] = 2048.0, 1536.0
    bboxes = torch.tensor([[100.0, 100.0, 900.0, 900.0]], device=dev)[None].expand(1, S, 4)

    def extra_tracker(frame_idx, n0, n1, grid):
        # pixel of frame_idx -> ground-truth surface -> the neighbouring frames (what a perfect tracker would return)
        g = grid[0].double().cpu().numpy()
        rays = np.stack([(g[:, 0] - K[frame_idx, 0, 2]) / K[frame_idx, 0, 0], (g[:, 1] - K[frame_idx, 1, 2]) / K[frame_idx, 1, 1],
                         np.ones(len(g))], 1)
        R, t = ext[frame_idx, :, :3], ext[frame_idx, :, 3]
        c, d = -R.T @ t, rays @ R                                                   # centre, world directions
        lam = (4.0 - c[2]) / d[:, 2]
        for _ in range(40):                                                         # fixed point along the ray
            X = c[None] + lam[:, None] * d
            lam = (surf(X[:, 0], X[:, 1]) - c[2]) / d[:, 2]
        X = c[None] + lam[:, None] * d
        hit = np.abs(X[:, 2] - surf(X[:, 0], X[:, 1])) < 1e-9                       # (grazing rays may not converge)
        uvn, dep = project(X, ext[n0:n1], K[n0:n1], None)
        v = (((uvn > 0) & (uvn < W)).all(-1) & (dep > 0) & hit[None]).astype(np.float32)
        return D(uvn.astype(np.float32))[None], D(v)[None], D(np.ones_like(v))[None]

    cfg = GeometryConfig(fmat_thres=4.0, max_ransac_iters=1024, lo_num=100, extra_pt_pixel_interval=40, extra_by_neighbor=6,
                         concat_extra_points=True, shift_point2d_to_original_res=True)
    np.random.seed(0)
    torch.manual_seed(0)
    runner = GeometryRunner(cfg)
    pred = runner.sparse_reconstruct_from_tracks(cams, D(tracks)[None], D(vis.astype(np.float32))[None],
                                                 D(np.ones(vis.shape, np.float32))[None], images, crop_params=crop,
                                                 image_paths=names, extra_tracker=extra_tracker, bound_bboxes=bboxes)
    n_sfm, n_add = pred["additional_points_dict"]["sfm_points_num"], pred["additional_points_dict"]["additional_points_num"]
    assert n_sfm > 0.85 * N and n_add > 0.5 * S * 20 * 20                          # 20 x 20 grid points per frame
    assert pred["points3D"].shape[0] == n_sfm + n_add == pred["reconstruction"].num_points3D()
    assert pred["extrinsics_opencv"].shape == (S, 3, 4) and pred["valid_2D_mask"].shape == (S, N)
    # the dense points lie on the surface: bring the model into the ground-truth frame with the similarity that aligns
    # the sparse points (Umeyama), then compare z with the surface
    P = pred["points3D"].cpu().numpy()
    sp, dn = P[:n_sfm], P[n_sfm:]
    gt = pts[pred["valid_tracks"].cpu().numpy()]
    mu_s, mu_g = sp.mean(0), gt.mean(0)
    U, sv, Vt = np.linalg.svd((gt - mu_g).T @ (sp - mu_s))
    Rg = U @ np.diag([1, 1, np.sign(np.linalg.det(U @ Vt))]) @ Vt
    sg = sv.sum() / ((sp - mu_s) ** 2).sum()
    to_gt = lambda X: sg * (X - mu_s) @ Rg.T + mu_g
    assert np.abs(to_gt(sp) - gt).max() < 0.05
    dg = to_gt(dn)
    assert np.percentile(np.abs(dg[:, 2] - surf(dg[:, 0], dg[:, 1])), 99) < 0.03 and np.median(np.abs(dg[:, 2] - surf(dg[:, 0], dg[:, 1]))) < 2e-3
    # intrinsics at the original resolution: focal x 2048 / 1024, principal point = size // 2
    Ko = pred["intrinsics_opencv"].cpu().numpy()
    np.testing.assert_allclose(Ko[:, 0, 0], 2.0 * K[:, 0, 0], rtol=0.03)
    assert (Ko[:, 0, 2] == 1024).all() and (Ko[:, 1, 2] == 768).all()
    rec = pred["reconstruction"]
    assert [rec.images[i].name for i in range(S)] == names and (rec.cameras[0].width, rec.cameras[0].height) == (2048, 1536)
    rec.write(str(tmp_path))
    from vggsfm_amd.pycolmap_compat import Reconstruction
    back = Reconstruction(str(tmp_path))                                           # COLMAP .bin reader
    assert back.num_points3D() == rec.num_points3D() == n_sfm + n_add and back.num_reg_images() == S
    assert all(back.points3D[p].track.length() == 0 for p in (n_sfm + 1, n_sfm + n_add))
    assert np.array_equal(back.images[2].points2D._xy, rec.images[2].points2D._xy)


@pytest.mark.gpu
def test_c1_kitchen_plumbing(tmp_path):
    """BASELINE configs[0] ("examples/kitchen, 8 frames, 2048 query points, plumbing only") from the committed fixture
    tests/golden/c1_kitchen_inputs.npz (file names, crop parameters and thumbnails of 8 of the reference's 25 kitchen images,
    made by scripts/run_c1_kitchen.py --stage prepare where /root/reference exists): synthetic tracks / cameras are injected at
    the Triangulator boundary, the post-tracker path runs on the GPU, the COLMAP model is written at the ORIGINAL 1558 x 1039
    resolution and read back with the committed .bin reader."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "run_c1_kitchen.py")
    spec = importlib.util.spec_from_file_location("run_c1_kitchen", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = np.load(mod.FIXTURE)
    names = [str(n) for n in g["names"]]
    assert len(names) == 8 and g["thumbs"].shape == (8, 64, 64, 3) and g["crop_params"].shape == (8, 8)
    assert tuple(g["crop_params"][0, :2]) == (1558.0, 1039.0)
    res = mod.run(None, model_dir=str(tmp_path))
    rec, back, pred = res["_reconstruction"], res["_read_back"], res["_predictions"]
    assert res["registered_images"] == 8 and res["image_names"] == names
    assert res["camera0"]["model"] in ("SIMPLE_PINHOLE", 0) and (res["camera0"]["width"], res["camera0"]["height"]) == (1558, 1039)
    # most of the 3 x 2048 injected tracks (97 % inliers, visible in the content rows of the padded images) become points
    assert res["valid_tracks"] > 0.6 * 3 * 2048 and res["points3D"] == res["valid_tracks"] == back.num_points3D() == rec.num_points3D()
    assert res["mean_track_length"] >= 2.0
    # principal point = centre of the original image, every registered image keeps its 2D points through the round trip
    assert abs(res["camera0"]["params"][1] - 1558 // 2) <= 1 and abs(res["camera0"]["params"][2] - 1039 // 2) <= 1
    for i in sorted(back.images):
        assert np.array_equal(back.images[i].points2D._xy, rec.images[i].points2D._xy)
        assert back.images[i].camera_id == rec.images[i].camera_id
    assert all(os.path.getsize(os.path.join(str(tmp_path), f)) > 0 for f in ("cameras.bin", "images.bin", "points3D.bin"))
    # the poses are the injected ground truth up to the gauge: relative rotation of frame 0 -> frame 7 within 1e-2 rad
    E = pred["extrinsics_opencv"].double().cpu().numpy()
    from vggsfm_amd.scene import make_scene
    sc = make_scene(8, 3 * 2048, "SIMPLE_PINHOLE", shared_camera=False, seed=7, outlier_frac=0.03)
    rel = lambda X: X[7, :, :3] @ X[0, :, :3].T
    dR = rel(E) @ rel(sc.extrinsics).T
    assert np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1)) < 1e-2
