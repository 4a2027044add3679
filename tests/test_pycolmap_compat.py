"""``vggsfm_amd.pycolmap_compat`` -- the pycolmap object surface over flat arrays -- on the CPU.

* object-model unit tests (run everywhere): construction through the pycolmap API (the way the reference's loops do it)
  against bulk construction, COLMAP's deletion semantics, ``.bin`` write -> read round trip;
* with the reference tree present (build container): the reference's OWN, UNMODIFIED code on top of it --
  ``tensor_to_pycolmap.{batch_matrix_to_pycolmap, pycolmap_to_batch_matrix}``, ``Triangulator.forward`` (reproduces the
  committed golden vectors), ``VGGSfMRunner.{sparse_reconstruct, rename_colmap_recons_and_rescale_camera,
  save_sparse_reconstruction, extract_sparse_depth_and_point_from_reconstruction}`` bound to a bare instance
  (runner.py:287-625, 744-772, 887-911, 1009-1054), read back with the reference's ``imc_helper.read_model``.

No GPU here, so the solvers behind the compat entry points are the CPU oracle (tests/cpu_backend.py, monkeypatched);
the same entry points on the device solvers: tests/test_gpu_pycolmap_compat.py.
"""
import glob
import os
import sys
import types
import warnings

import numpy as np
import pytest
import torch

from oracle import ref_harness
from vggsfm_amd import pycolmap_compat as pc
from vggsfm_amd.scene import make_scene, project

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
needs_reference = pytest.mark.skipif(not ref_harness.available(), reason="reference tree not present (GPU box)")


def _api_build(pts, ext, K, tracks, masks, size, shared, cam, extra):
    """The reference's construction loop (tensor_to_pycolmap.py:62-158), through the pycolmap API."""
    rec = pc.Reconstruction()
    valid = np.nonzero(masks.sum(0) >= 2)[0]
    for v in valid:
        rec.add_point3D(pts[v], pc.Track(), np.zeros(3))
    camera = None
    for f in range(len(ext)):
        if camera is None or not shared:
            prm = [K[f, 0, 0], K[f, 0, 2], K[f, 1, 2]] + ([extra[f, 0]] if cam == "SIMPLE_RADIAL" else [])
            camera = pc.Camera(model=cam, width=size[0], height=size[1], params=np.array(prm), camera_id=f)
            rec.add_camera(camera)
        image = pc.Image(id=f, name=f"image_{f}", camera_id=camera.camera_id,
                         cam_from_world=pc.Rigid3d(pc.Rotation3d(ext[f][:3, :3]), ext[f][:3, 3]))
        p2, k = [], 0
        for pid in range(1, len(valid) + 1):
            if (rec.points3D[pid].xyz < 3000).all() and masks[f][valid[pid - 1]]:
                p2.append(pc.Point2D(tracks[f][valid[pid - 1]], pid))
                rec.points3D[pid].track.add_element(f, k)
                k += 1
        image.points2D = pc.ListPoint2D(p2)
        image.registered = True
        rec.add_image(image)
    return rec


def _scene(cam="SIMPLE_RADIAL", shared=True, S=6, N=60, seed=5):
    sc = make_scene(S, N, cam, shared_camera=shared, seed=seed)
    sc.points3D[3] = [4000.0, 0.0, 1.0]                      # beyond max_points3D_val
    sc.mask[1:, 7] = False                                   # a single observation: not a track
    return sc


@pytest.mark.parametrize("cam,shared", [("SIMPLE_PINHOLE", False), ("SIMPLE_RADIAL", True)])
def test_api_construction_equals_bulk_construction(tmp_path, cam, shared):
    sc = _scene(cam, shared)
    a = _api_build(sc.points3D, sc.extrinsics, sc.intrinsics, sc.tracks, sc.mask, [1024, 1024], shared, cam, sc.extra_params)
    b = pc.Reconstruction.from_arrays(sc.points3D, sc.extrinsics, sc.intrinsics, sc.tracks, sc.mask, [1024, 1024],
                                      shared_camera=shared, camera_type=cam, extra_params=sc.extra_params)
    assert a.num_points3D() == b.num_points3D() and a.reg_image_ids() == b.reg_image_ids() == list(range(sc.S))
    assert sorted(a.cameras) == sorted(b.cameras) and len(a.cameras) == (1 if shared else sc.S)
    for x, y in zip(a.problem_arrays()[1:], b.problem_arrays()[1:]):
        assert (x is None and y is None) or np.array_equal(np.asarray(x), np.asarray(y))
    for pid in (1, 5, a.num_points3D()):
        assert a.points3D[pid].track.elements == b.points3D[pid].track.elements
        assert a.points3D[pid].track.length() == len(a.points3D[pid].track.elements)
    a.write(str(tmp_path / "a"))
    b.write(str(tmp_path / "b"))
    for name in ("cameras.bin", "images.bin", "points3D.bin"):
        assert (tmp_path / "a" / name).read_bytes() == (tmp_path / "b" / name).read_bytes()
    c = pc.Reconstruction(str(tmp_path / "a"))                 # reader
    for x, y in zip(a.problem_arrays()[1:-2], c.problem_arrays()[1:-2]):
        assert (x is None and y is None) or np.allclose(np.asarray(x, float), np.asarray(y, float), atol=1e-12)
    assert a.problem_arrays()[-2:] == c.problem_arrays()[-2:]
    assert c.images[2].name == "image_2" and c.cameras[0].model == cam


def test_views_write_through_and_point2d_semantics():
    sc = _scene()
    rec = pc.Reconstruction.from_arrays(sc.points3D, sc.extrinsics, sc.intrinsics, sc.tracks, sc.mask, [1024, 1024],
                                        shared_camera=True, camera_type="SIMPLE_RADIAL", extra_params=sc.extra_params)
    p = rec.points3D[2]
    p.xyz = [1.0, 2.0, 3.0]
    p.color = np.array([7.9, 8.2, 9.0])                        # (float colours are truncated to uint8)
    assert rec.points3D[2].xyz.tolist() == [1.0, 2.0, 3.0] and rec.points3D[2].color.tolist() == [7, 8, 9]
    assert torch.from_numpy(rec.points3D[2].xyz).dtype == torch.float64
    im = rec.images[1]
    q = im.points2D[0]
    before = q.xy
    q.xy = (q.xy - np.array([3.0, 4.0])) * 2.0
    assert np.array_equal(rec.images[1].points2D[0].xy, (before - [3.0, 4.0]) * 2.0)
    for pt in im.points2D:                                     # the runner's loop (runner.py:1043-1045)
        pt.xy = pt.xy + 1.0
    assert np.array_equal(rec.images[1].points2D[0].xy, (before - [3.0, 4.0]) * 2.0 + 1.0)
    assert q.has_point3D() and pc.Point2D([1.0, 2.0]).point3D_id == pc.INVALID_POINT3D_ID
    cam = rec.cameras[0]
    cam.width, cam.height = torch.tensor(1920.0), torch.tensor(1080.0)          # the reference assigns 0-d tensors
    cam.params = np.array([900.0, 960.0, 540.0, 0.01])
    assert (cam.width, cam.height) == (1920, 1080) and cam.calibration_matrix()[0, 2] == 960.0
    X = np.array([0.3, -0.2, 2.0])
    uv = cam.img_from_cam(X)
    np.testing.assert_allclose(cam.cam_from_img(uv), X[:2] / X[2], atol=1e-9)   # project -> iterative undistortion
    T = pc.Rigid3d(pc.Rotation3d(sc.extrinsics[2][:3, :3]), sc.extrinsics[2][:3, 3])
    np.testing.assert_allclose(T * X, sc.extrinsics[2][:, :3] @ X + sc.extrinsics[2][:, 3])
    np.testing.assert_allclose((T.inverse() * T).matrix(), np.eye(3, 4), atol=1e-12)
    np.testing.assert_allclose(pc.Rotation3d(T.rotation.quat).matrix(), T.rotation.matrix(), atol=1e-12)


def test_deletion_semantics_follow_colmap():
    """DeRegisterImage / DeleteObservation / DeletePoint3D [COLMAP 3.10 reconstruction.cc]: an element removed from a
    track of length <= 2 takes the whole point; 2D points of a deleted point lose their point3D id."""
    sc = _scene("SIMPLE_PINHOLE", False, S=5, N=40, seed=3)
    rec = pc.Reconstruction.from_arrays(sc.points3D, sc.extrinsics, sc.intrinsics, sc.tracks, sc.mask, [1024, 1024])
    valid = np.nonzero(sc.mask.sum(0) >= 2)[0]
    length = {k + 1: int(sc.mask[:, v].sum()) for k, v in enumerate(valid) if (sc.points3D[v] < 3000).all()}
    seen3 = {k + 1 for k, v in enumerate(valid) if sc.mask[3, v] and k + 1 in length}
    n0 = rec.num_points3D()
    rec.deregister_image(3)
    gone = {p for p in seen3 if length[p] <= 2}
    assert rec.reg_image_ids() == [0, 1, 2, 4] and not rec.images[3].registered and 3 in rec.images
    assert rec.num_points3D() == n0 - len(gone)
    assert rec.images[3].num_points3D() == 0
    for p in seen3 - gone:
        assert rec.points3D[p].track.length() == length[p] - 1
        assert 3 not in [e.image_id for e in rec.points3D[p].track.elements]
    for p in gone:
        assert p not in rec.points3D and all((im.points2D._pid != p).all() for im in rec.images.values())
    pid = rec.add_point3D(np.array([1.0, 2.0, 3.0]), pc.Track(), np.array([1, 2, 3]))
    assert pid == n0 + 1 and rec.points3D[pid].track.length() == 0 and max(rec.point3D_ids()) == pid
    del rec.points3D[pid]
    assert pid not in rec.point3D_ids()
    rec.images[3].registered = True
    assert rec.reg_image_ids() == [0, 1, 2, 4, 3]


def test_normalize_matches_tensor_version():
    from vggsfm_amd.ba import normalize_reconstruction
    sc = _scene("SIMPLE_PINHOLE", False, S=9, N=50, seed=8)
    rec = pc.Reconstruction.from_arrays(sc.points3D, sc.extrinsics, sc.intrinsics, sc.tracks, sc.mask, [1024, 1024])
    valid = np.nonzero(sc.mask.sum(0) >= 2)[0]
    e, p = normalize_reconstruction(torch.from_numpy(sc.extrinsics), torch.from_numpy(sc.points3D[valid]))
    rec.normalize(5.0, 0.1, 0.9, True)
    got = np.stack([rec.images[i].cam_from_world.matrix() for i in range(9)])
    assert np.array_equal(got, e.numpy()) and np.array_equal(rec._xyz[:len(valid)], p.numpy())


# ----------------------------------------------------------------------------------------------- reference code on top
def _reference_module(name, monkeypatch):
    """Import a reference module and bind ITS `pycolmap` / `pyceres` names to the compat modules."""
    from vggsfm_amd import pyceres_compat
    ref_harness.install()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mod = __import__(name, fromlist=["_"])
    for attr, val in (("pycolmap", pc), ("pyceres", pyceres_compat)):
        if hasattr(mod, attr):
            monkeypatch.setattr(mod, attr, val)
    return mod


@needs_reference
@pytest.mark.parametrize("cam,shared", [("SIMPLE_PINHOLE", False), ("SIMPLE_RADIAL", True)])
def test_reference_tensor_to_pycolmap_runs_on_compat(monkeypatch, cam, shared):
    """The reference's own batch_matrix_to_pycolmap / pycolmap_to_batch_matrix, unmodified, with compat as `pycolmap`
    == the drop-in's bulk versions."""
    ref = _reference_module("vggsfm.utils.tensor_to_pycolmap", monkeypatch)
    from vggsfm_amd.utils import tensor_to_pycolmap as ours
    sc = _scene(cam, shared)
    T = torch.from_numpy
    args = (T(sc.points3D), T(sc.extrinsics), T(sc.intrinsics), T(sc.tracks), T(sc.mask), torch.tensor([1024, 1024]))
    kw = dict(shared_camera=shared, camera_type=cam, extra_params=None if sc.extra_params is None else T(sc.extra_params))
    a, b = ref.batch_matrix_to_pycolmap(*args, **kw), ours.batch_matrix_to_pycolmap(*args, **kw)
    assert isinstance(a, pc.Reconstruction)
    for x, y in zip(a.problem_arrays()[1:], b.problem_arrays()[1:]):
        assert (x is None and y is None) or np.array_equal(np.asarray(x), np.asarray(y))
    a.delete_point3D(4), b.delete_point3D(4)
    for x, y in zip(ref.pycolmap_to_batch_matrix(a, device="cpu", camera_type=cam),
                    ours.pycolmap_to_batch_matrix(b, device="cpu", camera_type=cam)):
        assert (x is None and y is None) or torch.equal(x, y)
    for x, y in zip(ref.pycolmap_to_batch_matrix(b, device="cpu", camera_type=cam),      # and crosswise
                    ours.pycolmap_to_batch_matrix(a, device="cpu", camera_type=cam)):
        assert (x is None and y is None) or torch.equal(x, y)


def _golden_inputs(g, dev="cpu"):
    W = int(g["W"])
    t = torch.from_numpy(g["images_small"])
    r = W // t.shape[-1]
    images = t.repeat_interleave(r, dim=-1).repeat_interleave(r, dim=-2)
    cams = types.SimpleNamespace(R=torch.from_numpy(g["R"]), T=torch.from_numpy(g["T"]),
                                 focal_length=torch.from_numpy(np.stack([g["focal_ndc"]] * 2, -1)))
    prelim = {"fmat_inlier_mask": torch.from_numpy(g["fmat_inlier"])[None]}
    return cams, images, prelim


@needs_reference
@pytest.mark.parametrize("case", ["pinhole_shared_s8", "radial_shared_s12"])
def test_reference_triangulator_forward_on_compat_reproduces_golden(monkeypatch, case):
    """Cut line B2: the reference's unmodified ``Triangulator.forward`` (-> init_BA, init_refine_pose, refine_pose,
    global_BA, iterative_global_BA, batch_matrix_to_pycolmap, pycolmap_to_batch_matrix) with ``pycolmap_compat`` as
    its pycolmap.  The golden vectors were produced by the same reference code over oracle/pycolmap_shim.py (plain
    Python objects); with the same solver arithmetic behind it the compat object model must give the same answer."""
    from tests import cpu_backend
    from oracle.gen_golden import _StableSort
    cpu_backend.patch(monkeypatch)
    for m in ("vggsfm.utils.tensor_to_pycolmap", "vggsfm.utils.triangulation_helpers", "vggsfm.utils.triangulation"):
        _reference_module(m, monkeypatch)
    tri = _reference_module("vggsfm.models.triangulator", monkeypatch)
    g = np.load(os.path.join(GOLD, f"triangulator_{case}.npz"), allow_pickle=False)
    cam, shared = str(g["camera_type"]), bool(g["shared"])
    kw = {str(k): int(v) for k, v in zip(g["kw_keys"], g["kw_vals"])}
    cams, images, prelim = _golden_inputs(g)
    torch.manual_seed(0)
    with _StableSort(), warnings.catch_warnings(), torch.no_grad():
        warnings.simplefilter("ignore")
        out = tri.Triangulator()(cams, torch.from_numpy(g["tracks"])[None], torch.from_numpy(g["vis"])[None], images,
                                 prelim, pred_score=torch.from_numpy(g["score"])[None], shared_camera=shared,
                                 camera_type=cam, **kw)
    ext, K, extra, pts, rgb, rec, vframes, v2d, vtracks = out
    assert isinstance(rec, pc.Reconstruction)
    assert np.array_equal(vtracks.numpy(), g["out_valid_tracks"]) and np.array_equal(v2d.numpy(), g["out_valid_2D"])
    assert np.array_equal(vframes.numpy(), g["out_valid_frames"])
    # (not bitwise: Reconstruction.normalize runs as torch ops here and as numpy in the shim -- last-bit differences that
    #  the following LM solves amplify to ~1e-9)
    np.testing.assert_allclose(ext.numpy(), g["out_extrinsics"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(pts.numpy(), g["out_points3D"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(K.numpy(), g["out_intrinsics"], rtol=1e-8)
    assert rec.num_points3D() == int(g["rec_num_points3D"])
    # the colours the reference wrote point by point (triangulator.py:335-342)
    assert np.array_equal(rec._rgb[:len(pts)], np.round(rgb.numpy() * 255).astype(np.uint8))


class _ReplayTriangulator:
    """Stands where ``runner.triangulator`` is: replays the 9-tuple of a golden case (the drop-in reproduces it on the GPU,
    tests/test_gpu_triangulator_golden.py) with the reconstruction built by the drop-in's own tail
    (``vggsfm_amd.models.triangulator.build_reconstruction``)."""

    def __init__(self, g):
        self.g = g

    def __call__(self, pred_cameras, pred_tracks, pred_vis, images, preliminary_dict, pred_score=None, **kw):
        from vggsfm_amd.models.triangulator import build_reconstruction
        g = self.g
        cam, shared = str(g["camera_type"]), bool(g["shared"])
        T = torch.from_numpy
        ext, K, pts = T(g["out_extrinsics"]), T(g["out_intrinsics"]), T(g["out_points3D"])
        extra = T(g["out_extra"]) if cam == "SIMPLE_RADIAL" else None
        vt, v2d = T(g["out_valid_tracks"]), T(g["out_valid_2D"])
        size = torch.tensor([images.shape[-1], images.shape[-2]], dtype=pred_tracks.dtype)
        rec, rgb = build_reconstruction(pts, ext, K, extra, T(g["tracks"]), vt, v2d[:, vt], size, images,
                                        shared_camera=shared, camera_type=cam)
        return ext, K, extra, pts, rgb, rec, T(g["out_valid_frames"]), v2d, vt


def _runner_harness(monkeypatch, g, triangulator, W=1024):
    """A bare ``VGGSfMRunner`` (no hydra, no checkpoints) whose learned parts are synthetic: the camera predictor returns
    the golden case's predicted cameras, ``predict_tracks`` its tracks (and, for the dense pass, exact projections of
    the plane z = 4 through the ground-truth cameras).  Everything else is the reference's code, unmodified."""
    R = _reference_module("vggsfm.runners.runner", monkeypatch)
    cam, shared = str(g["camera_type"]), bool(g["shared"])
    S, N = g["tracks"].shape[:2]
    seeds = {"pinhole_s10": 31, "radial_shared_s12": 32, "pinhole_shared_s8": 33, "radial_s26_randperm": 34}
    case = [k for k in seeds if g["tracks"].shape[0] == int(k.split("_s")[-1].split("_")[0])][0]
    sc = make_scene(S, N, cam, shared_camera=shared, seed=seeds[case], outlier_frac=0.03)
    cams, images, prelim = _golden_inputs(g)

    def predict_tracks(query_method, max_query_pts, track_predictor, images_, masks_, fmaps_, query_frame_indexes,
                       fine_tracking, bound_bboxes=None, query_points_dict=None):
        Sw = images_.shape[1]
        if query_points_dict is None:
            assert Sw == S
            return (torch.from_numpy(g["tracks"])[None], torch.from_numpy(g["vis"])[None],
                    torch.from_numpy(g["score"])[None])
        f0 = int(round(float(fmaps_[0, 0, 0])))                 # first frame of the neighbourhood (encoded in the fmaps)
        rel = query_frame_indexes[0]
        grid = query_points_dict[rel][0].double().numpy()
        E, Kq = sc.extrinsics[f0 + rel], sc.intrinsics[f0 + rel]
        rays = np.concatenate([(grid - Kq[:2, 2]) / Kq[0, 0], np.ones((len(grid), 1))], 1) @ E[:, :3]    # world directions
        c = -E[:, :3].T @ E[:, 3]
        X = c[None] + ((4.0 - c[2]) / rays[:, 2])[:, None] * rays
        uv, depth = project(X, sc.extrinsics[f0:f0 + Sw], sc.intrinsics[f0:f0 + Sw],
                            None if sc.extra_params is None else sc.extra_params[f0:f0 + Sw])
        vis = ((uv > 0) & (uv < W)).all(-1) & (depth > 0)
        return (torch.from_numpy(uv.astype(np.float32))[None], torch.from_numpy(vis.astype(np.float32))[None],
                torch.ones((1, Sw, len(grid))))

    monkeypatch.setattr(R, "predict_tracks", predict_tracks)
    monkeypatch.setattr(R, "estimate_preliminary_cameras", lambda *a, **k: (None, prelim))
    runner = object.__new__(R.VGGSfMRunner)
    runner.cfg = types.SimpleNamespace(
        query_by_midpoint=True, query_by_interval=False, center_order=False, avg_pose=False, query_method="sp",
        max_query_pts=N, fine_tracking=False, comple_nonvis=False, visual_tracks=False, use_poselib=False, fmat_thres=4.0,
        BA_iters=1, shared_camera=shared, max_reproj_error=4, init_max_reproj_error=4, extract_color=True, robust_refine=1,
        camera_type=cam, extra_pt_pixel_interval=128, extra_by_neighbor=6, concat_extra_points=True,
        filter_invalid_frame=True, shift_point2d_to_original_res=True)
    runner.dtype, runner.remove_borders = torch.float32, 5
    runner.camera_predictor = lambda imgs, batch_size=1: {"pred_cameras": cams}
    runner.track_predictor = types.SimpleNamespace(
        process_images_to_fmaps=lambda imgs: torch.arange(S, dtype=torch.float32)[None, :, None].expand(1, S, 4).clone())
    runner.triangulator = triangulator
    # original images 2048 x 1536 (w, h) were padded to a square and resized to W: content starts at y = 128
    crop = torch.zeros((1, S, 8))
    crop[0, :, 0], crop[0, :, 1] = 2048.0, 1536.0
    crop[0, :, 4], crop[0, :, 5] = 0.0, -128.0
    paths = [f"/data/scene/images/frame_{S - s:03d}.png" for s in range(S)]      # (sorted order != frame order)
    return R, runner, images, crop, paths, sc


@needs_reference
@pytest.mark.parametrize("case", ["pinhole_s10", "radial_shared_s12"])
def test_reference_runner_tail_on_dropin_reconstruction(monkeypatch, tmp_path, case):
    """the reference's unmodified ``VGGSfMRunner.sparse_reconstruct`` (+ extra points, frame
    filtering, ``rename_colmap_recons_and_rescale_camera``, ``save_sparse_reconstruction``,
    ``extract_sparse_depth_and_point_from_reconstruction``) consumes the reconstruction the drop-in returns, with
    back_to_original_resolution=True and extra_pt_pixel_interval > 0; the written model is read back with the
    reference's own reader."""
    g = np.load(os.path.join(GOLD, f"triangulator_{case}.npz"), allow_pickle=False)
    Rm, runner, images, crop, paths, sc = _runner_harness(monkeypatch, g, _ReplayTriangulator(g))
    shared = bool(g["shared"])
    S = images.shape[1]
    with warnings.catch_warnings(), torch.no_grad():
        warnings.simplefilter("ignore")
        pred = runner.sparse_reconstruct(images, masks=None, crop_params=crop, query_frame_num=3, image_paths=paths,
                                         seq_name="synthetic", output_dir=str(tmp_path), back_to_original_resolution=True)
    rec = pred["reconstruction"]
    assert isinstance(rec, pc.Reconstruction)
    n_sfm, n_add = pred["additional_points_dict"]["sfm_points_num"], pred["additional_points_dict"]["additional_points_num"]
    assert n_sfm == int(g["out_valid_tracks"].sum()) and n_add > 50
    assert rec.num_points3D() == n_sfm + n_add == len(pred["points3D"])
    names = [os.path.basename(p) for p in paths]
    assert [rec.images[i].name for i in range(S)] == names
    # intrinsics at the original resolution, rows in the order of the sorted file names (runner.py:593-608)
    Ko = pred["intrinsics_opencv"].numpy()
    order = np.argsort(names)
    np.testing.assert_allclose(Ko[:, 0, 0], 2.0 * g["out_intrinsics"][order if not shared else 0][..., 0, 0], rtol=1e-6)   # (the reference scales in float32)
    assert (Ko[:, 0, 2] == 1024).all() and (Ko[:, 1, 2] == 768).all()
    runner.save_sparse_reconstruction(pred, output_dir=str(tmp_path))
    from vggsfm.datasets.imc_helper import read_model
    cameras, imgs, points3D = read_model(str(tmp_path / "sparse"), ext=".bin")
    assert len(cameras) == (1 if shared else S) and len(imgs) == S and len(points3D) == n_sfm + n_add
    assert (cameras[0].width, cameras[0].height) == (2048, 1536)
    v2d, vt = g["out_valid_2D"], g["out_valid_tracks"]
    for s in range(S):
        assert imgs[s].name == names[s]
        pids = np.nonzero(v2d[s, vt])[0]
        assert np.array_equal(imgs[s].point3D_ids, pids + 1)
        np.testing.assert_allclose(imgs[s].xys, (g["tracks"][s][vt][pids].astype(np.float64) - [0.0, 128.0]) * 2.0)
    assert all(len(points3D[p].image_ids) == 0 for p in range(n_sfm + 1, n_sfm + n_add + 1))
    assert all(len(points3D[p].image_ids) == int(v2d[:, vt][:, p - 1].sum()) for p in range(1, n_sfm + 1))
    # the dense pass triangulated the plane z = 4 of the ground-truth frame: in the model's gauge it must be planar too
    add = pred["points3D"][n_sfm:].numpy()
    c = add - add.mean(0)
    sv = np.linalg.svd(c, compute_uv=False)
    assert sv[2] / sv[1] < 2e-2
    # sparse depth extraction walks points3D -> track.elements -> images / cameras (runner.py:744-772)
    pred = runner.extract_sparse_depth_and_point_from_reconstruction(pred)
    assert set(pred["sparse_depth"]) == set(names)
    k = names[1]
    uvd = np.stack(pred["sparse_depth"][k])
    obs = imgs[1].xys[imgs[1].point3D_ids > 0]
    assert len(uvd) == len(obs) and np.median(np.linalg.norm(uvd[:, :2] - obs, axis=1)) < 2.0 and (uvd[:, 2] > 0).all()


@needs_reference
def test_reference_runner_end_to_end_on_compat(monkeypatch, tmp_path):
    """The whole post-tracker pipeline as the reference wrote it -- ``sparse_reconstruct`` -> the reference's own
    ``Triangulator`` -> ``pycolmap`` = compat -- runs and recovers the scene."""
    from tests import cpu_backend
    from oracle.gen_golden import _StableSort
    cpu_backend.patch(monkeypatch)
    for m in ("vggsfm.utils.tensor_to_pycolmap", "vggsfm.utils.triangulation_helpers", "vggsfm.utils.triangulation"):
        _reference_module(m, monkeypatch)
    tri = _reference_module("vggsfm.models.triangulator", monkeypatch)
    g = np.load(os.path.join(GOLD, "triangulator_pinhole_shared_s8.npz"), allow_pickle=False)
    Rm, runner, images, crop, paths, sc = _runner_harness(monkeypatch, g, tri.Triangulator())
    runner.cfg.extra_pt_pixel_interval = -1
    torch.manual_seed(0)
    with _StableSort(), warnings.catch_warnings(), torch.no_grad():
        warnings.simplefilter("ignore")
        pred = runner.sparse_reconstruct(images, masks=None, crop_params=crop, query_frame_num=3, image_paths=paths,
                                         seq_name="synthetic", output_dir=str(tmp_path), back_to_original_resolution=True)
    rec = pred["reconstruction"]
    assert isinstance(rec, pc.Reconstruction) and rec.num_points3D() > 0.8 * g["tracks"].shape[1]
    assert rec.cameras[0].width == 2048 and len(rec.cameras) == 1
    runner.save_sparse_reconstruction(pred, output_dir=str(tmp_path))
    back = pc.Reconstruction(str(tmp_path / "sparse"))
    assert back.num_points3D() == rec.num_points3D() and back.images[0].name == os.path.basename(paths[0])


@needs_reference
def test_reference_video_runner_joint_ba_and_alignment_on_compat(monkeypatch):
    """Cut line B2 for the video path: the reference's unmodified ``VideoRunner.{convert_pred_to_point_frame_dict,
    dicts_to_reconstruction, joint_BA, reconstruction_to_dicts, build_camera_for_video, align_next_window}`` and the
    module-level ``solve_bundle_adjustment`` (video_runner.py:354-473, 494-638, 940-1060, 1321-1331) bound to a bare
    instance, with ``pycolmap`` / ``pyceres`` = the compat modules.  The joint BA must agree with the drop-in's tensor
    path (``vggsfm_amd.video.joint_bundle_adjustment``) on the same state -- two routes, one answer."""
    from collections import defaultdict
    from tests import cpu_backend
    from vggsfm_amd import pyceres_compat
    from vggsfm_amd import video as V
    cpu_backend.patch(monkeypatch)
    VR = _reference_module("vggsfm.runners.video_runner", monkeypatch)
    sc = make_scene(12, 500, "SIMPLE_RADIAL", shared_camera=True, seed=41, outlier_frac=0.0)
    from vggsfm_amd.scene import perturb_for_ba
    ext0, K0, xp0, pts0 = perturb_for_ba(sc, seed=41)
    valid = sc.mask.sum(0) >= 3
    T_ = torch.from_numpy
    runner = object.__new__(VR.VideoRunner)
    runner.cfg = types.SimpleNamespace(camera_type="SIMPLE_RADIAL", shared_camera=True)
    runner.device = "cpu"
    runner.point_dict, runner.frame_dict = {}, defaultdict(dict)
    runner.intrinsics = T_(K0[0:1]).float()
    runner.extra_params = T_(xp0[0:1]).float()
    runner.image_size = torch.tensor([1024, 1024])
    pred = {"pred_track": T_(sc.tracks), "pred_vis": T_(sc.vis), "valid_2D_mask": T_(sc.mask[:, valid]),
            "valid_tracks": T_(valid), "points3D": T_(pts0[valid]).float(), "points3D_rgb": None,
            "extrinsics_opencv": T_(ext0)}
    runner.convert_pred_to_point_frame_dict(pred, 0, 12)
    rec = runner.dicts_to_reconstruction(0, 12)
    assert isinstance(rec, pc.Reconstruction) and rec.num_points3D() == int(valid.sum()) and len(rec.cameras) == 1
    # the drop-in's tensor path on the same state (float32 points, as the dicts hold them)
    p_opt, e_opt, K_opt, x_opt, inl, keep, _ = V.joint_bundle_adjustment(
        T_(pts0[valid]).float().double(), T_(ext0), runner.intrinsics.double(), T_(sc.tracks[:, valid]), T_(sc.mask[:, valid]),
        runner.extra_params.double(), "SIMPLE_RADIAL", reproj_error=2.0, tri_angle=1.5, normalize=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        runner.joint_BA(0, 12, reproj_error=2.0, tri_angle=1.5, normalize=True)
    got_ext = torch.stack([runner.frame_dict[i]["extri"] for i in range(12)]).numpy()
    np.testing.assert_allclose(got_ext, e_opt.numpy(), atol=1e-6)
    np.testing.assert_allclose(runner.intrinsics[0].double().numpy(), K_opt[0].numpy(), rtol=1e-6)
    np.testing.assert_allclose(float(runner.extra_params[0, 0]), float(x_opt[0, 0]), atol=1e-6)
    assert len(runner.point_dict) == int(keep.sum())
    kept = p_opt[keep].numpy()
    got_pts = np.stack([runner.point_dict[i]["xyz"].numpy() for i in sorted(runner.point_dict)])
    np.testing.assert_allclose(got_pts, kept, atol=1e-5)
    n_obs = sum(len(runner.point_dict[i]["track"]) for i in runner.point_dict)
    assert n_obs == int(inl[:, keep].sum())
    # pose alignment of a window through pycolmap.pose_refinement (video_runner.py:940-1017)
    ext_gt = np.stack([runner.frame_dict[i]["extri"].numpy() for i in range(12)])
    pert = ext_gt.copy()
    pert[1:, :, 3] += 0.02
    refined = runner.align_next_window(T_(pert), T_(sc.tracks[:, valid][:, keep.numpy()]), T_(sc.mask[:, valid][:, keep.numpy()]),
                                       T_(got_pts).double())
    assert np.abs(refined.numpy()[1:] - ext_gt[1:]).max() < 0.3 * 0.02 and np.array_equal(refined.numpy()[0], pert[0])
    # window BA through BundleAdjustmentConfig / BundleAdjuster / pyceres.solve (video_runner.py:813-836, 1321-1331)
    rec2 = runner.dicts_to_reconstruction(0, 12)
    opts = pc.BundleAdjustmentOptions()
    opts.refine_focal_length = opts.refine_extra_params = False
    cfg = pc.BundleAdjustmentConfig()
    for i in rec2.reg_image_ids():
        cfg.add_image(i)
    cfg.set_constant_cam_pose(rec2.reg_image_ids()[0])
    for pid in rec2.point3D_ids():
        (cfg.add_constant_point if pid <= 50 else cfg.add_variable_point)(pid)
    summary = VR.solve_bundle_adjustment(rec2, opts, cfg)
    assert isinstance(summary, pyceres_compat.SolverSummary) and VR.log_ba_summary(summary)
    assert summary.final_cost <= summary.initial_cost and summary.num_residuals_reduced > 0
