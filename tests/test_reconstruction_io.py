"""COLMAP binary model writer (vggsfm_amd/pycolmap_compat.py: Reconstruction.write) vs the reference's own reader
(vggsfm/datasets/imc_helper.py, imported through the stub harness when /root/reference exists) and vs a
minimal independent parser of the documented layout.  CPU only."""
import struct

import numpy as np
import pytest

from oracle import ref_harness
from vggsfm_amd.pycolmap_compat import Reconstruction
from vggsfm_amd.scene import make_scene


def _model(tmp_path, cam, shared):
    sc = make_scene(5, 40, cam, shared_camera=shared, seed=3)
    colors = (np.arange(40 * 3) % 255).reshape(40, 3).astype(np.uint8)
    rec = Reconstruction.from_arrays(sc.points3D, sc.extrinsics, sc.intrinsics, sc.tracks, sc.mask, [1024, 1024],
                                     shared_camera=shared, camera_type=cam, extra_params=sc.extra_params, colors=colors)
    rec.write(str(tmp_path))
    return sc, rec, colors


@pytest.mark.parametrize("cam,shared", [("SIMPLE_PINHOLE", False), ("SIMPLE_RADIAL", True)])
def test_binary_layout(tmp_path, cam, shared):
    sc, rec, colors = _model(tmp_path, cam, shared)
    with open(tmp_path / "cameras.bin", "rb") as f:
        n = struct.unpack("<Q", f.read(8))[0]
        assert n == (1 if shared else 5)
        cid, mid, w, h = struct.unpack("<iiQQ", f.read(24))
        assert (cid, mid, w, h) == (0, 2 if cam == "SIMPLE_RADIAL" else 0, 1024, 1024)
        k = 4 if cam == "SIMPLE_RADIAL" else 3
        prm = struct.unpack(f"<{k}d", f.read(8 * k))
        assert prm[0] == sc.intrinsics[0, 0, 0] and prm[1] == 512.0
    with open(tmp_path / "points3D.bin", "rb") as f:
        n = struct.unpack("<Q", f.read(8))[0]
        assert n == 40
        pid, x, y, z, r, g, b, err = struct.unpack("<Q3d3Bd", f.read(43))
        assert pid == 1 and (x, y, z) == tuple(sc.points3D[0]) and (r, g, b) == tuple(colors[0])
        ln = struct.unpack("<Q", f.read(8))[0]
        assert ln == int(sc.mask[:, 0].sum())


@pytest.mark.skipif(not ref_harness.available(), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("cam,shared", [("SIMPLE_PINHOLE", False), ("SIMPLE_RADIAL", True)])
def test_reference_reader_roundtrip(tmp_path, cam, shared):
    sc, rec, colors = _model(tmp_path, cam, shared)
    ref_harness.install()
    from vggsfm.datasets.imc_helper import read_model
    cameras, images, points3D = read_model(str(tmp_path), ext=".bin")
    assert len(cameras) == (1 if shared else 5) and len(images) == 5 and len(points3D) == 40
    assert cameras[0].model == cam
    for s in range(5):
        im = images[s]
        pids = np.nonzero(sc.mask[s])[0]
        assert np.array_equal(im.point3D_ids, pids + 1)
        np.testing.assert_allclose(im.xys, sc.tracks[s, pids].astype(np.float64))
        np.testing.assert_allclose(im.tvec, sc.extrinsics[s, :, 3])
        np.testing.assert_allclose(im.qvec2rotmat(), sc.extrinsics[s, :, :3], atol=1e-12)
    for p in range(40):
        pt = points3D[p + 1]
        np.testing.assert_array_equal(pt.xyz, sc.points3D[p])
        assert np.array_equal(pt.image_ids, np.nonzero(sc.mask[:, p])[0])
        # every track element points back at the right 2D point of its image
        for im_id, p2 in zip(pt.image_ids, pt.point2D_idxs):
            assert images[im_id].point3D_ids[p2] == p + 1


def test_tensor_to_pycolmap_drop_in_selection_rules():
    """vggsfm_amd.utils.tensor_to_pycolmap keeps the reference's rules (tensor_to_pycolmap.py:62-68,128-146): tracks with
    >= 2 inliers, 1-based ids in order, observations of points beyond max_points3D_val dropped, one shared camera."""
    import torch

    from vggsfm_amd.utils.tensor_to_pycolmap import batch_matrix_to_pycolmap, pycolmap_to_batch_matrix
    rng = np.random.default_rng(0)
    S, P = 5, 40
    pts = rng.normal(size=(P, 3))
    pts[3] = [4000.0, 0, 1]
    masks = rng.random((S, P)) > 0.4
    masks[:, 7] = False
    masks[1:, 8] = False                                   # a single observation: not a valid track
    ext = np.tile(np.eye(3, 4)[None], (S, 1, 1))
    K = np.tile(np.array([[500.0, 0, 320], [0, 500, 240], [0, 0, 1]])[None], (S, 1, 1))
    K[1:, 0, 0] = 600.0
    tracks = rng.random((S, P, 2)).astype(np.float32) * 100
    rec = batch_matrix_to_pycolmap(torch.from_numpy(pts), torch.from_numpy(ext), torch.from_numpy(K),
                                   torch.from_numpy(tracks), torch.from_numpy(masks), torch.tensor([640, 480]),
                                   shared_camera=True, camera_type="SIMPLE_RADIAL", extra_params=torch.zeros(S, 1))
    valid = np.nonzero(masks.sum(0) >= 2)[0]
    assert 7 not in valid and 8 not in valid
    assert rec.num_points3D() == len(valid) and rec.num_images() == S
    assert sorted(rec.point3D_ids()) == list(range(1, len(valid) + 1))
    k3 = int(np.nonzero(valid == 3)[0][0])
    assert rec.points3D[k3 + 1].track.length() == 0        # point 3 lies beyond max_points3D_val: no Point2D refers to it
    assert all((im.points2D._pid != k3 + 1).all() for im in rec.images.values())
    assert len(rec.cameras) == 1 and rec.cameras[0].params[0] == 500.0        # shared camera = frame 0's
    p, e, Kb, xp = pycolmap_to_batch_matrix(rec, device="cpu", camera_type="SIMPLE_RADIAL")
    assert p.shape == (len(valid), 3) and e.shape == (S, 3, 4) and Kb.shape == (S, 3, 3) and xp.shape == (S, 1)
    assert np.array_equal(p.numpy(), pts[valid])


@pytest.mark.skipif(not ref_harness.available(), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("shared", [False, True])
def test_runner_edits_roundtrip(tmp_path, shared):
    """The edits the runner applies after the solve (vggsfm/runners/runner.py:546-591, 1009-1054): original file names,
    cameras back at the original resolution, shifted 2D points, a deregistered image, trackless extra points -- read back
    with the reference's reader."""
    sc, rec, colors = _model(tmp_path, "SIMPLE_RADIAL", shared)
    names = [f"frame_{s:03d}.jpg" for s in range(5)]
    crop = np.zeros((1, 5, 8))
    crop[0, :, 0], crop[0, :, 1] = [1920, 1600, 1920, 1280, 1920], [1080, 1200, 1080, 960, 1080]       # real (w, h)
    crop[0, :, 4], crop[0, :, 5] = -3.0, -100.0                                                         # padded top-left
    from vggsfm_amd.runners import rename_colmap_recons_and_rescale_camera
    f_before = {c: rec.cameras[c].params[0] for c in rec.cameras}
    xy_before = {s: rec.images[s].points2D._xy.copy() for s in rec.images}
    rename_colmap_recons_and_rescale_camera(rec, names, crop, 1024, shift_point2d_to_original_res=True, shared_camera=shared)
    short = {p for p in range(40) if sc.mask[3, p] and sc.mask[:, p].sum() <= 2}   # these go with image 3 (COLMAP DeleteObservation)
    rec.deregister_image(3)
    rec.add_points3D(np.array([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]]), np.array([[9, 8, 7], [1, 2, 3]]))
    assert rec.num_points3D() == 42 - len(short)
    rec.write(str(tmp_path))
    ref_harness.install()
    from vggsfm.datasets.imc_helper import read_model
    cameras, images, points3D = read_model(str(tmp_path), ext=".bin")
    assert len(images) == 4 and 3 not in images and len(points3D) == 42 - len(short)
    assert sorted(im.name for im in images.values()) == [names[s] for s in (0, 1, 2, 4)]
    ncam = 1 if shared else 5
    assert len(cameras) == ncam
    for c in range(ncam):
        ratio = max(crop[0, c, 0], crop[0, c, 1]) / 1024.0
        np.testing.assert_allclose(cameras[c].params[0], ratio * f_before[c])
        assert (cameras[c].width, cameras[c].height) == (int(crop[0, c, 0]), int(crop[0, c, 1]))
        assert tuple(cameras[c].params[1:3]) == (crop[0, c, 0] // 2, crop[0, c, 1] // 2)
    for s in (0, 1, 2, 4):
        ratio = max(crop[0, 0 if shared else s, 0], crop[0, 0 if shared else s, 1]) / 1024.0
        np.testing.assert_allclose(images[s].xys, (xy_before[s] - np.array([3.0, 100.0])) * ratio)
        pids = np.nonzero(sc.mask[s])[0]
        expect = np.array([-1 if p in short else p + 1 for p in pids])
        assert np.array_equal(images[s].point3D_ids, expect)
    assert points3D[41].xyz.tolist() == [1.0, 2.0, 3.0] and points3D[42].rgb.tolist() == [1, 2, 3]
    assert len(points3D[41].image_ids) == 0
    # observations of the deregistered image are gone from the tracks
    for p in range(40):
        if p not in short:
            assert 3 not in points3D[p + 1].image_ids.tolist()
            assert len(points3D[p + 1].image_ids) == int(sc.mask[:, p].sum()) - int(sc.mask[3, p])
