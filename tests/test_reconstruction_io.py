"""COLMAP binary model writer (vggsfm_amd/reconstruction.py) vs the reference's own reader
(vggsfm/datasets/imc_helper.py, imported through the stub harness when /root/reference exists) and vs a
minimal independent parser of the documented layout.  CPU only."""
import os
import struct

import numpy as np
import pytest

from oracle import ref_harness
from vggsfm_amd.reconstruction import Reconstruction
from vggsfm_amd.scene import make_scene


def _model(tmp_path, cam, shared):
    sc = make_scene(5, 40, cam, shared_camera=shared, seed=3)
    colors = (np.arange(40 * 3) % 255).reshape(40, 3).astype(np.uint8)
    rec = Reconstruction(sc.points3D, sc.extrinsics, sc.intrinsics, sc.tracks, sc.mask, [1024, 1024], shared, cam,
                         sc.extra_params, colors)
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
