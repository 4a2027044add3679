"""Edge cases of the C-ABI entries: empty inputs, single tracks, tracks without observations, invalid arguments."""
import ctypes

import numpy as np
import pytest
import torch

from vggsfm_amd import _lib
from vggsfm_amd import ba as BA
from vggsfm_amd.ba_options import BundleAdjustmentOptions
from vggsfm_amd.scene import make_scene, perturb_for_ba
from vggsfm_amd.utils import triangulation as T
from vggsfm_amd.utils import triangulation_helpers as H

pytestmark = pytest.mark.gpu


def D(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()


def test_empty_track_axis():
    sc = make_scene(6, 50, "SIMPLE_RADIAL", shared_camera=True, seed=1)
    ext, K, xp = D(sc.extrinsics), D(sc.intrinsics), D(sc.extra_params)
    tr0 = D(sc.tracks[:, :0])
    tn = H.cam_from_img(tr0, K, xp)
    assert tn.shape == (6, 0, 2)
    pts, num, msk = T.triangulate_tracks(ext, tn, track_vis=D(sc.vis[:, :0]), track_score=D(sc.score[:, :0]))
    assert pts.shape == (0, 3) and num.shape == (0,) and msk.shape == (0, 6)
    m, d = H.filter_all_points3D(pts, tr0, ext, K, xp, return_detail=True)
    assert m.shape == (0,) and d.shape == (6, 0)
    assert H.project_3D_points(pts, ext, K, xp).shape == (6, 0, 2)


def test_single_track_and_two_frames():
    sc = make_scene(2, 1, "SIMPLE_PINHOLE", seed=2, full_visibility=True, noise_px=0.0, outlier_frac=0.0)
    ext, K = D(sc.extrinsics), D(sc.intrinsics)
    tn = H.cam_from_img(D(sc.tracks), K)
    pts, num, msk = T.triangulate_tracks(ext, tn, track_vis=D(sc.vis), track_score=D(sc.score))
    assert int(num[0]) == 2 and bool(msk.all())
    np.testing.assert_allclose(pts.cpu().numpy(), sc.points3D, atol=1e-5)


def test_ba_with_unobserved_and_short_tracks():
    """Tracks with < 2 masked observations never enter the problem (tensor_to_pycolmap.py:62-68); a camera without
    observations stays put."""
    sc = make_scene(6, 120, "SIMPLE_PINHOLE", seed=3, full_visibility=True, outlier_frac=0.0)
    ext0, K0, _, pts0 = perturb_for_ba(sc, seed=3)
    masks = sc.mask.copy()
    masks[:, :10] = False                      # unobserved tracks
    masks[1:, 10:20] = False                   # single-observation tracks
    masks[5, :] = False                        # camera 5 sees nothing
    pts, ext, K, extra, sg = BA.bundle_adjustment(D(pts0), D(ext0), D(K0), D(sc.tracks), D(masks), None, None, False,
                                                  "SIMPLE_PINHOLE", BundleAdjustmentOptions())
    assert pts.shape[0] == 100 and sg["valid_idx"].tolist() == list(range(20, 120))
    # (poses go through the quaternion parametrisation: unchanged up to the round trip)
    np.testing.assert_allclose(ext[5].cpu().numpy(), ext0[5], atol=1e-14)
    assert float(K[5, 0, 0]) == K0[5, 0, 0]
    assert sg["final_cost"] < sg["initial_cost"]


def test_ba_with_no_valid_track_is_a_no_op():
    sc = make_scene(4, 30, "SIMPLE_PINHOLE", seed=4, full_visibility=True)
    ext0, K0, _, pts0 = perturb_for_ba(sc, seed=4)
    masks = np.zeros_like(sc.mask)
    pts, ext, K, extra, sg = BA.bundle_adjustment(D(pts0), D(ext0), D(K0), D(sc.tracks), D(masks), None, None, False,
                                                  "SIMPLE_PINHOLE", BundleAdjustmentOptions())
    assert pts.shape[0] == 0
    np.testing.assert_allclose(ext.cpu().numpy(), ext0, atol=1e-14)


def test_invalid_arguments_are_reported_not_executed():
    L = _lib.lib()
    buf = torch.zeros(64, dtype=torch.float64, device="cuda")
    p = _lib.ptr(buf)
    assert L.vgg_cholesky_solve(p, p, 0, p, None, _lib.stream_ptr()) != 0                    # n = 0
    assert L.vgg_triangulate_by_pair(p, p, 1, 4, p, _lib.stream_ptr()) != 0                  # a single frame
    assert L.vgg_project_points(p, -1, p, p, None, 0, 2, p, None, _lib.stream_ptr()) != 0    # negative count
    thr = ctypes.c_double(7.0)
    assert L.vgg_triangulate_tracks(p, p, p, p, 2, 4, 300, 50, ctypes.c_double(2.0), ctypes.c_double(1.5), p, p, p,
                                    ctypes.byref(thr), p, _lib.stream_ptr()) != 0            # more than 256 hypotheses
    with pytest.raises(ValueError):
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float64, device="cuda")
        BA.compile_problem(z(2, 3), z(1, 3, 4), z(1, 3, 3), z(1, 2, 2), torch.ones(1, 2, dtype=torch.bool, device="cuda"),
                           None, False, "OPENCV")
    with pytest.raises(RuntimeError):
        H.project_3D_points(torch.zeros(3, 3, dtype=torch.float64), torch.zeros(1, 3, 4, dtype=torch.float64),
                            torch.zeros(1, 3, 3, dtype=torch.float64))                       # CPU tensors: no fallback
