"""CPU: small host-side mirrors checked against the REFERENCE's own functions, imported through the stub harness when
/root/reference exists (this container; skipped on the GPU box): window / grid helpers of the runner
(vggsfm/utils/utils.py:773-839) and the similarity alignment of the video path (vggsfm/utils/align.py:145-252)."""
import numpy as np
import pytest
import torch

from oracle import ref_harness
from vggsfm_amd import video as V
from vggsfm_amd.runners import generate_grid_samples, sample_subrange

pytestmark = pytest.mark.skipif(not ref_harness.available(), reason="reference tree not present (GPU box)")


def test_sample_subrange_and_grid_match_reference():
    ref_harness.install()
    from vggsfm.utils.utils import generate_grid_samples as ref_grid
    from vggsfm.utils.utils import sample_subrange as ref_sub
    for N in range(1, 40):
        for L in range(1, 45):
            for idx in range(N):
                assert sample_subrange(N, idx, L) == ref_sub(N, idx, L), (N, idx, L)
    for rect in ([[3.0, 7.0, 515.0, 300.0]], [[0.0, 0.0, 1024.0, 1024.0]], [[100.5, 20.25, 130.0, 700.0]]):
        r = torch.tensor(rect)
        for kw in (dict(N=500), dict(N=77), dict(N=2048), dict(pixel_interval=16), dict(pixel_interval=100), dict(pixel_interval=7)):
            assert torch.equal(generate_grid_samples(r, **kw), ref_grid(r, **kw)), (rect, kw)


def _rand_rot(n, g):
    q = torch.randn(n, 4, generator=g, dtype=torch.float64)
    w, x, y, z = (q / q.norm(dim=1, keepdim=True)).unbind(1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w),
                        1 - 2 * (x * x + z * z), 2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w),
                        1 - 2 * (x * x + y * y)], 1).reshape(n, 3, 3)


def test_camera_alignment_matches_reference():
    """Own derivation (polar factor + one-dimensional least squares, video.py) against the reference's functions: the same
    similarity to rounding (float64)."""
    ref_harness.install()
    from vggsfm.utils.align import align_camera_extrinsics as ref_align
    from vggsfm.utils.align import apply_transformation as ref_apply
    g = torch.Generator().manual_seed(0)
    for n in (2, 9, 17, 33):
        src = torch.cat([_rand_rot(n, g), torch.randn(n, 3, 1, generator=g, dtype=torch.float64)], -1)
        tgt = torch.cat([_rand_rot(n, g), torch.randn(n, 3, 1, generator=g, dtype=torch.float64)], -1)
        a0, a1 = ref_align(src, tgt), V.align_camera_extrinsics(src, tgt)
        for x, y in zip(a0, a1):
            torch.testing.assert_close(torch.as_tensor(y, dtype=torch.float64), torch.as_tensor(x, dtype=torch.float64), rtol=1e-12, atol=1e-13)
            assert torch.as_tensor(x).shape == torch.as_tensor(y).shape
        torch.testing.assert_close(V.apply_transformation(src, *a1), ref_apply(src, *a0), rtol=1e-12, atol=1e-13)
        for x, y in zip(ref_apply(src, *a0, return_extri=False), V.apply_transformation(src, *a1, return_extri=False)):
            torch.testing.assert_close(y, x, rtol=1e-12, atol=1e-13)
        # without scale estimation, and a single camera (the reference's n == 1 branch)
        b0, b1 = ref_align(src[:1], tgt[:1]), V.align_camera_extrinsics(src[:1], tgt[:1])
        torch.testing.assert_close(V.apply_transformation(src[:1], *b1), ref_apply(src[:1], *b0), rtol=1e-12, atol=1e-13)
        c0, c1 = ref_align(src, tgt, estimate_scale=False), V.align_camera_extrinsics(src, tgt, estimate_scale=False)
        torch.testing.assert_close(V.apply_transformation(src, *c1), ref_apply(src, *c0), rtol=1e-12, atol=1e-13)


def test_get_EFP_and_sample_features4d_match_reference():
    """vggsfm/models/utils.py:38-72 (cameras -> [R t], K with the clamped one-dof focal) and :415-447 (bilinear colour
    lookup) -- the two edge functions of the Triangulator drop-in."""
    import types

    ref_harness.install()
    from vggsfm.models.utils import get_EFP as ref_efp
    from vggsfm.models.utils import sample_features4d as ref_sample
    from vggsfm_amd.models.utils import get_EFP, sample_features4d
    g = torch.Generator().manual_seed(1)
    for S, size in ((7, (1024.0, 768.0)), (3, (512.0, 512.0))):
        cams = types.SimpleNamespace(R=torch.randn(S, 3, 3, generator=g), T=torch.randn(S, 3, generator=g),
                                     focal_length=torch.rand(S, 2, generator=g) * 12 + 0.01)      # some hit the clamp
        for default_focal in (False, True):
            a = ref_efp(cams, torch.tensor(size), 1, S, default_focal=default_focal)
            b = get_EFP(cams, torch.tensor(size), 1, S, default_focal=default_focal)
            for x, y in zip(a, b):
                assert x.shape == y.shape and x.dtype == y.dtype
                torch.testing.assert_close(y, x, rtol=1e-6, atol=1e-6)        # (float32 cameras: order of the scalings)
    img = torch.rand(2, 3, 40, 50, generator=g)
    co = torch.rand(2, 30, 2, generator=g) * torch.tensor([49.0, 39.0])
    assert torch.equal(ref_sample(img, co), sample_features4d(img, co))


@pytest.mark.skipif(not ref_harness.available(), reason="reference tree not present (GPU box)")
def test_average_camera_prediction_matches_reference():
    """vggsfm_amd.utils.utils.average_camera_prediction (torch ops, scipy's quaternion conventions restated) against the
    reference's own function (vggsfm/utils/utils.py:25-164, scipy on the host) with the same fake predictor."""
    import types
    import warnings
    ref_harness.install()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from vggsfm.utils import utils as RU
    from scipy.spatial.transform import Rotation
    from vggsfm_amd.utils import utils as U
    S = 9
    rng = np.random.default_rng(4)

    def predictor(images, batch_size=1):
        # poses that depend on the ORDER of the frames (as a learned predictor's do): frame ids are encoded in the images
        ids = images[:, 0, 0, 0].round().long()
        rot = Rotation.from_rotvec(0.35 * np.stack([np.sin(ids.numpy() * 1.3), np.cos(ids.numpy() * 0.7), 0.1 * np.arange(len(ids))], 1))
        R = torch.from_numpy(rot.as_matrix()).float()
        T = torch.stack([ids.float() * 0.1, torch.arange(len(ids)).float() * 0.05, torch.ones(len(ids))], 1)
        f = torch.stack([1.5 + 0.01 * ids.float(), 1.5 + 0.02 * torch.arange(len(ids)).float()], 1)
        return {"pred_cameras": types.SimpleNamespace(R=R, T=T, focal_length=f)}

    images = torch.arange(S).float()[:, None, None, None].expand(S, 3, 4, 4).contiguous()
    for q in ([0, S // 2, S - 1], [0, 3], None):
        import random
        random.seed(7)
        a = RU.average_camera_prediction(predictor, images, 1, query_indices=None if q is None else list(q))
        random.seed(7)
        b = U.average_camera_prediction(predictor, images, 1, query_indices=None if q is None else list(q))
        np.testing.assert_allclose(b.R.numpy(), np.asarray(a.R), rtol=0, atol=1e-12)
        np.testing.assert_allclose(b.T.numpy(), np.asarray(a.T), rtol=0, atol=1e-7)
        np.testing.assert_allclose(b.focal_length.numpy(), np.asarray(a.focal_length), rtol=0, atol=1e-7)
    # the quaternion restatement itself, sign included, on rotations of every branch
    M = Rotation.random(500, random_state=1).as_matrix()
    q = U.matrix_to_quaternion_scipy(torch.from_numpy(M)).numpy()
    np.testing.assert_allclose(q, Rotation.from_matrix(M).as_quat(), rtol=0, atol=1e-14)
    np.testing.assert_allclose(U.quaternion_to_matrix_scipy(torch.from_numpy(q)).numpy(), M, rtol=0, atol=1e-14)


def test_small_driver_helpers_match_reference():
    """find_best_initial_pair (triangulator.py:442-476), get_valid_frame_mask (triangulation.py:1222-1242),
    create_intri_matrix and generate_combinations (triangulation_helpers.py:590-645): the host-side mirrors against the
    reference's own functions on random inputs, bit for bit."""
    ref_harness.install()
    from vggsfm.models.triangulator import find_best_initial_pair as ref_pair
    from vggsfm.utils.triangulation import get_valid_frame_mask as ref_valid
    from vggsfm.utils.triangulation_helpers import create_intri_matrix as ref_K
    from vggsfm.utils.triangulation_helpers import generate_combinations as ref_comb
    from vggsfm_amd.models.triangulator import find_best_initial_pair
    from vggsfm_amd.utils.triangulation import get_valid_frame_mask
    from vggsfm_amd.utils.triangulation_helpers import create_intri_matrix, generate_combinations
    g = torch.Generator().manual_seed(0)
    for trial in range(30):
        S, N = int(torch.randint(3, 12, (1,), generator=g)), int(torch.randint(50, 600, (1,), generator=g))
        dens = float(torch.rand(1, generator=g))
        geo = torch.rand(S - 1, N, generator=g) < dens
        che = torch.rand(S - 1, N, generator=g) < 0.9
        ang = torch.rand(S - 1, N, generator=g, dtype=torch.float64) * float(torch.randint(1, 40, (1,), generator=g))
        for thr in (16, 8, 3, 1):
            a, ta = find_best_initial_pair(geo, che, ang, thr)
            b, tb = ref_pair(geo, che, ang, thr)
            assert torch.equal(a, b) and ta == tb
        K = torch.zeros(S, 3, 3, dtype=torch.float64)
        K[:, 0, 0] = torch.rand(S, generator=g, dtype=torch.float64) * 40000 - 2000
        ext = torch.randn(S, 3, 4, generator=g, dtype=torch.float64) * 20
        extra = torch.randn(S, 1, generator=g, dtype=torch.float64)
        for ex in (None, extra, extra[:, 0]):
            assert torch.equal(get_valid_frame_mask(K, ext, ex, 1024.0), ref_valid(K, ext, ex, 1024.0))
        f, pp = torch.rand(S, 2, generator=g), torch.rand(S, 2, generator=g)
        assert torch.equal(create_intri_matrix(f, pp), ref_K(f, pp))
        assert torch.equal(create_intri_matrix(f[None], pp[None]), ref_K(f[None], pp[None]))
    for S in (2, 3, 7, 24, 50):
        assert np.array_equal(generate_combinations(S), ref_comb(S))


def test_order_and_se3_helpers_match_reference():
    """calculate_index_mappings / switch_tensor_order (utils/utils.py:167-187), closed_form_inverse_OpenCV
    (utils/metric.py:233-268) against the reference's own functions."""
    ref_harness.install()
    from vggsfm.utils.metric import closed_form_inverse_OpenCV as ref_inv
    from vggsfm.utils.utils import calculate_index_mappings as ref_map
    from vggsfm.utils.utils import switch_tensor_order as ref_switch
    from vggsfm_amd.utils.utils import calculate_index_mappings, closed_form_inverse_OpenCV, switch_tensor_order
    g = torch.Generator().manual_seed(1)
    for S in (1, 2, 5, 9):
        for q in range(S):
            order = calculate_index_mappings(q, S)
            assert torch.equal(order, ref_map(q, S))
            ts = [torch.randn(2, S, 3, generator=g), None, torch.randn(2, S, generator=g)]
            for a, b in zip(switch_tensor_order(ts, order), ref_switch(ts, order)):
                assert (a is None and b is None) or torch.equal(a, b)
    se3 = torch.eye(4, dtype=torch.float64)[None].repeat(6, 1, 1)
    se3[:, :3, :3] = _rand_rot(6, g)
    se3[:, :3, 3] = torch.randn(6, 3, generator=g, dtype=torch.float64)
    assert torch.equal(closed_form_inverse_OpenCV(se3), ref_inv(se3))
