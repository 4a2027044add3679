"""End to end over BASELINE configs[4]'s path at test size: the geometric half of ``VideoRunner.run``
(vggsfm/runners/video_runner.py:64-247,640-905) -- ``vggsfm_amd.video.VideoGeometry`` -- slides over a synthetic
sequence with a simulated camera predictor (ground truth + noise, in an arbitrary similarity gauge) and a simulated
tracker (ground-truth projections + pixel noise, visibility = inside the image).  The learned parts are out of scope;
what is checked is that register -> triangulate -> local BA -> table update -> joint BA, all on the device, recovers
the trajectory of every frame up to the similarity gauge."""
import numpy as np
import pytest
import torch

from vggsfm_amd import video as V
from vggsfm_amd.scene import project

pytestmark = pytest.mark.gpu


def _rodrigues(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def _sequence(T, N, seed, W=1024.0, f=1000.0, k1=0.02):
    rng = np.random.default_rng(seed)
    ext = np.zeros((T, 3, 4))
    for t in range(T):
        R = _rodrigues(np.array([0.02 * np.sin(0.3 * t), 0.06 * np.sin(0.15 * t), 0.01 * np.cos(0.2 * t)]))
        c = np.array([0.11 * t, 0.05 * np.sin(0.25 * t), 0.03 * np.cos(0.2 * t)])
        ext[t, :, :3], ext[t, :, 3] = R, -R @ c
    pts = np.stack([rng.uniform(-2.5, 0.11 * T + 2.5, N), rng.uniform(-1.5, 1.5, N), rng.uniform(3.0, 6.0, N)], 1)
    K = np.tile(np.array([[f, 0, W / 2], [0, f, W / 2], [0, 0, 1.0]]), (T, 1, 1))
    extra = np.full((T, 1), k1)
    uv, cam = project(pts, ext, K, extra)                       # (pixels, depth)
    vis = (uv[..., 0] > 8) & (uv[..., 0] < W - 8) & (uv[..., 1] > 8) & (uv[..., 1] < W - 8) & (cam > 0.5)
    tracks = (uv + rng.normal(0, 0.3, uv.shape)).astype(np.float32)
    return ext, K, extra, pts, tracks, vis


@pytest.mark.parametrize("use_pnp", [False, True])
def test_video_geometry_recovers_trajectory(use_pnp):
    T, N, INIT, WS = 56, 6000, 16, 8
    ext, K, extra, pts, tracks, vis = _sequence(T, N, seed=3)
    dev = "cuda"
    tr_all, vis_all = torch.from_numpy(tracks).to(dev), torch.from_numpy(vis).to(dev)
    rng = np.random.default_rng(9)
    gen = torch.Generator(device=dev).manual_seed(4)

    # simulated camera predictor: noisy ground truth in an arbitrary similarity gauge, a fresh one per call
    def camera_prior(f0, f1):
        e = ext[f0:f1].copy()
        Rg, s, tg = _rodrigues(rng.normal(0, 0.5, 3)), float(rng.uniform(0.5, 2.0)), rng.normal(0, 1.0, 3)
        out = np.zeros_like(e)
        for i in range(len(e)):
            Rn = _rodrigues(rng.normal(0, 0.01, 3)) @ e[i, :, :3]
            tn = e[i, :, 3] + rng.normal(0, 0.02, 3)
            out[i, :, :3] = Rn @ Rg.T                                  # X_world' = s Rg X + tg
            out[i, :, 3] = s * tn - out[i, :, :3] @ tg
        return torch.from_numpy(out).to(dev)

    # simulated tracker: the query pixel identifies the ground-truth point (exact match in the query frame)
    def track_existing(f0, f1, uv):
        d = torch.cdist(uv.double(), tr_all[f0].double())
        dist, idx = d.min(dim=1)
        assert float(dist.max()) < 1e-3                                # (cdist rounds; the match itself is exact)
        return tr_all[f0:f1][:, idx], vis_all[f0:f1][:, idx].float()

    def track_new(f0, f1):
        cand = torch.nonzero(vis_all[f0]).squeeze(1)
        sel = cand[torch.randperm(cand.numel(), device=dev, generator=gen)[:900]]
        v = vis_all[f0:f1][:, sel].float()
        return tr_all[f0:f1][:, sel], v, torch.ones_like(v)

    # initial window: a perturbed reconstruction of frames 0..INIT (what sparse_reconstruct hands over)
    seen = vis[:INIT].sum(0) >= 4
    idx0 = np.nonzero(seen)[0][:1500]
    pred = {"extrinsics_opencv": torch.from_numpy(ext[:INIT]).to(dev),
            "pred_track": tr_all[:INIT][:, idx0], "pred_vis": vis_all[:INIT][:, idx0].float(),
            "valid_2D_mask": vis_all[:INIT][:, idx0], "valid_tracks": torch.ones(len(idx0), dtype=torch.bool, device=dev),
            "points3D": torch.from_numpy(pts[idx0] + rng.normal(0, 0.01, (len(idx0), 3))).to(dev), "points3D_rgb": None}
    vg = V.VideoGeometry(torch.from_numpy(K[:1]).to(dev), torch.from_numpy(extra[:1]).to(dev), "SIMPLE_RADIAL",
                         max_query_pts=1200, device=dev, generator=gen)
    vg.add_initial_window(pred, 0, INIT)
    table = vg.run(T, INIT, WS, camera_prior, track_existing, track_new, joint_BA_interval=3, use_pnp=use_pnp)

    assert bool(table.has_extri[:T].all()) and table.extri.shape[0] == T
    assert table.num_points > 2000 and table.num_observations > 20 * table.num_points // 4
    # similarity gauge: align the estimate onto the ground truth, then compare
    est = table.extri[:T]
    gt = torch.from_numpy(ext).to(dev)
    aR, aT, a_s = V.align_camera_extrinsics(est, gt)
    al = V.apply_transformation(est, aR, aT, a_s)
    Rrel = torch.bmm(al[:, :, :3], gt[:, :, :3].transpose(1, 2))
    ang = torch.acos(((Rrel.diagonal(dim1=1, dim2=2).sum(1) - 1) / 2).clamp(-1, 1))
    c_est = -torch.bmm(al[:, :, :3].transpose(1, 2), al[:, :, 3:])[..., 0]
    c_gt = -torch.bmm(gt[:, :, :3].transpose(1, 2), gt[:, :, 3:])[..., 0]
    path = float((c_gt[-1] - c_gt[0]).norm())
    print(f"max rotation error {float(ang.max()):.2e} rad, max centre error {float((c_est - c_gt).norm(dim=1).max()):.3e} "
          f"over a {path:.2f} path, focal {float(vg.intrinsics[0, 0, 0]):.2f}, k {float(vg.extra_params[0, 0]):.4f}, "
          f"{table.num_points} points / {table.num_observations} observations")
    assert float(ang.max()) < 5e-3
    assert float((c_est - c_gt).norm(dim=1).max()) < 0.01 * path
    assert abs(float(vg.intrinsics[0, 0, 0]) / 1000.0 - 1) < 5e-3 and abs(float(vg.extra_params[0, 0]) - 0.02) < 5e-3
    # every stored observation reprojects within the joint BA's 2 px filter
    xyz, e, trk, msk, _ = table.window_tensors(0, T)
    from vggsfm_amd.utils.triangulation_helpers import project_3D_points
    uv = project_3D_points(xyz, e, vg.intrinsics.double().expand(T, -1, -1), vg.extra_params.double().expand(T, -1))
    err = (uv - trk.double()).norm(dim=-1)[msk]
    assert float(err.max()) < 2.0 + 1e-6 and float(err.median()) < 0.6
