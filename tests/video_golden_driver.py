"""Shared by tests/test_video_golden_cpu.py (driver logic on the CPU oracle arithmetic) and tests/test_gpu_video_golden.py
(the device kernels): ``vggsfm_amd.video.VideoGeometry.run`` against tests/golden/video_<case>.npz -- the snapshots the
reference's UNMODIFIED ``VideoRunner.run`` loop left behind (oracle/gen_golden_video.py): every ``move_window`` /
``joint_BA`` call of the loop, in order, with window bounds, success flag, observation table, poses, points, camera."""
import hashlib
import os

import numpy as np
import torch

from oracle.video_world import VideoWorld
from vggsfm_amd import video as V

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def run_against_golden(case, device, pose_tol=1e-4):
    g = np.load(os.path.join(GOLD, f"video_{case}.npz"), allow_pickle=False)
    T, INIT, WS = int(g["T"]), int(g["init"]), int(g["window"])
    occl = {int(k): int(v) for k, v in zip(g["occl_calls"], g["occl_first_bad"])}
    k1 = float(g["k1"]) if "k1" in g else 0.02              # (the first golden predates the field: the world's default)
    world = VideoWorld(T, int(g["N"]), int(g["seed"]), n_new=int(g["n_new"]), occlusions=occl, k1=k1)
    assert world.digest() == str(g["world_sha256"]), "the synthetic world is not the one the golden was generated from"
    dev = torch.device(device)
    D = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def camera_prior(f0, f1):
        return D(world.camera_prior(f0, f1))

    def track_existing(f0, f1, uv):
        tr, vis = world.track_existing(f0, f1, uv.detach().cpu().numpy())
        return D(tr), D(vis)

    def track_new(f0, f1):
        ws = f1 - f0 - 1                                   # (query frames of triangulate_window_points, video_runner.py:1216)
        tr, vis, score = world.track_new(f0, f1, [ws // 2, ws])
        return D(tr), D(vis), D(score)

    init = world.initial_prediction(INIT)
    n0 = init["tracks"].shape[1]
    pred = {"extrinsics_opencv": D(init["extrinsics"]), "pred_track": D(init["tracks"]), "pred_vis": D(init["vis"]),
            "valid_2D_mask": D(init["mask"]), "valid_tracks": torch.ones(n0, dtype=torch.bool, device=dev),
            "points3D": D(init["points3D"]), "points3D_rgb": None}
    # (float64 camera, as pycolmap_to_batch_matrix hands it over; the joint BA leaves float32 -- video_runner.py:517-532)
    # (SIMPLE_PINHOLE: the reference carries extra_params = zeros(1, 1) through the loop, video_runner.py:148-152)
    radial = str(g["camera_type"]) == "SIMPLE_RADIAL"
    vg = V.VideoGeometry(D(world.K)[None], torch.full((1, 1), world.k1 if radial else 0.0, dtype=torch.float64, device=dev),
                         str(g["camera_type"]), max_query_pts=int(g["max_query_pts"]), device=dev)
    vg.add_initial_window(pred, 0, INIT)

    snaps = []
    mw, jb = vg.move_window, vg.joint_BA

    def snap(kind, ret):
        t = vg.table
        snaps.append(dict(kind=kind, ret=ret, frames=torch.nonzero(t.has_extri).squeeze(1).cpu().numpy(),
                          extri=t.extri[t.has_extri].cpu().numpy(), xyz=t.xyz.cpu().numpy(),
                          obs_point=t.obs_point.cpu().numpy(), obs_frame=t.obs_frame.cpu().numpy(),
                          obs_uv=t.obs_uv.cpu().numpy(), intrinsics=vg.intrinsics.double().cpu().numpy(),
                          extra=vg.extra_params.double().cpu().numpy()))

    def move_window(start_idx, end_idx, window_size, *a, **k):
        ret = mw(start_idx, end_idx, window_size, *a, **k)
        snap(f"move_window({start_idx},{end_idx},{window_size})", ret)
        return ret

    def joint_BA(start_idx, end_idx, **k):
        out = jb(start_idx, end_idx, **k)
        snap(f"joint_BA({start_idx},{end_idx})", None)
        return out

    vg.move_window, vg.joint_BA = move_window, joint_BA
    import random
    random.seed(0)                                          # (as the generator: the cap on carried-over points draws from it)
    np.random.seed(0)
    torch.manual_seed(0)
    vg.run(T, INIT, WS, camera_prior, track_existing, track_new, joint_BA_interval=int(g["joint_interval"]))

    kinds = [str(k) for k in g["kinds"]]
    assert [s["kind"] for s in snaps] == kinds, ([s["kind"] for s in snaps], kinds)
    assert [f"{k}:{a}:{b}:{d}" for k, a, b, d in world.log] == [str(x) for x in g["provider_log"]]
    worst = dict(extri=0.0, xyz=0.0)
    for i, s in enumerate(snaps):
        tag = f"snapshot {i} {s['kind']}"
        if s["ret"] is not None:
            assert [int(s["ret"][0]), int(s["ret"][1]), int(bool(s["ret"][2]))] == g[f"s{i}_ret"].tolist(), tag
        assert np.array_equal(s["frames"], g[f"s{i}_frames"]), tag
        if f"s{i}_pids" in g:
            assert len(s["xyz"]) == len(g[f"s{i}_pids"]) and np.array_equal(g[f"s{i}_pids"], np.arange(len(s["xyz"]))), tag
        else:
            assert len(s["xyz"]) == int(g[f"s{i}_pids_count"]), tag
        # the observation table: which (point, frame) pairs exist, and their pixels -- bit for bit
        for key in ("obs_point", "obs_frame"):
            if f"s{i}_{key}" in g:
                assert np.array_equal(s[key], g[f"s{i}_{key}"].astype(np.int64)), (tag, key)
            else:                                              # compact golden: count + digest of the int64 table
                assert len(s[key]) == int(g[f"s{i}_{key}_count"]), (tag, key, len(s[key]), int(g[f"s{i}_{key}_count"]))
                assert hashlib.sha256(np.ascontiguousarray(s[key].astype(np.int64)).tobytes()).hexdigest() == \
                    str(g[f"s{i}_{key}_sha256"]), (tag, key)
        if f"s{i}_obs_uv" in g:
            assert np.array_equal(s["obs_uv"], g[f"s{i}_obs_uv"]), tag
        else:
            assert hashlib.sha256(np.ascontiguousarray(s["obs_uv"]).tobytes()).hexdigest() == str(g[f"s{i}_obs_uv_sha256"]), tag
        d_e, d_x = np.abs(s["extri"] - g[f"s{i}_extri"]).max(), np.abs(s["xyz"] - g[f"s{i}_xyz"].astype(np.float64)).max()
        worst["extri"], worst["xyz"] = max(worst["extri"], d_e), max(worst["xyz"], d_x)
        assert d_e < pose_tol and d_x < pose_tol, (tag, d_e, d_x)
        np.testing.assert_allclose(s["intrinsics"], g[f"s{i}_intrinsics"], rtol=1e-6, err_msg=tag)
        np.testing.assert_allclose(s["extra"], g[f"s{i}_extra"], atol=1e-6, err_msg=tag)
    print(f"video golden {case}: {len(snaps)} calls ({sum('move' in k for k in kinds)} windows incl. a shrink and a step-back), "
          f"max |d pose| {worst['extri']:.2e}, max |d point| {worst['xyz']:.2e}")
    return worst
