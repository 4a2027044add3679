"""Golden vectors for the VIDEO driver: the reference's own, UNMODIFIED ``VideoRunner.run`` loop
(vggsfm/runners/video_runner.py:156-191) with its ``move_window`` (:640-905, incl. the shrink / step-back rule :712-751),
``prepare_window_data`` (:1051-1187), ``align_next_window``, ``filter_points_and_compute_masks``,
``triangulate_window_points``, the window BA through ``BundleAdjustmentConfig`` / ``solve_bundle_adjustment`` and
``joint_BA`` (:494-541) -- run on the CPU in the build container with

  * ``pycolmap`` / ``pyceres`` = oracle/pycolmap_shim.py (the CPU oracle behind a pycolmap-shaped surface, cut line B2),
  * the two learned parts replaced by the seeded providers of oracle/video_world.py, bound to the names the reference's
    module calls (``average_camera_prediction``, ``predict_tracks``); the frame range a call refers to is read from the
    image tensor itself (pixel value = frame index), so the providers need no hidden state,
  * ``sparse_reconstruct`` (the initial window: tracker + Triangulator, pinned elsewhere) returning a perturbed
    ground-truth reconstruction, and the output tail (``dicts_to_output``, ``_update_points_color``) stubbed.

    python -m oracle.gen_golden_video            (needs /root/reference)

Writes tests/golden/video_<case>.npz: the world's seed + digest, the scripted occlusions, and after EVERY ``move_window`` /
``joint_BA`` call of the loop a snapshot of the runner's state: return value, extrinsics of every registered frame, the
point table (ids, xyz) and the observation table ((point, frame) pairs with their pixels -- as a sha256 except in the last
snapshot --, de-duplicated -- after a
step-back the reference appends a re-registered frame's carried-over points a second time to its ``visible_points`` list,
video_runner.py:454; the lists are treated as sets here), intrinsics and distortion.
``tests/test_gpu_video_golden.py`` drives ``vggsfm_amd.video.VideoGeometry`` through the same calls on the GPU: window
bounds and success flags equal, observation tables bit-exact, poses / points to 1e-4.
"""
import os
import sys
import types
import warnings
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pycolmap_shim, ref_harness  # noqa: E402
from oracle.gen_golden import _StableSort  # noqa: E402
from oracle.video_world import VideoWorld  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class _PromotingBmm:
    """The reference wraps its projections in ``autocast(dtype=torch.double)`` (triangulation_helpers.py:331): on the GPU
    that casts the operands of ``bmm`` to float64, which is what lets it multiply float64 extrinsics with the float32
    points / intrinsics its dicts hold after a joint BA (video_runner.py:517-532, 620).  On the CPU the CUDA autocast is
    inert and the same call raises; this context gives ``torch.bmm`` that promotion for mixed operands."""

    def __enter__(self):
        self._orig = torch.bmm

        def bmm(a, b, *args, **kw):
            if a.dtype != b.dtype:
                t = torch.promote_types(a.dtype, b.dtype)
                a, b = a.to(t), b.to(t)
            return self._orig(a, b, *args, **kw)

        torch.bmm = bmm
        return self

    def __exit__(self, *exc):
        torch.bmm = self._orig


CASES = {
    # name: world + loop parameters.  occlusions = {k-th existing-point tracking call: first window index without visibility}
    #   call 2 -> index 5 > 2: SHRINK the window to 4 frames;  call 4 -> index 2: STEP BACK, the loop retries (call 5, clean)
    "radial_t60": dict(T=60, N=6000, seed=11, n_new=400, init=16, window=8, joint_interval=3, max_query_pts=4096,
                       occlusions={2: 5, 4: 2}, camera_type="SIMPLE_RADIAL"),
    # the reference's own defaults (cfgs/video_demo.yaml:8-13: init 32 / window 16 / joint BA every 6 windows / 1024 query points)
    # on the OTHER camera branch it allows (SIMPLE_PINHOLE shared, video_runner.py:1019-1039; extra_params = zeros, :148-152):
    #   call 2 -> index 9 > 2: SHRINK the 16-frame window to 8;  call 5 -> index 2: STEP BACK, retry with doubled query budget.
    # `compact`: the (point, frame) tables of all snapshots but the last are kept as digests too (the file stays < 2 MB)
    "pinhole_t160": dict(T=160, N=16000, seed=23, n_new=400, init=32, window=16, joint_interval=6, max_query_pts=1024,
                         occlusions={2: 9, 5: 2}, camera_type="SIMPLE_PINHOLE", k1=0.0, compact=True),
}


def snapshot(runner, kind, ret):
    frames = sorted(f for f in runner.frame_dict if "extri" in runner.frame_dict[f])
    pids = sorted(runner.point_dict)
    obs = sorted({(p, f) for p in pids for f in runner.point_dict[p]["track"]})
    # the frame lists must describe the same set as the point tracks (they are what prepare_window_data reads)
    from_frames = {(p, f) for f in frames for p in runner.frame_dict[f].get("visible_points", [])}
    assert from_frames == set(obs), (kind, len(from_frames), len(obs))
    uv = np.array([runner.point_dict[p]["track"][f]["uv"].numpy() for p, f in obs], np.float32).reshape(-1, 2)
    return dict(kind=kind, ret=np.array([ret[0], ret[1], int(ret[2])], np.int64) if ret is not None else np.zeros(3, np.int64),
                frames=np.array(frames, np.int64),
                extri=np.stack([runner.frame_dict[f]["extri"].numpy() for f in frames]).astype(np.float64),
                pids=np.array(pids, np.int64),
                xyz=np.array([runner.point_dict[p]["xyz"].numpy() for p in pids], np.float64).reshape(-1, 3),
                obs_point=np.array([o[0] for o in obs], np.int32), obs_frame=np.array([o[1] for o in obs], np.int16), obs_uv=uv,
                intrinsics=runner.intrinsics.numpy().astype(np.float64),
                extra=runner.extra_params.numpy().astype(np.float64))


def run_case(name, p):
    sys.modules["pycolmap"] = pycolmap_shim
    sys.modules["pyceres"] = pycolmap_shim.pyceres
    ref_harness.install()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import vggsfm.runners.video_runner as VR
    VR.pycolmap, VR.pyceres = pycolmap_shim, pycolmap_shim.pyceres
    for m in ("vggsfm.utils.tensor_to_pycolmap",):
        mod = __import__(m, fromlist=["_"])
        mod.pycolmap = pycolmap_shim
    world = VideoWorld(p["T"], p["N"], p["seed"], n_new=p["n_new"], occlusions=p["occlusions"], k1=p.get("k1", 0.02))
    T = world.T

    def frames_of(images):                                   # images (..., C, H, W) with pixel value = frame index
        return images.reshape(-1, *images.shape[-3:])[:, 0, 0, 0].long().tolist()

    def fake_camera_prediction(camera_predictor, reshaped_image, batch_size, repeat_times=5, query_indices=None):
        fr = frames_of(reshaped_image)
        assert fr == list(range(fr[0], fr[-1] + 1)) and query_indices == [0, len(fr) // 2, len(fr) - 1]
        e = torch.from_numpy(world.camera_prior(fr[0], fr[-1] + 1))
        return types.SimpleNamespace(R=e[:, :, :3], T=e[:, :, 3])

    def fake_predict_tracks(query_method, max_query_pts, track_predictor, images, masks, fmaps_for_tracker,
                            query_frame_indexes, fine_tracking, bound_bboxes=None, query_points_dict=None,
                            max_points_num=163840):
        fr = frames_of(images)
        f0, f1 = fr[0], fr[-1] + 1
        assert fr == list(range(f0, f1))
        if query_points_dict is not None:                    # carried-over points (+ the reference's support points)
            assert list(query_points_dict) == [0] and list(query_frame_indexes) == [0]
            tr, vis = world.track_existing(f0, f1, query_points_dict[0][0].numpy())
            score = np.ones_like(vis)
        else:
            tr, vis, score = world.track_new(f0, f1, list(query_frame_indexes))
        return torch.from_numpy(tr)[None], torch.from_numpy(vis)[None], torch.from_numpy(score)[None]

    VR.average_camera_prediction = fake_camera_prediction
    VR.predict_tracks = fake_predict_tracks

    runner = object.__new__(VR.VideoRunner)
    runner.cfg = types.SimpleNamespace(camera_type=p["camera_type"], shared_camera=True, max_query_pts=p["max_query_pts"],
                                       query_frame_num=3, query_method="synthetic", fine_tracking=False,
                                       extra_pt_pixel_interval=-1, save_to_disk=False, dense_depth=False,
                                       make_reproj_video=False, viz_visualize=False, gr_visualize=False,
                                       shift_point2d_to_original_res=False)
    runner.device = "cpu"
    runner.remove_borders = 4
    runner.camera_predictor = None
    runner.track_predictor = types.SimpleNamespace(process_images_to_fmaps=lambda images: images[:, :, :1, :1, :1].clone())
    runner.point_dict, runner.frame_dict = {}, defaultdict(dict)
    runner.crop_params, runner.intrinsics = None, None

    init = world.initial_prediction(p["init"])
    K64 = torch.from_numpy(world.K)      # (float64, as pycolmap_to_batch_matrix returns it; float32 after the first joint BA)
    init_pred = {"pred_track": torch.from_numpy(init["tracks"]), "pred_vis": torch.from_numpy(init["vis"]),
                 "valid_2D_mask": torch.from_numpy(init["mask"]),
                 "valid_tracks": torch.ones(init["tracks"].shape[1], dtype=torch.bool),
                 "points3D": torch.from_numpy(init["points3D"]), "points3D_rgb": None,
                 "extrinsics_opencv": torch.from_numpy(init["extrinsics"]),
                 "intrinsics_opencv": K64[None].expand(p["init"], -1, -1).clone(),
                 "extra_params": (torch.full((p["init"], 1), world.k1, dtype=torch.float64)
                                  if p["camera_type"] == "SIMPLE_RADIAL" else None)}
    runner.sparse_reconstruct = lambda *a, **k: init_pred
    runner.dicts_to_output = lambda *a, **k: {}
    runner._update_points_color = lambda *a, **k: None

    snaps = []
    cls = VR.VideoRunner

    def move_window(start_idx, end_idx, window_size, **kw):
        ret = cls.move_window(runner, start_idx, end_idx, window_size, **kw)
        snaps.append(snapshot(runner, f"move_window({start_idx},{end_idx},{window_size})", ret))
        print("  move_window", (start_idx, end_idx, window_size), "->", ret[:3], "points", len(runner.point_dict))
        return ret

    def joint_BA(start_idx, end_idx, **kw):
        cls.joint_BA(runner, start_idx, end_idx, **kw)
        snaps.append(snapshot(runner, f"joint_BA({start_idx},{end_idx})", None))
        print("  joint_BA", (start_idx, end_idx), "points", len(runner.point_dict))

    runner.move_window, runner.joint_BA = move_window, joint_BA

    images = torch.arange(T, dtype=torch.float32).reshape(1, T, 1, 1, 1).expand(1, T, 3, 2, 2).clone()
    crop = torch.zeros(1, T, 8)
    pycolmap_shim.CALLS.clear()
    # (demo.py:34 seed_all_random_engines: the loop draws from Python's `random` -- prepare_window_data's cap, video_runner.py:1070)
    import random
    random.seed(0)
    np.random.seed(0)
    torch.manual_seed(0)
    with _StableSort(), _PromotingBmm(), warnings.catch_warnings(), torch.no_grad():
        warnings.simplefilter("ignore")
        runner.run(images, masks=None, crop_params=crop, image_paths=[f"f{t}.png" for t in range(T)], query_frame_num=3,
                   seq_name="synthetic", output_dir="/tmp/unused", init_window_size=p["init"], window_size=p["window"],
                   joint_BA_interval=p["joint_interval"])
    kinds = [s["kind"] for s in snaps]
    print(name, "calls:", kinds)
    print("  solver calls:", [(c[0], c[1]["num_iterations"]) for c in pycolmap_shim.CALLS if c[0] != "pose_refinement"])
    out = dict(T=np.int64(T), N=np.int64(p["N"]), seed=np.int64(p["seed"]), n_new=np.int64(p["n_new"]),
               init=np.int64(p["init"]), window=np.int64(p["window"]), joint_interval=np.int64(p["joint_interval"]),
               max_query_pts=np.int64(p["max_query_pts"]), camera_type=p["camera_type"], k1=np.float64(world.k1),
               occl_calls=np.array(sorted(p["occlusions"]), np.int64),
               occl_first_bad=np.array([p["occlusions"][k] for k in sorted(p["occlusions"])], np.int64),
               world_sha256=world.digest(), num_snapshots=np.int64(len(snaps)), kinds=np.array(kinds),
               provider_log=np.array([f"{k}:{a}:{b}:{d}" for k, a, b, d in world.log]))
    import hashlib
    compact = bool(p.get("compact"))
    for i, s in enumerate(snaps):
        last = i + 1 == len(snaps)
        for k, v in s.items():
            if k == "kind":
                continue
            if k == "obs_uv" and (compact or not last):     # pixels: a digest (bit-exact check); the first golden keeps the last array
                out[f"s{i}_obs_uv_sha256"] = hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()
            elif k in ("obs_point", "obs_frame") and compact:
                out[f"s{i}_{k}_sha256"] = hashlib.sha256(np.ascontiguousarray(v.astype(np.int64)).tobytes()).hexdigest()
                out[f"s{i}_{k}_count"] = np.int64(len(v))
            elif k == "pids" and compact:                    # (0 .. n-1 in every snapshot: asserted, only the count is kept)
                assert np.array_equal(v, np.arange(len(v)))
                out[f"s{i}_pids_count"] = np.int64(len(v))
            elif k == "xyz" and compact:                     # compared to 1e-4: float32 keeps 1e-7 of the scene's extent
                out[f"s{i}_xyz"] = v.astype(np.float32)
            else:
                out[f"s{i}_{k}"] = v
    path = os.path.join(OUT, f"video_{name}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, f"{os.path.getsize(path) / 1e6:.2f} MB")


def main():
    only = sys.argv[1:]
    for name, p in CASES.items():
        if not only or name in only:
            run_case(name, p)


if __name__ == "__main__":
    main()
