"""Golden vectors for the END-TO-END drop-in: the reference's own ``vggsfm.models.Triangulator.forward`` run on
the CPU, with ``oracle/pycolmap_shim.py`` standing in for pycolmap (cut line B2).

Run in the build container only (needs /root/reference):
    python -m oracle.gen_golden_triangulator
Writes tests/golden/triangulator_<case>.npz = the inputs handed to ``forward`` and its outputs
(extrinsics, intrinsics, extra_params, points3D, valid_frame_mask, valid_2D_mask, valid_tracks).
What these vectors pin is the reference's DRIVER (mask bookkeeping, thresholds schedule, problem construction,
normalisation, read-back); the solver behind the shim is the oracle restatement (parity with pycolmap 3.10
itself stays unpinned, oracle/ba_oracle.h).  ``torch.manual_seed(0)`` right before ``forward``; ``torch.sort`` is
forced stable as in oracle/gen_golden.py.
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pycolmap_shim, ref_harness  # noqa: E402
from oracle.gen_golden import _StableSort  # noqa: E402
from vggsfm_amd.scene import make_scene, perturb_for_ba  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

CASES = {
    # name: (S, N, camera_type, shared_camera, seed, kwargs of forward)
    "pinhole_s10": (10, 500, "SIMPLE_PINHOLE", False, 31, dict(BA_iters=2, robust_refine=2)),
    "radial_shared_s12": (12, 600, "SIMPLE_RADIAL", True, 32, dict(BA_iters=2, robust_refine=2)),
    "pinhole_shared_s8": (8, 400, "SIMPLE_PINHOLE", True, 33, dict(BA_iters=1, robust_refine=1)),
    # S >= 24: C(S,2) > 256, so every triangulate_tracks call draws one torch.randperm from the global CPU RNG
    # (vggsfm/utils/triangulation.py:804-813); the drop-in must consume the same draws in the same order
    "radial_s26_randperm": (26, 400, "SIMPLE_RADIAL", False, 34, dict(BA_iters=2, robust_refine=1)),
    # BASELINE configs[1] at full size (50 frames x 20k tracks, per-frame SIMPLE_PINHOLE): ~40 min of reference CPU time.
    # Stored COMPACT: the inputs are regenerated from the seed by inputs() (a sha256 of them is kept), only outputs are saved.
    "pinhole_s50_c2": (50, 20000, "SIMPLE_PINHOLE", False, 35, dict(BA_iters=1, robust_refine=1)),
    # the camera model and the view count of BASELINE configs[2] (200 frames, shared SIMPLE_RADIAL) at 3000 tracks: the
    # 200-view regime of the triangulation kernel, the shared-intrinsics BA and the undistortion, end to end (compact)
    "radial_shared_s200": (200, 3000, "SIMPLE_RADIAL", True, 36, dict(BA_iters=1, robust_refine=1)),
    # the camera model of BASELINE configs[3] (per-frame SIMPLE_RADIAL: 8 x 8 camera blocks, 128 x 128 Schur tiles) at 120
    # frames x 2000 tracks (compact)
    "radial_s120": (120, 2000, "SIMPLE_RADIAL", False, 37, dict(BA_iters=1, robust_refine=1)),
    # the `force_estimate` branch of refine_pose (triangulation.py:406-432): frames 6 and 8 see fewer than 100 tracks, so the
    # last robust-refine round re-estimates them with absolute_pose_estimation -- the shim's deterministic restatement of the
    # device pipeline (oracle/pycolmap_shim.py ESTIMATION), its uniform draws recorded per frame for the GPU test to replay
    "pinhole_s10_estimate": (10, 600, "SIMPLE_PINHOLE", False, 38, dict(BA_iters=1, robust_refine=2)),
}
ESTIMATE_CASES = {"pinhole_s10_estimate": (6, 8)}       # frames whose visibility is cut down to ~80 tracks
COMPACT = {"pinhole_s50_c2", "radial_shared_s200", "radial_s120"}


def input_digest(inp):
    import hashlib
    h = hashlib.sha256()
    for k in sorted(inp):
        h.update(np.ascontiguousarray(inp[k]).tobytes())
    return h.hexdigest()


def inputs(S, N, cam, shared, seed, W=1024, sparse_frames=()):
    sc = make_scene(S, N, cam, shared_camera=shared, seed=seed, outlier_frac=0.03, full_visibility=bool(sparse_frames))
    for k, f in enumerate(sparse_frames):                 # only every 7th / 8th track stays visible in these frames
        keep = np.zeros(N, bool)
        keep[k::7 + k] = True
        sc.vis[f, ~keep] = 0.0
        sc.mask[f, ~keep] = False
    ext0, K0, _, _ = perturb_for_ba(sc, seed=seed, rot_deg=0.5, trans=0.02, focal_rel=0.02)
    rng = np.random.default_rng(seed)
    images = rng.random((1, S, 3, 32, 32), dtype=np.float32)      # colours only; resized to (W,W) by the consumer
    return dict(R=ext0[:, :, :3].astype(np.float32), T=ext0[:, :, 3].astype(np.float32),
                focal_ndc=(K0[:, 0, 0] / (W / 2.0)).astype(np.float32), tracks=sc.tracks.astype(np.float32),
                vis=sc.vis.astype(np.float32), score=sc.score.astype(np.float32),
                fmat_inlier=(sc.mask[1:] & sc.mask[0:1]), images_small=images, W=np.int64(W))


def expand_images(images_small, W):
    """(1,S,3,h,w) -> (1,S,3,W,W) by nearest-neighbour repetition (exactly reproducible on any device)."""
    t = torch.from_numpy(images_small)
    r = W // t.shape[-1]
    return t.repeat_interleave(r, dim=-1).repeat_interleave(r, dim=-2)


def main():
    sys.modules["pycolmap"] = pycolmap_shim
    ref_harness.install()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from vggsfm.models.triangulator import Triangulator
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]
    for name, (S, N, cam, shared, seed, kw) in CASES.items():
        if only and name not in only:
            continue
        inp = inputs(S, N, cam, shared, seed, sparse_frames=ESTIMATE_CASES.get(name, ()))
        W = int(inp["W"])
        pycolmap_shim.ESTIMATION.update(enabled=name in ESTIMATE_CASES, rng=np.random.default_rng(1000 + seed), log=[])
        cams = types.SimpleNamespace(R=torch.from_numpy(inp["R"]), T=torch.from_numpy(inp["T"]),
                                     focal_length=torch.from_numpy(np.stack([inp["focal_ndc"]] * 2, -1)))
        images = expand_images(inp["images_small"], W)
        prelim = {"fmat_inlier_mask": torch.from_numpy(inp["fmat_inlier"])[None]}
        pycolmap_shim.CALLS.clear()
        torch.manual_seed(0)
        with _StableSort(), warnings.catch_warnings(), torch.no_grad():
            warnings.simplefilter("ignore")
            out = Triangulator()(cams, torch.from_numpy(inp["tracks"])[None], torch.from_numpy(inp["vis"])[None], images,
                                 prelim, pred_score=torch.from_numpy(inp["score"])[None], shared_camera=shared,
                                 camera_type=cam, **kw)
        ext, K, extra, pts, rgb, rec, vframes, v2d, vtracks = out
        print(name, "valid tracks", int(vtracks.sum()), "of", N, "solver calls", len(pycolmap_shim.CALLS),
              [c[1]["num_iterations"] for c in pycolmap_shim.CALLS if c[0] == "bundle_adjustment"])
        stored = dict(S=np.int64(S), N=np.int64(N), seed=np.int64(seed), input_sha256=input_digest(inp), W=inp["W"]) \
            if name in COMPACT else dict(inp)
        if name in ESTIMATE_CASES:
            log = pycolmap_shim.ESTIMATION["log"]
            frames = [l[0] for l in log]
            print("   absolute_pose_estimation calls for frames", frames, "candidates", [l[1] for l in log])
            assert len(set(frames)) == len(frames) >= 1, "one estimate per frame expected (no retry with all points)"
            stored["est_frames"] = np.array(frames, np.int64)
            stored["est_candidates"] = np.array([l[1] for l in log], np.int64)
            stored["est_uniforms"] = np.stack([l[2] for l in log])
        pycolmap_shim.ESTIMATION["enabled"] = False
        np.savez_compressed(
            os.path.join(OUT, f"triangulator_{name}.npz"), camera_type=cam, shared=shared,
            kw_keys=np.array(list(kw.keys())), kw_vals=np.array(list(kw.values())), **stored,
            out_extrinsics=ext.numpy(), out_intrinsics=K.numpy(),
            out_extra=np.zeros((0,)) if extra is None else extra.numpy(), out_points3D=pts.numpy(),
            out_rgb=rgb.numpy(), out_valid_frames=vframes.numpy(), out_valid_2D=v2d.numpy(),
            out_valid_tracks=vtracks.numpy(), rec_num_points3D=np.int64(rec.num_points3D()))


if __name__ == "__main__":
    main()
