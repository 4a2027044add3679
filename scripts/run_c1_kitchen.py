#!/usr/bin/env python
"""BASELINE configs[0] -- "examples/kitchen, 8 frames, 2048 query points x 3 query frames, plumbing only" -- as far as it
can run without the learned front end (no checkpoint, no network: SURVEY.md section 8d).

  --stage prepare   (build container, needs /root/reference): 8 of the 25 kitchen PNGs (1558 x 1039) are padded to a
                    square and resized to 1024 exactly as ``demo_loader.pad_and_resize_image(crop_longest=True)`` does
                    (vggsfm/datasets/demo_loader.py:437-484, crop parameters :398-434); what travels to the GPU box is
                    tests/golden/c1_kitchen_inputs.npz = file names, crop parameters and 64 x 64 thumbnails of the
                    padded images (colours only).
  --stage run       (GPU box): tracks / visibilities / predicted cameras of a synthetic 8-frame scene are injected at the
                    Triangulator boundary together with those images and crop parameters; the drop-in runs the whole
                    post-tracker path (two-view stage -> Triangulator -> frame filtering -> cameras and 2D points back
                    at the ORIGINAL 1558 x 1039 resolution -> ``reconstruction.write``), the COLMAP model is read back.
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FIXTURE = os.path.join(ROOT, "tests", "golden", "c1_kitchen_inputs.npz")
IMG = 1024


def prepare():
    from PIL import Image
    src = "/root/reference/examples/kitchen/images"
    names = sorted(os.listdir(src))[::3][:8]
    thumbs, crops = [], []
    for n in names:
        im = Image.open(os.path.join(src, n)).convert("RGB")
        w, h = im.size
        crop_dim = max(w, h)
        left, top = (w - crop_dim) // 2, (h - crop_dim) // 2
        bbox = np.array([left, top, left + crop_dim, top + crop_dim], dtype=np.float64)
        s = crop_dim / min(w, h)
        crop_width = 2 * s * (bbox[2] - bbox[0]) / crop_dim
        after = bbox / crop_dim * IMG
        crops.append([w, h, crop_width, s, after[0], after[1], after[2], after[3]])
        sq = Image.new("RGB", (crop_dim, crop_dim))
        sq.paste(im, (-left, -top))
        thumbs.append(np.asarray(sq.resize((64, 64), Image.BILINEAR), dtype=np.uint8))
    np.savez_compressed(FIXTURE, names=np.array(names), crop_params=np.array(crops, np.float32), thumbs=np.stack(thumbs))
    print("wrote", FIXTURE, names, np.array(crops)[0])


def run(out, model_dir=None):
    """-> the result dict (also printed / written to `out`); the COLMAP model goes to `model_dir`
    (tests/test_gpu_runner.py::test_c1_kitchen_plumbing calls this with a temporary directory)."""
    import torch

    from vggsfm_amd.pycolmap_compat import Reconstruction
    from vggsfm_amd.runners import GeometryConfig, GeometryRunner
    from vggsfm_amd.scene import make_scene, perturb_for_ba
    g = np.load(FIXTURE)
    names = [str(n) for n in g["names"]]
    S, N, dev = 8, 3 * 2048, "cuda"
    sc = make_scene(S, N, "SIMPLE_PINHOLE", shared_camera=False, seed=7, outlier_frac=0.03)
    ext0, K0, _, _ = perturb_for_ba(sc, seed=7, rot_deg=0.5, trans=0.02, focal_rel=0.02)
    D = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    images = D(g["thumbs"]).permute(0, 3, 1, 2).float().div(255.0)                    # (8,3,64,64)
    images = images.repeat_interleave(IMG // 64, dim=-1).repeat_interleave(IMG // 64, dim=-2)[None]
    crop = D(g["crop_params"])[None]
    # tracks inside the padded area are forced invisible by the runner (runner.py:453-463); here: the content rows only
    y0 = float(g["crop_params"][0, 5])
    vis = sc.vis * ((sc.tracks[..., 1] >= abs(y0)) & (sc.tracks[..., 1] <= IMG - abs(y0)))
    cams = types.SimpleNamespace(R=D(ext0[:, :, :3]).float(), T=D(ext0[:, :, 3]).float())
    fl = D(K0[:, 0, 0] / (IMG / 2.0)).float()
    cams.focal_length = torch.stack([fl, fl], -1)
    cfg = GeometryConfig(fmat_thres=4.0, max_ransac_iters=1024, lo_num=100, shift_point2d_to_original_res=True)
    torch.manual_seed(0)
    np.random.seed(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pred = GeometryRunner(cfg).sparse_reconstruct_from_tracks(cams, D(sc.tracks)[None], D(vis.astype(np.float32))[None],
                                                              D(sc.score)[None], images, crop_params=crop, image_paths=names)
    torch.cuda.synchronize()
    seconds = time.perf_counter() - t0
    rec = pred["reconstruction"]
    model_dir = model_dir or os.path.join(os.path.dirname(out) if out else "/tmp", "c1_kitchen_sparse")
    rec.write(model_dir)
    back = Reconstruction(model_dir)
    cam0 = back.cameras[0]
    res = dict(workload="BASELINE configs[0]: kitchen, 8 frames, 3 x 2048 injected tracks, SIMPLE_PINHOLE (plumbing: real images / "
                        "crop parameters, synthetic tracks and cameras at the Triangulator boundary)",
               seconds_post_tracker=seconds, valid_tracks=int(pred["valid_tracks"].sum()), points3D=back.num_points3D(),
               registered_images=back.num_reg_images(), image_names=[back.images[i].name for i in sorted(back.images)],
               camera0=dict(model=cam0.model, width=cam0.width, height=cam0.height, params=[float(p) for p in cam0.params]),
               mean_track_length=float(np.mean([back.points3D[p].track.length() for p in list(back.points3D)[:2000]])),
               model_bytes={f: os.path.getsize(os.path.join(model_dir, f)) for f in ("cameras.bin", "images.bin", "points3D.bin")})
    assert (cam0.width, cam0.height) == (1558, 1039) and res["image_names"] == names
    print(json.dumps(res))
    if out:
        with open(out, "w") as fh:
            json.dump(res, fh, indent=1)
    res["_reconstruction"], res["_read_back"], res["_predictions"] = rec, back, pred
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", choices=["prepare", "run"], required=True)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    prepare() if a.stage == "prepare" else run(a.out)
