"""Secondary measurements (not the bench.py metric): LO-RANSAC triangulation, filter, undistortion and projection
at BASELINE configs[1] / configs[2] sizes.  Prints one JSON line per configuration."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vggsfm_amd.scene import make_scene  # noqa: E402
from vggsfm_amd.utils import triangulation as T  # noqa: E402
from vggsfm_amd.utils import triangulation_helpers as H  # noqa: E402


def D(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, out


CONFIGS = {"c2": (50, 20000, "SIMPLE_PINHOLE", False), "c3": (200, 100000, "SIMPLE_RADIAL", True)}
for name, (S, N, cam, shared) in CONFIGS.items():
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    sc = make_scene(S, N, cam, shared_camera=shared, seed=0)
    ext, K, xp = D(sc.extrinsics), D(sc.intrinsics), D(sc.extra_params)
    tracks, vis, score = D(sc.tracks), D(sc.vis), D(sc.score)
    t_und, tn = timed(lambda: H.cam_from_img(tracks, K, xp))
    torch.manual_seed(0)
    t_tri, (p3, num, msk) = timed(lambda: T.triangulate_tracks(ext, tn, track_vis=vis, track_score=score), reps=2)
    t_flt, _ = timed(lambda: H.filter_all_points3D(p3, tracks, ext, K, xp, check_triangle=True))
    t_prj, _ = timed(lambda: H.project_3D_points(p3, ext, K, xp))
    Hh = min(256, S * (S - 1) // 2)
    flops = N * (Hh * (2.5e3 + 70 * S) + 60 * (150 * S + 2e3 + 70 * S) + 60 * 40 * S * S)
    print(json.dumps({"config": name, "frames": S, "tracks": N, "triangulate_tracks_s": t_tri, "tracks_per_s": N / t_tri,
                      "reference_formulation_TFLOPs": flops / t_tri / 1e12, "valid_frac": float((num >= 3).float().mean()),
                      "cam_from_img_ms": t_und * 1e3, "filter_all_points3D_ms": t_flt * 1e3, "project_ms": t_prj * 1e3,
                      "filter_GBs": S * N * 9 / t_flt / 1e9, "project_GBs": S * N * 24 / t_prj / 1e9}))
