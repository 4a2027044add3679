"""triangulate_tracks timing at configs[1] / configs[2] for whatever library VGGSFM_AMD_LIB selects (variant builds of
triangulate.hip: -DVGG_TRI_EIG=0 / -DVGG_TRI_FAST_ERR=0 / -DVGG_TRI_OCC=n / -DVGG_TRI_STATS); the scene is cached in /tmp
so that several variants in one GPU call do not each pay for its generation.  Prints one JSON line per configuration."""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vggsfm_amd import _lib  # noqa: E402
from vggsfm_amd.scene import make_scene  # noqa: E402
from vggsfm_amd.utils import triangulation as T  # noqa: E402
from vggsfm_amd.utils import triangulation_helpers as H  # noqa: E402

CONFIGS = {"c2": (50, 20000, "SIMPLE_PINHOLE", False), "c3": (200, 100000, "SIMPLE_RADIAL", True)}
D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
L = _lib.lib()
for name in (sys.argv[1:] or ["c2", "c3"]):
    S, N, cam, shared = CONFIGS[name]
    cache = f"/tmp/tri_scene_{name}.npz"
    if os.path.exists(cache):
        g = np.load(cache)
        ext, K, xp, tracks, vis, score = (g[k] if k in g else None for k in ("ext", "K", "xp", "tracks", "vis", "score"))
    else:
        sc = make_scene(S, N, cam, shared_camera=shared, seed=0)
        ext, K, xp, tracks, vis, score = sc.extrinsics, sc.intrinsics, sc.extra_params, sc.tracks, sc.vis, sc.score
        np.savez(cache, **{k: v for k, v in dict(ext=ext, K=K, xp=xp, tracks=tracks, vis=vis, score=score).items() if v is not None})
    ext, K, xp, tracks, vis, score = D(ext), D(K), D(xp), D(tracks), D(vis), D(score)
    tn = H.cam_from_img(tracks, K, xp)
    torch.manual_seed(0)
    out = T.triangulate_tracks(ext, tn, track_vis=vis, track_score=score)      # warm-up
    if hasattr(L, "vgg_debug_tri_stats"):
        st = (ctypes.c_ulonglong * 4)()
        L.vgg_debug_tri_stats(st, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    torch.manual_seed(0)
    e0.record()
    for _ in range(reps):
        pts, num, msk = T.triangulate_tracks(ext, tn, track_vis=vis, track_score=score)
    e1.record()
    torch.cuda.synchronize()
    res = dict(config=name, lib=os.path.basename(os.environ.get("VGGSFM_AMD_LIB", "default")), ms=e0.elapsed_time(e1) / reps,
               valid_frac=float((num >= 3).float().mean()), inliers_sum=int(num.sum()), mask_bits=int(msk.sum()),
               pts_checksum=float(pts[num >= 3].abs().sum()))
    if hasattr(L, "vgg_debug_tri_stats"):
        st = (ctypes.c_ulonglong * 4)()
        L.vgg_debug_tri_stats(st, 0)
        res["eigen_solves_per_lane_call"] = int(st[0])
        res["fallbacks"] = int(st[1])
        res["wave_calls_with_fallback"] = int(st[2])
    print(json.dumps(res), flush=True)
