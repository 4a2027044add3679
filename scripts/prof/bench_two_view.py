"""Wall time of the two-view stage (estimate_preliminary_cameras) at the sizes of BASELINE configs[1] / [2]."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vggsfm_amd.scene import make_scene
from vggsfm_amd.two_view_geo import estimate_preliminary_cameras

for S, N in ((8, 6000), (50, 20000), (200, 100000)):
    sc = make_scene(S, N, "SIMPLE_PINHOLE", seed=1, full_visibility=True, outlier_frac=0.1)
    tr = torch.from_numpy(sc.tracks).cuda()[None]
    vis = torch.from_numpy(sc.vis).cuda()[None]
    np.random.seed(0)
    estimate_preliminary_cameras(tr[:, :3], vis[:, :3], 1024, 1024, max_error=4.0)          # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, pre = estimate_preliminary_cameras(tr, vis, 1024, 1024, max_error=4.0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    m = pre["fmat_inlier_mask"][0]
    print(f"{S} frames x {N} tracks: {S - 1} pairs x 4096 samples (12288 + 450 hypotheses each): {dt * 1e3:.1f} ms, "
          f"inlier fraction {float(m.float().mean()):.3f}", flush=True)
