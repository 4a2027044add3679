"""triangulate_tracks at BASELINE configs[2] size: where the time goes (hypotheses vs local optimisation)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from vggsfm_amd.scene import make_scene
from vggsfm_amd.utils.triangulation import triangulate_tracks
from vggsfm_amd.utils.triangulation_helpers import cam_from_img

dev = torch.device("cuda:0")
S, N = 200, 100000
sc = make_scene(S, N, "SIMPLE_RADIAL", shared_camera=True, seed=0)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
ext, K, extra = T(sc.extrinsics), T(sc.intrinsics), T(sc.extra_params)
tn = cam_from_img(T(sc.tracks), K, extra)
vis = T(sc.vis)
for iters, lo in ((256, 50), (256, 2), (32, 50), (32, 2), (128, 50)):
    for rep in range(2):
        torch.manual_seed(0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pts, num, mask = triangulate_tracks(ext, tn, max_ransac_iters=iters, lo_num=lo, track_vis=vis)
        torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"max_ransac_iters {iters:4d} lo_num {lo:3d}: {1e3*(t1-t0):7.2f} ms   mean inliers {num.float().mean().item():.1f}")

import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    torch.manual_seed(0)
    triangulate_tracks(ext, tn, max_ransac_iters=256, lo_num=50, track_vis=vis)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
