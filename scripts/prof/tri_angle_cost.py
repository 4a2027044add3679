"""How much of the LO-RANSAC kernel time is the exhaustive triangulation-angle scan?  (min_tri_angle = -1 makes
every finite point pass at the first pair.)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vggsfm_amd.scene import make_scene
from vggsfm_amd.utils import triangulation as T, triangulation_helpers as H
D = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
sc = make_scene(200, 100000, "SIMPLE_RADIAL", shared_camera=True, seed=0)
tn = H.cam_from_img(D(sc.tracks), D(sc.intrinsics), D(sc.extra_params))
for kw in (dict(), dict(max_ransac_iters=64)):
    for rep in range(2):
        torch.manual_seed(0); torch.cuda.synchronize(); t0 = time.perf_counter()
        T.triangulate_tracks(D(sc.extrinsics), tn, track_vis=D(sc.vis), track_score=D(sc.score), **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(kw, "time", round(dt, 4))
