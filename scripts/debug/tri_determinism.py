"""Runs triangulate_tracks repeatedly on identical inputs and reports bitwise differences between runs."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vggsfm_amd.scene import make_scene
from vggsfm_amd.utils import triangulation as T, triangulation_helpers as H

D = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
sc = make_scene(50, 20000, "SIMPLE_PINHOLE", seed=2, outlier_frac=0.05)
tn = H.cam_from_img(D(sc.tracks), D(sc.intrinsics))
ref = None
for rep in range(6):
    torch.manual_seed(0)
    if rep % 2 == 1:   # dirty the allocator between runs
        junk = torch.full((64 * 1024 * 1024,), float("nan"), device="cuda"); del junk
    pts, num, msk = T.triangulate_tracks(D(sc.extrinsics), tn, track_vis=D(sc.vis), track_score=D(sc.score))
    err = (pts - D(sc.points3D)).norm(dim=-1)[num >= 3]
    print(rep, "valid", int((num >= 3).sum()), "median", float(err.median()), "q99", float(err.quantile(0.99)), "max", float(err.max()))
    if ref is None:
        ref = (pts.clone(), num.clone(), msk.clone())
    else:
        dp = (pts != ref[0]).any(-1)
        print("   differs: pts", int(dp.sum()), "num", int((num != ref[1]).sum()), "mask", int((msk != ref[2]).any(-1).sum()))
