"""Window-like BA problems of 2..20 frames: final cost and iteration count (run once per VGG_DF_MIN_N setting and compare)."""
import sys, os, json; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vggsfm_amd import ba as BA
from vggsfm_amd.scene import make_scene, perturb_for_ba
D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
for S in range(2, 21):
    for cam, shared in (("SIMPLE_RADIAL", True), ("SIMPLE_PINHOLE", False)):
        sc = make_scene(S, 800, cam, shared_camera=shared, seed=S)
        ext0, K0, xp0, pts0 = perturb_for_ba(sc, seed=S)
        a = [D(x) for x in (pts0, ext0, K0, sc.tracks, sc.mask, xp0)]
        out = BA.window_bundle_adjustment(a[0], a[1], a[2], a[3], a[4], 200, a[5], shared, cam)
        summ = out[4]
        print(json.dumps(dict(frames=S, cam=cam, n=summ["n_reduced"], it=summ["num_iterations"], cost=summ["final_cost"], term=summ["termination"])))
