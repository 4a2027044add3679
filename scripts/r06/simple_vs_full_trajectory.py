"""Does the simple work list of small problems (ba.SIMPLE_WORKLIST_MAX_OBS) change the LM trajectory?  One video-window-like
problem solved with the full and with the simple list: cost / radius / acceptance per iteration side by side."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vggsfm_amd import ba as BA
from vggsfm_amd.ba_options import BundleAdjustmentOptions
from vggsfm_amd.scene import make_scene, perturb_for_ba
D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
for seed, S, N in ((3, 17, 3000), (5, 17, 6000), (7, 33, 4000)):
    sc = make_scene(S, N, "SIMPLE_RADIAL", shared_camera=True, seed=seed)
    ext0, K0, xp0, pts0 = perturb_for_ba(sc, seed=seed)
    args = [D(x) for x in (pts0, ext0, K0, sc.tracks, sc.mask, xp0)]
    logs = {}
    for name, lim in (("full", 0), ("simple", 10**9)):
        BA.SIMPLE_WORKLIST_MAX_OBS = lim
        prob, vi, de = BA.compile_problem(*args, True, "SIMPLE_RADIAL", gauge="config", camera_split=True, refine_focal_length=False,
                                          refine_extra_params=False, sort_points=True, filter_negative_depth=False)
        prob.cam_const[0] = 1
        opt = BundleAdjustmentOptions(); opt.refine_focal_length = opt.refine_extra_params = False
        summ, _ = BA.solve(prob, opt)
        logs[name] = summ
        print(json.dumps(dict(seed=seed, S=S, N=N, obs=int(prob.num_obs), variant=name, iterations=summ["num_iterations"], final_cost=summ["final_cost"],
                              termination=summ["termination_str"],
                              costs=[it["cost"] for it in summ.get("iterations", [])][:40])), flush=True)
