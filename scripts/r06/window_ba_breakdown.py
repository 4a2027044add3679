"""Where do the ~5 ms of a video-window BA call go?  compile_problem / solve / the rest, synchronising timers, 17 frames."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vggsfm_amd import ba as BA
from vggsfm_amd.ba_options import BundleAdjustmentOptions
from vggsfm_amd.scene import make_scene, perturb_for_ba
from torch.profiler import profile, ProfilerActivity
D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
sc = make_scene(17, 3000, "SIMPLE_RADIAL", shared_camera=True, seed=3)
ext0, K0, xp0, pts0 = perturb_for_ba(sc, seed=3)
args = [D(x) for x in (pts0, ext0, K0, sc.tracks, sc.mask, xp0)]
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for rep in range(4):
    t0 = sync()
    prob, vi, de = BA.compile_problem(*args, True, "SIMPLE_RADIAL", gauge="config", camera_split=True, refine_focal_length=False, refine_extra_params=False, sort_points=True, filter_negative_depth=False)
    t1 = sync()
    opt = BundleAdjustmentOptions(); opt.refine_focal_length = opt.refine_extra_params = False
    prob.cam_const[0] = 1
    summ, _ = BA.solve(prob, opt)
    t2 = sync()
    out = BA.window_bundle_adjustment(args[0], args[1], args[2], args[3], args[4], 1000, args[5], True, "SIMPLE_RADIAL")
    t3 = sync()
    print(json.dumps(dict(rep=rep, obs=prob.num_obs, compile_ms=round(1e3*(t1-t0),3), solve_ms=round(1e3*(t2-t1),3), iterations=summ["num_iterations"], whole_call_ms=round(1e3*(t3-t2),3))), flush=True)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    BA.compile_problem(*args, True, "SIMPLE_RADIAL", gauge="config", camera_split=True, refine_focal_length=False, refine_extra_params=False, sort_points=True, filter_negative_depth=False)
    torch.cuda.synchronize()
ev = prof.key_averages()
print("kernel launches:", sum(e.count for e in ev if e.device_type.name == "CUDA" or "Memcpy" in e.key), "aten calls:", sum(e.count for e in ev if e.key.startswith("aten::")), "items:", sum(e.count for e in ev if e.key == "aten::item"))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=14))
