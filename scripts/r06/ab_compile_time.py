"""compile_problem wall time (synchronised) at a video-window size and at configs[1] / configs[2] size; usage: ab_compile_time.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vggsfm_amd import ba as BA
from vggsfm_amd.scene import make_scene, make_scene_device, perturb_for_ba
import bench as B
if len(sys.argv) > 1 and sys.argv[1] == "no_output_size":      # (the repeat_interleave calls of ba.py as they were before round 6)
    _ri = torch.repeat_interleave
    torch.repeat_interleave = lambda *a, output_size=None, **k: _ri(*a, **k)
D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
for name, (S, N, cam, shared) in (("window 17x3000", (17, 3000, "SIMPLE_RADIAL", True)), ("c2", B.WORKLOADS["c2"]), ("c3", B.WORKLOADS["c3"])):
    sc = make_scene(S, N, cam, shared_camera=shared, seed=0, track_seed=1000)
    ext0, K0, xp0, pts0 = perturb_for_ba(sc, seed=0)
    args = [D(x) for x in (pts0, ext0, K0, sc.tracks, sc.mask, xp0)]
    ts = []
    for rep in range(6):
        torch.cuda.synchronize(); t = time.perf_counter()
        prob, _, _ = BA.compile_problem(*args, shared, cam, camera_split=True, sort_points=True)
        torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
    print(json.dumps(dict(variant=(sys.argv[1] if len(sys.argv) > 1 else "tree"), problem=name, obs=int(prob.num_obs), compile_ms_min=round(min(ts[1:]), 3), compile_ms=[round(t, 2) for t in ts])), flush=True)
