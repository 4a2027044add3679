"""Where does ba.compile_problem spend its time at 200 x 100k?  (torch profiler, top ops by device+host time)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vggsfm_amd import ba as BA
from vggsfm_amd.scene import make_scene, perturb_for_ba
D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
sc = make_scene(200, 100000, "SIMPLE_RADIAL", shared_camera=True, seed=0)
ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=0)
args = (D(pts0), D(ext0), D(K0), D(sc.tracks), D(sc.mask), D(extra0), True, "SIMPLE_RADIAL")
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); BA.compile_problem(*args); torch.cuda.synchronize()
    print("compile_problem", round(time.perf_counter() - t0, 4))
import torch.profiler as tp
with tp.profile(activities=[tp.ProfilerActivity.CPU, tp.ProfilerActivity.CUDA]) as prof:
    BA.compile_problem(*args); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=45))
