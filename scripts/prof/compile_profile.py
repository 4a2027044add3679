"""Where ba.compile_problem spends its time (BASELINE configs[2] by default): torch profiler, device kernels and host ops.
usage: python scripts/prof/compile_profile.py [frames tracks]"""
import os
import sys
import time

import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vggsfm_amd import ba as BA  # noqa: E402
from vggsfm_amd.scene import make_scene, perturb_for_ba  # noqa: E402

dev = torch.device("cuda:0")
S_, N_ = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 100000)
sc = make_scene(S_, N_, "SIMPLE_RADIAL", shared_camera=True, seed=1)
ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=1)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
args = (T(pts0), T(ext0), T(K0), T(sc.tracks), T(sc.mask), T(extra0), True, "SIMPLE_RADIAL")
for _ in range(2):
    BA.compile_problem(*args, camera_split=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    BA.compile_problem(*args, camera_split=True)
torch.cuda.synchronize()
print(f"compile_problem: {1e3 * (time.perf_counter() - t0) / 5:.2f} ms per call")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        BA.compile_problem(*args, camera_split=True)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=25, max_name_column_width=60))
