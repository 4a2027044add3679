"""Where a small (video-window sized) bundle adjustment spends its time: problem compilation (host tensor ops), the solve,
the glue of ba.bundle_adjustment around them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from vggsfm_amd import ba as BA
from vggsfm_amd.ba_options import BundleAdjustmentOptions
from vggsfm_amd.scene import make_scene, perturb_for_ba

dev = torch.device("cuda:0")
for S, N in ((17, 6000), (33, 8000), (50, 20000)):
    sc = make_scene(S, N, "SIMPLE_RADIAL", shared_camera=True, seed=1)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=1)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    args = (T(pts0), T(ext0), T(K0), T(sc.tracks), T(sc.mask), T(extra0), True, "SIMPLE_RADIAL")
    opt = BundleAdjustmentOptions()
    opt.solver_options.max_num_iterations = 10
    opt.solver_options.gradient_tolerance = 0.0
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        prob, vi, dele = BA.compile_problem(*args)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        summ, ws = BA.solve(prob, opt)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        out = BA.bundle_adjustment(args[0], args[1], args[2], args[3], args[4], None, args[5], True, "SIMPLE_RADIAL", opt)
        torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"{S} x {N}: compile_problem {1e3*(t1-t0):.2f} ms, solve ({summ['num_iterations']} iterations) {1e3*(t2-t1):.2f} ms, "
          f"bundle_adjustment end to end {1e3*(t3-t2):.2f} ms")
