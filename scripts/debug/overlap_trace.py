"""One short overlapped solve for a kernel-trace timeline (rocprofv3 --kernel-trace): which stream waits for which."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vggsfm_amd import ba as BA
from vggsfm_amd.ba_options import BundleAdjustmentOptions
from vggsfm_amd.dist import ShardedBA
from vggsfm_amd.scene import make_scene, perturb_for_ba

D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
overlap = (sys.argv[1] != "0") if len(sys.argv) > 1 else True
if len(sys.argv) > 2:
    BA.CHOL_CUS = int(sys.argv[2])
if len(sys.argv) > 3:
    BA.TILE_BATCHES = int(sys.argv[3])
sc = make_scene(200, 100000, "SIMPLE_RADIAL", shared_camera=True, seed=0)
ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=0)
prob, _, _ = BA.compile_problem(D(pts0), D(ext0), D(K0), D(sc.tracks), D(sc.mask), D(extra0), True, "SIMPLE_RADIAL", overlap=overlap)
opts = BundleAdjustmentOptions()
opts.solver_options.max_num_iterations = 8
for name in ("function_tolerance", "gradient_tolerance", "parameter_tolerance"):
    setattr(opts.solver_options, name, -1.0)
s = ShardedBA(prob, opts)
s.begin()
for _ in range(6):
    s.iteration()
torch.cuda.synchronize()
print(s.finish(10)["final_cost"])
