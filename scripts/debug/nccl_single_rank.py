"""RCCL plumbing check on one GPU: world_size-1 process group, the ShardedBA phases with REAL dist.all_reduce calls on
the reduce buffers (float64 views into the uint8 workspace) -- must reproduce the plain single-rank solve."""
import os, sys
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vggsfm_amd import ba as BA
from vggsfm_amd.ba_options import BundleAdjustmentOptions
from vggsfm_amd.dist import ShardedBA
from vggsfm_amd.scene import make_scene, perturb_for_ba
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
sc = make_scene(30, 3000, "SIMPLE_RADIAL", shared_camera=True, seed=3)
ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=3)
opts = BundleAdjustmentOptions(); opts.solver_options.max_num_iterations = 12
def compile_():
    return BA.compile_problem(D(pts0), D(ext0), D(K0), D(sc.tracks), D(sc.mask), D(extra0), True, "SIMPLE_RADIAL")[0]
ref = ShardedBA(compile_(), opts).solve()
def ar(t, op):
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
s = ShardedBA(compile_(), opts, rank=0, world_size=1, all_reduce=ar)
s.begin()
for _ in range(13):
    s._phase(0); ar(s.bufs[0], "sum")
    s._phase(1); s._phase(4); ar(s.bufs[4], "sum"); s._phase(5); ar(s.bufs[2], "max")
    s._phase(2); ar(s.bufs[3], "sum")
    s._phase(3)
out = s.finish(20)
print("iterations", out["num_iterations"], ref["num_iterations"], "final cost", out["final_cost"], ref["final_cost"])
assert out["num_iterations"] == ref["num_iterations"] and out["final_cost"] == ref["final_cost"]
print("RCCL single-rank plumbing OK")
dist.destroy_process_group()
