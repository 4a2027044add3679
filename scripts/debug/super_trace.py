"""Per-workgroup cycle counts of super_tile_kernel on bench workload c3 (needs a library built with -DVGG_SUPER_TRACE=1,
passed as VGGSFM_AMD_LIB): two LM iterations, the kernel prints one TRACE line per workgroup."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from vggsfm_amd import ba as BA
from vggsfm_amd.ba_options import BundleAdjustmentOptions
from vggsfm_amd.scene import make_scene, perturb_for_ba

dev = torch.device("cuda:0")
S, N = 200, int(os.environ.get("TRACE_TRACKS", "100000"))
sc = make_scene(S, N, "SIMPLE_RADIAL", shared_camera=True, seed=0, track_seed=1000)
ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=0)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
prob, _, _ = BA.compile_problem(T(pts0), T(ext0), T(K0), T(sc.tracks), T(sc.mask), T(extra0), True, "SIMPLE_RADIAL", camera_split=True)
opt = BundleAdjustmentOptions()
opt.solver_options.max_num_iterations = 1
opt.solver_options.gradient_tolerance = 0.0
opt.solver_options.function_tolerance = 0.0
opt.solver_options.parameter_tolerance = 0.0
summ, _ = BA.solve(prob, opt)
torch.cuda.synchronize()
print("done", summ["num_iterations"], prob.chunk_desc.shape)
import ctypes
from vggsfm_amd import _lib
n = 1024 * 8 * 8
buf = (ctypes.c_longlong * n)()
rc = _lib.lib().vgg_debug_read_super_trace(buf, ctypes.c_size_t(n))
tr = np.array(buf[:], dtype=np.int64).reshape(1024, 8, 8)[:prob.chunk_desc.shape[0]]
np.save(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "super_trace.npy"), tr)
w0 = tr[:, 0]
tot = w0[:, 3]
print("cycles per workgroup: min %d median %d max %d" % (tot.min(), np.median(tot), tot.max()))
print("cycles per batch:", np.round(np.percentile(tot / np.maximum(w0[:, 2], 1), [0, 25, 50, 75, 100])))
nbt = np.maximum(w0[:, 2], 1)
for name, col in (("load wait", 4), ("barrier wait", 5), ("pre-MFMA (index readfirstlane, index DMA, fetch issue)", 6), ("MFMA phase (operand reads, tests, matrix instructions, DMA pieces)", 7)):
    for wv in (0, 4):
        print("wave %d  %-70s cycles per batch: %s" % (wv, name, np.round(np.percentile(tr[:, wv, col] / nbt, [0, 25, 50, 75, 100])).tolist()))
