"""Phase cycle sums of the off-diagonal Schur tile launch per wavefront (library built with -DVGG_TILE_TRACE=1 as
vggsfm_amd/_variants/lib_tile_trace.so):  VGGSFM_AMD_LIB=vggsfm_amd/_variants/lib_tile_trace.so python scripts/prof/tile_trace.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vggsfm_amd import _lib  # noqa: E402
from vggsfm_amd import ba as BA  # noqa: E402
from vggsfm_amd.ba_options import BundleAdjustmentOptions  # noqa: E402
from vggsfm_amd.scene import make_scene, perturb_for_ba  # noqa: E402

dev = torch.device("cuda:0")
D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev)
sc = make_scene(200, 100000, "SIMPLE_RADIAL", shared_camera=True, seed=0)
ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=0)
opt = BundleAdjustmentOptions()
opt.solver_options.max_num_iterations = 3
BA.bundle_adjustment(D(pts0), D(ext0), D(K0), D(sc.tracks), D(sc.mask), None, D(extra0), True, "SIMPLE_RADIAL", opt)
torch.cuda.synchronize()
n = 768 * 4 * 8
buf = (ctypes.c_longlong * n)()
assert _lib.lib().vgg_debug_read_tile_trace(buf, ctypes.c_size_t(n)) == 0
t = np.array(buf[:], dtype=np.int64).reshape(768, 4, 8)
# s_memtime counts shader cycles, wall_clock64 a constant 100 MHz: their ratio over the launch is the shader clock
nb = t[:, :, 0].astype(float)
ok = t[:, :, 6] > 0
print("shader clock during the launch: %.0f MHz (wall time per workgroup %.3f ms)" % (
    (t[:, :, 5][ok] / t[:, :, 6][ok]).mean() * 100.0, t[:, :, 6][ok].mean() / 100e6 * 1e3))
print("workgroups", t.shape[0], "batches per workgroup: mean %.1f min %d max %d" % (nb.mean(), nb.min(), nb.max()))
for name, k in (("issue", 1), ("matrix", 2), ("lds write", 3), ("barrier", 4), ("total", 5)):
    per = t[:, :, k] / np.maximum(nb, 1) * 1.0
    print("%-10s shader cycles per batch: mean %8.0f  p10 %8.0f  p90 %8.0f   (sum over the launch: %.3f ms per wave)" % (
        name, per.mean(), np.percentile(per, 10), np.percentile(per, 90), t[:, :, k].mean() / 2.4e9 * 1e3))
for w in range(4):
    print("wave", w, {name: round(float((t[:, w, k] / np.maximum(nb[:, w], 1)).mean())) for name, k in
                      (("issue", 1), ("matrix", 2), ("write", 3), ("barrier", 4))})
# where does the spread between workgroups come from?  by XCD (launch position mod 8) and by launch position
tot = t[:, :, 6].mean(1) / 100e6 * 1e3                               # wall ms per workgroup
print("wall ms per workgroup: min %.3f p10 %.3f median %.3f p90 %.3f max %.3f" % (
    tot.min(), np.percentile(tot, 10), np.median(tot), np.percentile(tot, 90), tot.max()))
print("by XCD (position mod 8):", [round(float(tot[x::8].mean()), 3) for x in range(8)])
print("by position octile:", [round(float(tot[i * 96:(i + 1) * 96].mean()), 3) for i in range(8)])
print("batches by position octile:", [round(float(nb[i * 96:(i + 1) * 96].mean()), 1) for i in range(8)])
per_batch = t[:, :, 5].mean(1) / np.maximum(nb.mean(1), 1)
print("cycles per batch by position octile:", [round(float(per_batch[i * 96:(i + 1) * 96].mean())) for i in range(8)])
mat = t[:, :, 2].mean(1) / np.maximum(nb.mean(1), 1)
print("matrix-phase cycles per batch by position octile:", [round(float(mat[i * 96:(i + 1) * 96].mean())) for i in range(8)])
