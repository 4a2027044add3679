"""Calibration of the tile cost model (ba._entry_cost / TILE_FIXED_COST): per-workgroup cycle totals of the off-diagonal launch
(library built with -DVGG_TILE_TRACE=1 -DVGG_TILE_PIPE=0 as vggsfm_amd/_variants/lib_tile_trace.so) against what the work list
says each workgroup executes -- batches, and matrix instructions of its busiest / mean wavefront -- by least squares:
    cycles(workgroup) ~ a * batches + b * matrix_instructions
usage: VGGSFM_AMD_LIB=vggsfm_amd/_variants/lib_tile_trace.so python scripts/prof/tile_cost_fit.py"""
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
sc = make_scene(200, 100000, "SIMPLE_RADIAL", shared_camera=True, seed=0, track_seed=1000)
ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=0)
prob, _, _ = BA.compile_problem(D(pts0), D(ext0), D(K0), D(sc.tracks), D(sc.mask), D(extra0), True, "SIMPLE_RADIAL", camera_split=True)
opt = BundleAdjustmentOptions()
opt.solver_options.max_num_iterations = 3
BA.solve(prob, opt)
torch.cuda.synchronize()
n = 768 * 4 * 8
buf = (ctypes.c_longlong * n)()
assert _lib.lib().vgg_debug_read_tile_trace(buf, ctypes.c_size_t(n)) == 0
t = np.array(buf[:], dtype=np.int64).reshape(768, 4, 8)
cd = prob.chunk_desc.cpu().numpy()
ent = prob.entries.cpu().numpy()
noff = int((cd[:, 0] != cd[:, 1]).sum())
BITS = np.array([((2 << min(15, (16 * b + 15) // 6)) - 1) & ~((1 << ((16 * b) // 6)) - 1) for b in range(6)])
SUB = BA.SUB
rows = []
for c in range(min(noff, 768)):
    e0, e1, cj, cJ = cd[c, 2:6]
    nsub = (e1 - e0 + SUB - 1) // SUB
    subs = np.arange(cj, nsub, cJ)
    q0 = (e0 + subs[:, None] * SUB + np.arange(0, SUB, 4)[None]).reshape(-1)
    q0 = q0[q0 < e1]
    qm = ent[q0, 3].astype(np.int64)
    ra = ((qm[:, None] & 0xFFFF) & BITS[None]) != 0
    rb = ((qm[:, None] >> 16) & BITS[None]) != 0
    per = np.zeros((len(qm), 4))
    for wr in range(2):
        for wc in range(2):
            per[:, 2 * wr + wc] = ra[:, wr::2].sum(1) * rb[:, wc::2].sum(1) * 3
    nbatch = len(q0)
    cyc = t[c, :, 5].mean()
    rows.append((nbatch, per.max(1).sum(), per.mean(1).sum(), cyc, t[c, :, 6].mean()))
r = np.array(rows, float)
ok = r[:, 3] > 0
r = r[ok]
for name, col in (("busiest", 1), ("mean", 2)):
    A = np.stack([r[:, 0], r[:, col]], 1)
    coef, res, *_ = np.linalg.lstsq(A, r[:, 3], rcond=None)
    pred = A @ coef
    print(f"{name:8s}: cycles ~ {coef[0]:.0f} * batches + {coef[1]:.1f} * matrix_instructions   (fixed cost = {coef[0] / coef[1]:.1f} instructions; "
          f"rms rel err {np.sqrt(np.mean(((pred - r[:, 3]) / r[:, 3]) ** 2)):.3f})")
cur = 18.0 * r[:, 0] + r[:, 1]
print("current model: spread of predicted cost per workgroup (p10 / p90 of cost/mean):", np.percentile(cur / cur.mean(), [10, 90]).round(3),
      " measured cycles/mean:", np.percentile(r[:, 3] / r[:, 3].mean(), [10, 50, 90, 100]).round(3))
print("batches per workgroup p10/p50/p90:", np.percentile(r[:, 0], [10, 50, 90]), " matrix instr per batch (busiest) p10/p50/p90:",
      np.percentile(r[:, 1] / r[:, 0], [10, 50, 90]).round(1))

# per TILE: cycles per batch (mean over its workgroups) against its matrix instructions per batch
tiles = {}
for c in range(min(noff, 768)):
    if t[c, :, 5].mean() <= 0:
        continue
    key = (int(cd[c, 0]), int(cd[c, 1]))
    tiles.setdefault(key, []).append(c)
rowsT = []
for key, cs in tiles.items():
    rr = np.array([rows[c] for c in cs])
    rowsT.append((key[0], key[1], len(cs), rr[:, 0].sum(), rr[:, 1].sum() / rr[:, 0].sum(), rr[:, 2].sum() / rr[:, 0].sum(), rr[:, 3].sum() / rr[:, 0].sum()))
T = np.array(rowsT, float)
for name, col in (("busiest", 4), ("mean", 5)):
    A = np.stack([np.ones(len(T)), T[:, col]], 1)
    coef, *_ = np.linalg.lstsq(A, T[:, 6], rcond=None)
    pred = A @ coef
    print(f"per tile, {name}: cycles/batch ~ {coef[0]:.0f} + {coef[1]:.1f} * instr/batch  (fixed = {coef[0] / coef[1]:.1f} instr; rms rel {np.sqrt(np.mean(((pred - T[:, 6]) / T[:, 6]) ** 2)):.3f})")
print("tile gI gJ wgs batches busiest/batch mean/batch cycles/batch")
for r_ in T[np.argsort(-T[:, 6])][:8]:
    print("  slow", r_.round(1))
for r_ in T[np.argsort(T[:, 6])][:6]:
    print("  fast", r_.round(1))
