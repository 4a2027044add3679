"""Host-side estimate of the MFMA work of super_tile_kernel on bench workload c3 from the real quad masks."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vggsfm_amd import ba as BA
from vggsfm_amd.scene import make_scene

S, N = 200, int(os.environ.get("TRACKS", "100000"))
sc = make_scene(S, N, "SIMPLE_RADIAL", shared_camera=True, seed=0, track_seed=1000)
m = torch.from_numpy(sc.mask)
perm, fg = BA.find_camera_order(m)
if perm is None:
    perm, _ = BA.find_camera_split(m)
print("camera perm:", None if perm is None else perm[:40].tolist())
if perm is not None:
    m = m[perm]
pm = torch.nonzero(m.t())
obs_cam = pm[:, 1].to(torch.int32)
row_ptr = torch.zeros(N + 1, dtype=torch.int32)
row_ptr[1:] = torch.cumsum(m.sum(0), 0).to(torch.int32)
cd, ent, qm, td, slot, nseg = BA.build_schur_supertiles(row_ptr, obs_cam)
cd, qm, td = cd.numpy(), qm.numpy().astype(np.int64) & 0xffffffff, td.numpy()
def blockbits(blk):
    half, b = divmod(blk, 6)
    s0, s1 = (16 * b) // 6, min(15, (16 * b + 15) // 6)
    return (((2 << s1) - 1) & ~((1 << s0) - 1)) << (16 * half)
bits = np.array([blockbits(b) for b in range(12)])
ent = ent.numpy()
seg_mask = np.zeros(nseg + 1, np.int64)
np.bitwise_or.at(seg_mask, slot.numpy() // 16, 1 << (slot.numpy() % 16))
emA = seg_mask[ent[:, 0]] | (seg_mask[ent[:, 1]] << 16)
emB = seg_mask[ent[:, 2]] | (seg_mask[ent[:, 3]] << 16)
tot_entry = 0.0; tot_sorted = 0.0
tot_off = tot_diag = tot_off2 = tot_off_ideal = tot_diag_ideal = 0.0
for sI, sJ, c0, c1 in td:
    tb, te, q0 = cd[c0, 2], cd[c0, 3], cd[c0, 6]
    nq = (te - tb + 3) // 4
    qa, qb = qm[q0:q0 + nq, 0], qm[q0:q0 + nq, 1]
    ra = (qa[:, None] & bits[None]) != 0          # (nq, 12)
    cb = (qb[:, None] & bits[None]) != 0
    eA = (emA[tb:te, None] & bits[None]) != 0
    eB = (emB[tb:te, None] & bits[None]) != 0
    if sI != sJ:
        tot_entry += (eA.sum(1) * eB.sum(1)).sum() * 3 / 4 / 4          # per entry, perfectly balanced over 4 SIMDs, 4 entries per MFMA
        # entries of the tile re-sorted by their own block pattern, then quads
        ka = (eA * (1 << np.arange(12))).sum(1); kb = (eB * (1 << np.arange(12))).sum(1)
        o = np.lexsort((kb, ka))
        n4 = (len(o) + 3) // 4 * 4
        pa = np.zeros((n4, 12), bool); pb_ = np.zeros((n4, 12), bool)
        pa[:len(o)] = eA[o]; pb_[:len(o)] = eB[o]
        ua = pa.reshape(-1, 4, 12).any(1); ub = pb_.reshape(-1, 4, 12).any(1)
        ps2 = np.zeros((ua.shape[0], 4))
        for r in range(12):
            for c in range(12):
                ps2[:, (r + c) % 4] += (ua[:, r] & ub[:, c]) * 3
        tot_sorted += ps2.max(1).sum()
    else:
        n_act = eA.sum(1)
        tot_entry += (n_act * (n_act + 1) / 2).sum() * 3 / 4 / 4
    if sI != sJ:
        rows = np.stack([ra[:, wr::2].sum(1) for wr in range(2)], 1)       # (nq, 2)
        cols = np.stack([cb[:, wc::4].sum(1) for wc in range(4)], 1)       # (nq, 4)
        per_simd = (rows.sum(1)[:, None] * cols) * 3                       # both wr waves of a SIMD (same wc)
        t = per_simd.max(1)
        tot_off += t.sum()
        # diagonal striping: sub-tile (r, c) on SIMD (r + c) % 4
        ps = np.zeros((nq, 4))
        for r in range(12):
            for c in range(12):
                ps[:, (r + c) % 4] += (ra[:, r] & cb[:, c]) * 3
        tot_off2 += ps.max(1).sum(); tot_off_ideal += ps.sum(1).sum() / 4
        print("tile", sI, sJ, "quads", nq, "mean MFMAs per SIMD per batch %.1f (of 108), mean active rows %.1f cols %.1f" % (t.mean(), ra.sum(1).mean(), cb.sum(1).mean()))
    else:
        tt = [(r, c) for r in range(12) for c in range(r + 1)]
        per_wave = np.zeros((nq, 8))
        for t_, (r, c) in enumerate(tt):
            per_wave[:, t_ % 8] += (ra[:, r] & ra[:, c]) * 3
        per_simd = per_wave[:, :4] + per_wave[:, 4:]
        t = per_simd.max(1)
        tot_diag += t.sum(); tot_diag_ideal += per_simd.sum(1).sum() / 4
        print("tile", sI, sJ, "quads", nq, "mean MFMAs per SIMD per batch %.1f (of 60)" % t.mean())
f = 64.8 / 2.4e9 / 256 * 1e3
print("striped: off %.3f ms; perfectly balanced: off %.3f diag %.3f ms" % (tot_off2 * f, tot_off_ideal * f, tot_diag_ideal * f))
print("per-entry lower bound (off + diag) %.3f ms; off tiles re-sorted by pattern, striped: %.3f ms" % (tot_entry * f, tot_sorted * f))
print("MFMA-bound time on 256 CUs: off %.3f ms, diag %.3f ms" % (tot_off * 64.8 / 2.4e9 / 256 * 1e3, tot_diag * 64.8 / 2.4e9 / 256 * 1e3))
