"""What the Schur tile launch executes, counted on the CPU from the work list of the c3 problem (no GPU needed):
matrix instructions per quad for the entries' own block patterns, for the union over the four K-packed entries, and for
the busiest of the four wavefronts -- with and without the pattern grouping of ba.QUAD_SORT_WINDOW, and for other
assignments of the 36 sub-tiles to the wavefronts.  DESIGN.md section 6 quotes its output.
usage: python scripts/prof/tile_structure.py [frames tracks]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vggsfm_amd import ba as BA  # noqa: E402
from vggsfm_amd.scene import make_scene, perturb_for_ba  # noqa: E402

S, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 100000)
sc = make_scene(S, N, "SIMPLE_RADIAL", shared_camera=True, seed=0, track_seed=1000)
_, _, _, pts0 = perturb_for_ba(sc, seed=0)
ext0, K0, extra0, _ = perturb_for_ba(sc, seed=0)
T = lambda x: torch.from_numpy(np.asarray(x))  # noqa: E731
BITS = np.array([((2 << min(15, (16 * b + 15) // 6)) - 1) & ~((1 << ((16 * b) // 6)) - 1) for b in range(6)])


def blocks(m):
    return (m[:, None] & BITS[None]) != 0


for W in (0, 128, 512, 1 << 30):
    BA.QUAD_SORT_WINDOW = W
    prob, _, _ = BA.compile_problem(T(pts0), T(ext0), T(K0), T(sc.tracks), T(sc.mask), T(extra0), True, "SIMPLE_RADIAL",
                                    camera_split=True)
    ent = prob.entries.numpy()
    off = ent[:, 1] != ent[:, 2]
    q = ent[off, 3].astype(np.int64)
    # own patterns of the entries: presence masks of their segments
    slot = prob.obs_slot.numpy()
    seg_mask = np.zeros(prob.num_segments, np.int64)
    np.bitwise_or.at(seg_mask, slot // 16, 1 << (slot % 16))
    oa, ob = blocks(seg_mask[ent[off, 1]]), blocks(seg_mask[ent[off, 2]])
    ra, rb = blocks(q & 0xFFFF), blocks(q >> 16)
    cells = ra[::4, :, None] & rb[::4, None, :]                   # one row per quad (entries of a quad share the union)
    tot = cells.sum((1, 2))

    def busiest(assign):
        c = np.zeros((len(cells), 4), int)
        for i in range(6):
            for j in range(6):
                c[:, assign(i, j)] += cells[:, i, j]
        return c.max(1).mean()
    print(f"window {W if W < 1 << 30 else 'tile':>5}: blocks per side own {oa.sum(1).mean():.2f} / {ob.sum(1).mean():.2f}, "
          f"union {ra.sum(1).mean():.2f} / {rb.sum(1).mean():.2f}; sub-tiles per quad {tot.mean():.2f} "
          f"(own {(oa.sum(1) * ob.sum(1)).mean():.2f}); busiest wavefront: parity grid {busiest(lambda i, j: (i % 2) * 2 + j % 2):.2f}, "
          f"(2i+j)%4 {busiest(lambda i, j: (2 * i + j) % 4):.2f}, ideal {np.ceil(tot / 4).mean():.2f}")
