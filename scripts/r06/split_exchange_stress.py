import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from vggsfm_amd import ba as BA
from vggsfm_amd.dist import ShardedBA, shard_slice
from vggsfm_amd.scene import make_scene, perturb_for_ba
from vggsfm_amd.utils.triangulation_helpers import prepare_ba_options
D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
BA.MERGED_TILE_MAX_OBS = 0
def sg(padded, mine, gathered, rides):
    total = torch.stack(padded).sum(0); c = mine[0].numel() - rides
    for r, m in enumerate(mine): m[:c].copy_(total[r * c:(r + 1) * c])
    allm = torch.cat(mine)
    for g in gathered: g.copy_(allm)
bad = 0
for frames, cam, shared, W in ((16, "SIMPLE_RADIAL", True, 2), (17, "SIMPLE_RADIAL", False, 3), (31, "SIMPLE_PINHOLE", False, 2), (32, "SIMPLE_RADIAL", True, 4),
                               (47, "SIMPLE_PINHOLE", True, 3), (64, "SIMPLE_RADIAL", False, 2), (81, "SIMPLE_RADIAL", True, 5), (129, "SIMPLE_PINHOLE", False, 2)):
    sc = make_scene(frames, 900, cam, shared_camera=shared, seed=frames)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=frames)
    opts = prepare_ba_options()
    snaps = {}
    for mode in ("one", "split"):
        solvers = []
        for r in range(W):
            tr, mk, pt, _ = shard_slice(D(sc.tracks), D(sc.mask), D(pts0), r, W)
            pr, _, _ = BA.compile_problem(pt, D(ext0), D(K0), tr, mk, D(extra0), shared, cam)
            solvers.append(ShardedBA(pr, opts, rank=r, world_size=W, all_reduce=lambda t, op: None, split_exchange="emulated" if mode == "split" else False))
        if mode == "split" and not solvers[0]._split:
            print(frames, cam, shared, W, "split unavailable (expected for <= 16 cameras)"); snaps[mode] = None; continue
        for s in solvers: s.begin()
        out = []
        for it in range(3):
            for s in solvers: s._phase(0)
            tot = torch.stack([s.bufs[0] for s in solvers]).sum(0)
            for s in solvers: s.bufs[0].copy_(tot)
            if mode == "one":
                for s in solvers: s._phase(1); s._phase(4)
                sg([s._padded for s in solvers], [s._mine for s in solvers], [s._gathered for s in solvers], 1)
                for s in solvers: s._phase(6)
            else:
                for s in solvers: s._phase(7); s._phase(8)
                sg([s._padded_a for s in solvers], [s._mine_a for s in solvers], [s._gathered_a for s in solvers], 0)
                for s in solvers: s._phase(9); s._phase(10)
                sg([s._padded_b for s in solvers], [s._mine_b for s in solvers], [s._gathered_b for s in solvers], 1)
                for s in solvers: s.bufs[1].fill_(float("nan")); s._phase(11)
            s = solvers[-1]
            n = int(round((-1 + (1 + 4 * s.bufs[1].numel()) ** 0.5) / 2))
            out.append((torch.tril(s.bufs[1][:n * n].view(n, n)).clone(), s.bufs[1][n * n:].clone(), s.bufs[2][:1].clone()))
            for s in solvers: s._phase(2)
            tot = torch.stack([s.bufs[3] for s in solvers]).sum(0)
            for s in solvers: s.bufs[3].copy_(tot)
            for s in solvers: s._phase(3)
        snaps[mode] = out
    if snaps["split"] is None: continue
    ok = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) for a, b in zip(snaps["one"], snaps["split"]))
    print(frames, cam, shared, W, "identical" if ok else "DIFFERENT"); bad += (not ok)
print("bad", bad)
