"""The sharded BA path (ShardedBA phases + all-reduces) with two ranks.  Only one GPU is visible on the
test box, so the two ranks run one after the other inside ONE process in lock step, and the collectives are
emulated by summing / maxing their reduce buffers -- exactly the data an RCCL all-reduce would exchange.
The result must equal the single-rank solve of the whole problem."""
import numpy as np
import pytest
import torch

from vggsfm_amd import ba as BA
from vggsfm_amd.dist import ShardedBA, shard_slice
from vggsfm_amd.scene import make_scene, perturb_for_ba
from vggsfm_amd.utils.triangulation_helpers import prepare_ba_options

pytestmark = pytest.mark.gpu


def D(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()


class _LockStep:
    """emulates all_reduce across the ranks living in this process"""

    def __init__(self, world):
        self.world, self.pending = world, {}

    def exchange(self, tensors, op):
        stacked = torch.stack(tensors)
        red = stacked.max(0).values if op == "max" else stacked.sum(0)
        for t in tensors:
            t.copy_(red)


@pytest.mark.parametrize("in_place", [False, True, "split"])
@pytest.mark.parametrize("cam,shared,world", [("SIMPLE_RADIAL", True, 2), ("SIMPLE_PINHOLE", False, 3), ("SIMPLE_RADIAL", False, 2)])
def test_sharded_equals_single_rank(cam, shared, world, in_place, monkeypatch):
    """in_place: the exchange of the reduced system as dist.Collectives.system_scatter does it -- a reduce-scatter of the
    padded packed buffer (phase 4 pads it) into every rank's slice buffer and an all-gather of the slices, each with the
    rank's gradient maximum behind it, into the buffer phase 6 unpacks from (no staging copies, no MAX collective);
    emulated here slice by slice on the solvers' own reduce buffers 4 / 5 / 6.
    "split" (round 6): phases 7..11 -- the system exchanged in two parts (the off-diagonal tile launch's share before the
    diagonal launch has run), each part reduce-scattered and gathered on its own regions of the three buffers."""
    if in_place == "split":
        monkeypatch.setattr(BA, "MERGED_TILE_MAX_OBS", 0)      # (the split needs the two tile launches apart: not the small-problem form)
    sc = make_scene(24, 1500, cam, shared_camera=shared, seed=17)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=17)
    opts = prepare_ba_options()
    opts.solver_options.max_num_iterations = 12
    # single rank
    prob, _, _ = BA.compile_problem(D(pts0), D(ext0), D(K0), D(sc.tracks), D(sc.mask), D(extra0), shared, cam)
    ref = ShardedBA(prob, opts).solve()
    # `world` ranks in lock step
    solvers, problems, slices = [], [], []
    for r in range(world):
        tr, mk, pt, sl = shard_slice(D(sc.tracks), D(sc.mask), D(pts0), r, world)
        pr, _, _ = BA.compile_problem(pt, D(ext0), D(K0), tr, mk, D(extra0), shared, cam)
        problems.append(pr)
        slices.append(sl)
        solvers.append(ShardedBA(pr, opts, rank=r, world_size=world, all_reduce=lambda t, op: None,
                                 split_exchange="emulated" if in_place == "split" else False))
    hub = _LockStep(world)
    for s in solvers:
        s.begin()
    for _ in range(opts.solver_options.max_num_iterations + 1):
        for s in solvers:
            s._phase(0)
        hub.exchange([s.bufs[0] for s in solvers], "sum")
        n = ref["n_reduced"]
        if in_place == "split":
            assert all(s._split for s in solvers)

            def scatter_gather(padded, mine, gathered, rides):
                total = torch.stack(padded).sum(0)
                c = mine[0].numel() - rides
                for r, m in enumerate(mine):
                    m[:c].copy_(total[r * c:(r + 1) * c])
                allm = torch.cat(mine)
                for g in gathered:
                    g.copy_(allm)
            for s in solvers:
                s._phase(7)
                s._phase(8)
            assert 0 < solvers[0]._padded_a.numel() < n * (n + 1) // 2
            scatter_gather([s._padded_a for s in solvers], [s._mine_a for s in solvers], [s._gathered_a for s in solvers], 0)
            for s in solvers:
                s._phase(9)
                s._phase(10)
            scatter_gather([s._padded_b for s in solvers], [s._mine_b for s in solvers], [s._gathered_b for s in solvers], 1)
            for s in solvers:
                s._phase(11)
                s._phase(2)
            hub.exchange([s.bufs[3] for s in solvers], "sum")
            for s in solvers:
                s._phase(3)
            continue
        for s in solvers:
            s._phase(1)
            s._phase(4)                                     # pack lower triangle + rhs (the real collective's payload)
        assert solvers[0].bufs[4].numel() == n * (n + 1) // 2 + n
        if in_place:
            chunk = -(-solvers[0].bufs[4].numel() // world)
            assert all(s._padded.numel() == world * chunk and s._mine.numel() == chunk + 1 for s in solvers)
            total = torch.stack([s._padded for s in solvers]).sum(0)            # reduce-scatter: rank r keeps slice r of the sum
            for r, s in enumerate(solvers):
                s._mine[:chunk].copy_(total[r * chunk:(r + 1) * chunk])
            allm = torch.cat([s._mine for s in solvers])                         # all-gather of the (chunk + 1)-element slices
            for s in solvers:
                s._gathered.copy_(allm)
                s._phase(6)
        else:
            hub.exchange([s.bufs[4] for s in solvers], "sum")
            for s in solvers:
                s._phase(5)
            hub.exchange([s.bufs[2] for s in solvers], "max")
        for s in solvers:
            s._phase(2)
        hub.exchange([s.bufs[3] for s in solvers], "sum")
        for s in solvers:
            s._phase(3)
    outs = [s.finish(20) for s in solvers]
    for o in outs:
        assert o["num_iterations"] == ref["num_iterations"] and o["termination"] == ref["termination"]
        assert abs(o["final_cost"] - ref["final_cost"]) <= 1e-9 * ref["final_cost"]
        for a, b in zip(o["iterations"], ref["iterations"]):
            assert a["successful"] == b["successful"] and abs(a["cost"] - b["cost"]) <= 1e-9 * b["cost"]
    # replicated cameras agree on every rank and with the single-rank solve; points agree shard by shard
    for pr in problems:
        np.testing.assert_allclose(pr.cam_q.cpu().numpy(), prob.cam_q.cpu().numpy(), atol=1e-9)
        np.testing.assert_allclose(pr.cam_t.cpu().numpy(), prob.cam_t.cpu().numpy(), atol=1e-9)
        np.testing.assert_allclose(pr.intr.cpu().numpy(), prob.intr.cpu().numpy(), rtol=1e-10)
    got = torch.cat([pr.pts for pr in problems]).cpu().numpy()
    np.testing.assert_allclose(got, prob.pts.cpu().numpy(), atol=1e-8)


def test_sharded_solve_stops_enqueuing_after_termination():
    """ShardedBA.solve polls the device-side `done` flag every POLL iterations (vgg_ba_poll_done), like vgg_ba_solve: a
    solve that converges after a dozen iterations must not enqueue the remaining ~90 (with several ranks: their
    all-reduces) -- and must return exactly what the single-call solver returns."""
    sc = make_scene(16, 1200, "SIMPLE_RADIAL", shared_camera=True, seed=23, outlier_frac=0.0)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=23)
    from vggsfm_amd.ba_options import BundleAdjustmentOptions
    opts = BundleAdjustmentOptions()                       # 100 iterations, gradient tolerance 1e-4
    prob, _, _ = BA.compile_problem(D(pts0), D(ext0), D(K0), D(sc.tracks), D(sc.mask), D(extra0), True, "SIMPLE_RADIAL")
    ref, _ = BA.solve(prob, opts)
    prob2, _, _ = BA.compile_problem(D(pts0), D(ext0), D(K0), D(sc.tracks), D(sc.mask), D(extra0), True, "SIMPLE_RADIAL")
    s = ShardedBA(prob2, opts)
    calls = [0]
    inner = s.iteration

    def counted():
        calls[0] += 1
        inner()
    s.iteration = counted
    out = s.solve()
    assert ref["num_iterations"] < 40 and ref["termination"] != 0
    assert out["num_iterations"] == ref["num_iterations"] and out["termination"] == ref["termination"]
    assert out["final_cost"] == ref["final_cost"]
    assert calls[0] <= ref["num_iterations"] + 1 + ShardedBA.POLL
    assert torch.equal(prob.pts, prob2.pts) and torch.equal(prob.cam_t, prob2.cam_t)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_triangulation_and_filter_equal_single_rank(world):
    """dist.triangulate_tracks_sharded / filter_all_points3D_sharded: the ranks share the work by whole reference chunks
    (every rank consumes the RNG draws of all chunks) resp. by point ranges; concatenated in rank order the results are the
    single-rank results BIT FOR BIT (inlier counts, masks, points).  The ranks run one after the other in this process
    (gather=False: no communicator needed)."""
    from vggsfm_amd import dist as DI
    from vggsfm_amd.utils import triangulation as T
    from vggsfm_amd.utils import triangulation_helpers as H
    sc = make_scene(30, 700, "SIMPLE_RADIAL", shared_camera=True, seed=41, outlier_frac=0.1)
    ext, K, xp = D(sc.extrinsics), D(sc.intrinsics), D(sc.extra_params)
    tracks, vis, score = D(sc.tracks), D(sc.vis), D(sc.score)
    tn = H.cam_from_img(tracks, K, xp)
    kw = dict(max_tri_points_num=30 * 100)                           # 7 reference chunks of 100 tracks
    torch.manual_seed(5)
    p_ref, n_ref, m_ref = T.triangulate_tracks(ext, tn, track_vis=vis, track_score=score, **kw)
    parts = []
    for r in range(world):
        torch.manual_seed(5)                                          # every rank is seeded alike
        p, n, m, (lo, hi) = DI.triangulate_tracks_sharded(ext, tn, r, world, track_vis=vis, track_score=score, gather=False, **kw)
        assert p.shape[0] == hi - lo
        parts.append((p, n, m, lo, hi))
    assert parts[0][3] == 0 and parts[-1][4] == 700 and all(a[4] == b[3] for a, b in zip(parts, parts[1:]))
    assert torch.equal(torch.cat([a[1] for a in parts]), n_ref) and torch.equal(torch.cat([a[2] for a in parts]), m_ref)
    assert torch.equal(torch.cat([a[0] for a in parts]), p_ref)
    # more ranks than chunks: the surplus ranks get an empty range
    torch.manual_seed(5)
    p, n, m, rng = DI.triangulate_tracks_sharded(ext, tn, 0, 16, track_vis=vis, track_score=score, gather=False, **kw)
    assert p.shape[0] == rng[1] - rng[0] == 0
    # filter
    fm, fd = H.filter_all_points3D(p_ref, tracks, ext, K, xp, max_reproj_error=4, return_detail=True)
    masks, dets = [], []
    for r in range(world):
        (mk, dt), _ = DI.filter_all_points3D_sharded(p_ref, tracks, ext, K, r, world, extra_params=xp, gather=False,
                                                     max_reproj_error=4, return_detail=True)
        masks.append(mk), dets.append(dt)
    assert torch.equal(torch.cat(masks), fm) and torch.equal(torch.cat(dets, 1), fd)


@pytest.mark.parametrize("cam,shared,frames", [("SIMPLE_RADIAL", False, 56), ("SIMPLE_PINHOLE", False, 40), ("SIMPLE_RADIAL", True, 72),
                                               ("SIMPLE_PINHOLE", True, 33)])
def test_split_exchange_rebuilds_the_same_system_as_the_one_piece_exchange(cam, shared, frames, monkeypatch):
    """Phases 7..11 (include/vggsfm_amd.h) against phases 1 / 4 / 6 with two emulated ranks, over several LM iterations in lock
    step: S | rhs and the gradient maximum that come out of the exchange are BIT-identical -- every element of the lower
    triangle is in exactly one of the two parts, the off-diagonal launch fills nothing of part B and the diagonal launch /
    assemble nothing of part A (3-5 camera groups, the last one partial; 6 x 6, 7 x 7 and 8 x 8 blocks; odd n)."""
    monkeypatch.setattr(BA, "MERGED_TILE_MAX_OBS", 0)
    W = 2
    sc = make_scene(frames, 1200, cam, shared_camera=shared, seed=5)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=5)
    opts = prepare_ba_options()
    snaps = {}

    def sg(padded, mine, gathered, rides):
        total = torch.stack(padded).sum(0)
        c = mine[0].numel() - rides
        for r, m in enumerate(mine):
            m[:c].copy_(total[r * c:(r + 1) * c])
        allm = torch.cat(mine)
        for g in gathered:
            g.copy_(allm)
    for mode in ("one_piece", "split"):
        solvers = []
        for r in range(W):
            tr, mk, pt, _ = shard_slice(D(sc.tracks), D(sc.mask), D(pts0), r, W)
            pr, _, _ = BA.compile_problem(pt, D(ext0), D(K0), tr, mk, D(extra0), shared, cam)
            solvers.append(ShardedBA(pr, opts, rank=r, world_size=W, all_reduce=lambda t, op: None,
                                     split_exchange="emulated" if mode == "split" else False))
        assert all(s._split == (mode == "split") for s in solvers)
        for s in solvers:
            s.begin()
        snaps[mode] = []
        for _ in range(5):
            for s in solvers:
                s._phase(0)
            tot = torch.stack([s.bufs[0] for s in solvers]).sum(0)
            for s in solvers:
                s.bufs[0].copy_(tot)
            if mode == "one_piece":
                for s in solvers:
                    s._phase(1)
                    s._phase(4)
                sg([s._padded for s in solvers], [s._mine for s in solvers], [s._gathered for s in solvers], 1)
                for s in solvers:
                    s._phase(6)
            else:
                for s in solvers:
                    s._phase(7)
                    s._phase(8)
                a, n = solvers[0]._padded_a.numel(), solvers[0].bufs[4].numel()
                assert 0 < a < n
                sg([s._padded_a for s in solvers], [s._mine_a for s in solvers], [s._gathered_a for s in solvers], 0)
                for s in solvers:
                    s._phase(9)
                    s._phase(10)
                sg([s._padded_b for s in solvers], [s._mine_b for s in solvers], [s._gathered_b for s in solvers], 1)
                for s in solvers:
                    s.bufs[1].fill_(float("nan"))           # (phase 11 rebuilds everything that is read)
                    s._phase(11)
            for s in solvers:
                nn = int(round((-1 + (1 + 4 * s.bufs[1].numel()) ** 0.5) / 2))
                snaps[mode].append((torch.tril(s.bufs[1][:nn * nn].view(nn, nn)).clone(), s.bufs[1][nn * nn:].clone(), s.bufs[2][:1].clone()))
            for s in solvers:
                s._phase(2)
            tot = torch.stack([s.bufs[3] for s in solvers]).sum(0)
            for s in solvers:
                s.bufs[3].copy_(tot)
            for s in solvers:
                s._phase(3)
    for (Sa, ra, ga), (Sb, rb, gb) in zip(snaps["one_piece"], snaps["split"]):
        assert torch.equal(Sa, Sb) and torch.equal(ra, rb) and torch.equal(ga, gb)
