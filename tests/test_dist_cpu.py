"""N>1 path on CPU: world_size-2 gloo processes exercise the point sharding and the collective that the
GPU path performs every LM iteration (SUM all-reduce of the camera-side blocks of each rank's shard)."""
import ctypes
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ba as OB
from vggsfm_amd import ba as BA
from vggsfm_amd.dist import partition_points, shard_slice
from vggsfm_amd.scene import make_scene, perturb_for_ba


def test_partition_balanced_and_exhaustive():
    torch.manual_seed(0)
    counts = torch.randint(2, 80, (5000,))
    for w in (1, 2, 3, 8):
        b = partition_points(counts, w)
        assert b[0] == 0 and b[-1] == 5000 and (b[1:] >= b[:-1]).all()
        loads = torch.stack([counts[b[r]:b[r + 1]].sum() for r in range(w)]).double()
        assert float(loads.max() / loads.mean()) < 1.02
    # degenerate: fewer points than ranks
    b = partition_points(torch.tensor([5, 5]), 4)
    assert b[0] == 0 and b[-1] == 2 and (b[1:] >= b[:-1]).all()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _camera_side(sc, ext0, K0, extra0, pts0, tracks, masks, cam_type):
    """camera-side gradient sum_i F_i^T r_i and cost of a (shard of a) problem, through the oracle's
    per-observation residual / Jacobian -- the content of reduce buffer 0."""
    L = OB.lib()
    S = len(ext0)
    q = np.ascontiguousarray(OB.rotmat_to_quat(ext0[:, :, :3]))
    t = np.ascontiguousarray(ext0[:, :, 3])
    g = np.zeros((S, 8))
    cost = 0.0
    c = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for s in range(S):
        intr = np.array([K0[s, 0, 0], K0[s, 0, 2], K0[s, 1, 2], 0.0 if extra0 is None else extra0[s, 0]])
        for p in np.nonzero(masks[s])[0]:
            r = np.zeros(2); Jp = np.zeros(12); Ji = np.zeros(4); Jx = np.zeros(6)
            X = np.ascontiguousarray(pts0[p])
            L.bao_obs_eval(ctypes.c_int(OB.MODEL[cam_type]), c(q[s]), c(t[s]), c(intr), c(X),
                           ctypes.c_double(float(tracks[s, p, 0])), ctypes.c_double(float(tracks[s, p, 1])), c(r), c(Jp),
                           c(Ji), c(Jx))
            F = np.concatenate([Jp.reshape(2, 6), Ji.reshape(2, 2)], 1)
            g[s] += F.T @ r
            cost += float(r @ r)
    return g, cost


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cam_type = "SIMPLE_RADIAL"
    sc = make_scene(6, 90, cam_type, shared_camera=False, seed=5)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=5)
    T = lambda x: torch.from_numpy(np.ascontiguousarray(x))
    tr, mk, pt, (lo, hi) = shard_slice(T(sc.tracks), T(sc.mask), T(pts0), rank, world)
    # the shard compiles into a well-formed device problem (host logic runs on CPU tensors too)
    prob, valid_idx, deleted = BA.compile_problem(pt, T(ext0), T(K0), tr, mk, T(extra0), False, cam_type)
    n_obs = torch.tensor([prob.num_obs], dtype=torch.int64)
    cam_counts = (prob.col_ptr[1:] - prob.col_ptr[:-1]).to(torch.int64)
    g, cost = _camera_side(sc, ext0, K0, extra0, pt.numpy(), tr.numpy(), mk.numpy(), cam_type)
    buf = torch.from_numpy(np.concatenate([g.ravel(), [cost]]))
    dist.all_reduce(n_obs)
    dist.all_reduce(cam_counts)
    dist.all_reduce(buf)                       # what ShardedBA does with reduce buffer 0 over RCCL
    gmax = torch.tensor([float(np.abs(g).max())], dtype=torch.float64)
    dist.all_reduce(gmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        out.put((int(n_obs), cam_counts.numpy(), buf.numpy(), float(gmax), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_sharded_camera_blocks_sum_to_global():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    n_obs, cam_counts, buf, gmax, _ = out.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sc = make_scene(6, 90, "SIMPLE_RADIAL", shared_camera=False, seed=5)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=5)
    assert n_obs == int(sc.mask.sum())
    assert np.array_equal(cam_counts, sc.mask.sum(1))
    g, cost = _camera_side(sc, ext0, K0, extra0, pts0, sc.tracks, sc.mask, "SIMPLE_RADIAL")
    np.testing.assert_allclose(buf[:-1].reshape(6, 8), g, rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(buf[-1], cost, rtol=1e-12)
    assert gmax <= np.abs(g).max() * 1.0000001 + 1e-12


def _split_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # rank 0 sees sliding-window tracks; rank 1 additionally owns ONE track that joins the first and the last cameras
    sc = make_scene(200, 1500, "SIMPLE_RADIAL", shared_camera=True, seed=3, track_seed=100 + rank)
    mask = torch.from_numpy(sc.mask.copy())
    if rank == 1:
        mask[:, 0] = False
        mask[[0, 199], 0] = True
    alone = BA.find_camera_split(mask)
    reduced = BA.find_camera_split(mask, adjacency_reduce=lambda t: dist.all_reduce(t, op=dist.ReduceOp.MAX))
    out.put((rank, None if alone[0] is None else alone[0].tolist(), alone[1],
             None if reduced[0] is None else reduced[0].tolist(), reduced[1]))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_camera_split_is_agreed_between_ranks():
    """The camera ordering [A, B, rest] must hold for the SUM of the ranks' reduced systems: every rank takes the
    MAX-all-reduced group adjacency, so both end up with the same answer -- here "no split", because rank 1 owns a track
    that couples the first and the last cameras, although rank 0 alone would have split."""
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, alone0, split0, red0, rsplit0), (_, alone1, split1, red1, rsplit1) = res
    assert alone0 is not None and split0[0] % 64 == 0 and alone1 is None
    assert red0 == red1 and rsplit0 == rsplit1 == (0, 0) and red0 is None


def _collectives_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vggsfm_amd.dist import Collectives
    res = {}
    for M in (1, 7, 1000, 4097):                     # payloads that do and do not divide by the world size
        g = torch.Generator().manual_seed(100 * M + rank)
        local = torch.randn(M, dtype=torch.float64, generator=g)
        for plain in (False, True):
            co = Collectives(world, plain_all_reduce=plain)
            packed, mx = local.clone(), torch.tensor([float(rank + 1) * 0.5 + M], dtype=torch.float64)
            co.system_(packed, mx)
            co.system_(packed.clone(), mx.clone())    # (buffers are reused across calls)
            res[(M, plain)] = (packed.numpy().copy(), float(mx))
        four = torch.full((4,), float(rank + 1), dtype=torch.float64)
        Collectives(world).sum_(four)
        res[(M, "sum")] = four.numpy().copy()
    out.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world3_reduce_scatter_all_gather_equals_all_reduce():
    """dist.Collectives.system_: reduce-scatter + all-gather of the padded payload with the local maxima riding in the
    gather = SUM all-reduce + MAX all-reduce, on every rank, for payload sizes that are not multiples of the world size --
    the collective of the GPU path (RCCL), run here over gloo with three processes."""
    world = 3
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_collectives_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(out.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for M in (1, 7, 1000, 4097):
        expect = sum(torch.randn(M, dtype=torch.float64, generator=torch.Generator().manual_seed(100 * M + r)).numpy()
                     for r in range(world))
        for r in range(world):
            for plain in (False, True):
                packed, mx = res[r][(M, plain)]
                np.testing.assert_allclose(packed, expect, rtol=1e-13, atol=1e-13)
                assert mx == world * 0.5 + M
            assert np.array_equal(res[r][(M, "sum")], np.full(4, 6.0))
        # every rank holds the same bits (the sum is formed once per slice and distributed)
        for r in range(1, world):
            assert np.array_equal(res[r][(M, False)][0], res[0][(M, False)][0])


def test_chunk_ranges_cover_every_chunk_once():
    from vggsfm_amd.dist import chunk_ranges
    from vggsfm_amd.utils.triangulation import reference_chunks
    assert reference_chunks(200, 100000) == (4000, 25) and reference_chunks(8, 6000) == (6000, 1)
    assert reference_chunks(400, 300000)[1] == 147                       # serial chunks at configs[3]
    for nc in (1, 7, 25, 147):
        for w in (1, 2, 8, 16):
            b = chunk_ranges(nc, w)
            assert b[0] == 0 and b[-1] == nc and all(x <= y for x, y in zip(b, b[1:])) and len(b) == w + 1
            assert max(y - x for x, y in zip(b, b[1:])) - min(y - x for x, y in zip(b, b[1:])) <= 1


def _rng_worker(rank, world, port, out):
    """triangulate_tracks_sharded's RNG contract: for the duration of the call every rank draws from rank 0's
    stream; afterwards rank 0 is where a single-rank run would be and every other rank is back on its OWN stream."""
    from vggsfm_amd.dist import _adopt_rank0_rng
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                              # ranks arrive with different states
    expect_own = torch.get_rng_state().clone()
    own = _adopt_rank0_rng(torch.device("cpu"))
    draw = torch.randperm(1000)                                # (what _draw_pairs does, once per reference chunk)
    if rank != 0:
        torch.set_rng_state(own)                               # what triangulate_tracks_sharded does behind the call
    after = torch.randperm(1000)
    out.put((rank, draw.numpy(), after.numpy(), bool(torch.equal(own, expect_own))))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_rank0_rng_is_borrowed_not_kept():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rng_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict()
    for _ in range(world):
        r, draw, after, own_ok = q.get(timeout=120)
        res[r] = (draw, after, own_ok)
    for p in procs:
        p.join(timeout=60)
    assert np.array_equal(res[0][0], res[1][0])                # the call's draws agree on both ranks
    assert res[0][2] and res[1][2]
    g0 = torch.Generator().manual_seed(100)
    assert np.array_equal(res[0][0], torch.randperm(1000, generator=g0).numpy())
    assert np.array_equal(res[0][1], torch.randperm(1000, generator=g0).numpy())      # rank 0: advanced by the draws
    g1 = torch.Generator().manual_seed(101)
    assert np.array_equal(res[1][1], torch.randperm(1000, generator=g1).numpy())      # rank 1: its own stream, untouched


def _split_exchange_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from vggsfm_amd.dist import Collectives
    co = Collectives(world, host_staging=False)
    res = {}
    for a, b in ((5, 3), (1000, 77), (4097, 4096)):
        ca, cb = -(-a // world), -(-b // world)
        g = torch.Generator().manual_seed(1000 * a + rank)
        pa, pb = torch.zeros(world * ca, dtype=torch.float64), torch.zeros(world * cb, dtype=torch.float64)
        pa[:a] = torch.randn(a, dtype=torch.float64, generator=g)
        pb[:b] = torch.randn(b, dtype=torch.float64, generator=g)
        ma, mb = torch.empty(ca, dtype=torch.float64), torch.empty(cb + 1, dtype=torch.float64)
        mb[cb] = float(rank) + 0.25
        ga, gb = torch.empty(world * ca, dtype=torch.float64), torch.empty(world * (cb + 1), dtype=torch.float64)
        # both parts in flight at once, waited for in order (what ShardedBA.iteration does around the diagonal tile launch)
        ha = co.system_scatter_begin(pa, ma, ga, rides=0)
        hb = co.system_scatter_begin(pb, mb, gb, rides=1)
        co.system_scatter_end(ha)
        co.system_scatter_end(hb)
        res[(a, b)] = (ga[:a].numpy().copy(), gb.view(world, cb + 1)[:, :cb].reshape(-1)[:b].numpy().copy(),
                       gb.view(world, cb + 1)[:, cb].numpy().copy())
    out.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world3_split_exchange_two_parts_in_flight():
    """dist.Collectives.system_scatter_begin / _end (round 6, the split exchange of the reduced system): two reduce-scatter +
    all-gather pairs begun back to back and ended afterwards (asynchronous on RCCL only: gloo does not order the pair) give every rank the sums of both parts and
    the riding per-rank maxima, for part sizes that do not divide by the world size."""
    world = 3
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_split_exchange_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(out.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for a, b in ((5, 3), (1000, 77), (4097, 4096)):
        ea, eb = 0, 0
        for r in range(world):
            g = torch.Generator().manual_seed(1000 * a + r)
            ea = ea + torch.randn(a, dtype=torch.float64, generator=g).numpy()
            eb = eb + torch.randn(b, dtype=torch.float64, generator=g).numpy()
        for r in range(world):
            ga, gb, rides = res[r][(a, b)]
            np.testing.assert_allclose(ga, ea, rtol=1e-13, atol=1e-13)
            np.testing.assert_allclose(gb, eb, rtol=1e-13, atol=1e-13)
            assert np.array_equal(rides, np.arange(world) + 0.25)
            assert np.array_equal(ga, res[0][(a, b)][0]) and np.array_equal(gb, res[0][(a, b)][1])
