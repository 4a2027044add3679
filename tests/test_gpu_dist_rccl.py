"""The sharded BA path over a REAL RCCL communicator: one process per GPU, world size 2 (or 4 / 8 when the box has them),
backend "nccl", the collectives of vggsfm_amd.dist.Collectives (all-reduce of the camera blocks, reduce-scatter +
all-gather of the packed reduced system, all-reduce of the step scalars) -- against the single-rank solve of the same
problem.  Skipped on a box with one GPU (the builder's box has one: the 8-GPU box of the driver runs it)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, cam, shared, out):
    import torch.distributed as dist

    from vggsfm_amd import ba as BA
    from vggsfm_amd.dist import ShardedBA, shard_slice
    from vggsfm_amd.scene import make_scene, perturb_for_ba
    from vggsfm_amd.utils.triangulation_helpers import prepare_ba_options
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    sc = make_scene(160, 6000, cam, shared_camera=shared, seed=17)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=17)
    opts = prepare_ba_options()
    opts.solver_options.max_num_iterations = 12
    tr, mk, pt, sl = shard_slice(D(sc.tracks), D(sc.mask), D(pts0), rank, world)
    reduce_adj = lambda t: dist.all_reduce(t, op=dist.ReduceOp.MAX)
    pr, _, _ = BA.compile_problem(pt, D(ext0), D(K0), tr, mk, D(extra0), shared, cam, camera_split=True, adjacency_reduce=reduce_adj)
    res = ShardedBA(pr, opts, rank=rank, world_size=world).solve()
    torch.cuda.synchronize()
    payload = dict(rank=rank, slice=sl, its=[(i["successful"], i["cost"]) for i in res["iterations"]], final=res["final_cost"],
                   n_it=res["num_iterations"], cam_t=pr.cam_t.cpu().numpy(), intr=pr.intr.cpu().numpy(), pts=pr.pts.cpu().numpy(),
                   perm=None if pr.cam_perm is None else pr.cam_perm.cpu().numpy())
    if rank == 0:
        full, _, _ = BA.compile_problem(D(pts0), D(ext0), D(K0), D(sc.tracks), D(sc.mask), D(extra0), shared, cam, camera_split=True)
        ref = ShardedBA(full, opts).solve()
        payload["ref"] = dict(its=[(i["successful"], i["cost"]) for i in ref["iterations"]], final=ref["final_cost"],
                              n_it=ref["num_iterations"], cam_t=full.cam_t.cpu().numpy(), intr=full.intr.cpu().numpy(),
                              pts=full.pts.cpu().numpy(), perm=None if full.cam_perm is None else full.cam_perm.cpu().numpy())
    out.put(payload)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL with one rank per GPU)")
@pytest.mark.parametrize("cam,shared", [("SIMPLE_RADIAL", True), ("SIMPLE_RADIAL", False)])
def test_rccl_sharded_solve_equals_single_rank(cam, shared):
    world = min(torch.cuda.device_count(), 8)
    world = 1 << (world.bit_length() - 1)                    # 2, 4 or 8 ranks
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, world, port, cam, shared, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((out.get(timeout=600) for _ in range(world)), key=lambda d: d["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    ref = got[0]["ref"]
    for g in got:
        assert g["n_it"] == ref["n_it"]
        assert abs(g["final"] - ref["final"]) <= 1e-9 * ref["final"]
        for (sa, ca), (sb, cb) in zip(g["its"], ref["its"]):
            assert sa == sb and abs(ca - cb) <= 1e-9 * cb
        # replicated cameras: identical bits on every rank (same sums, same decisions), equal to the single-rank solve
        assert np.array_equal(g["cam_t"], got[0]["cam_t"]) and np.array_equal(g["intr"], got[0]["intr"])
        assert (g["perm"] is None) == (ref["perm"] is None) and (g["perm"] is None or np.array_equal(g["perm"], ref["perm"]))
        np.testing.assert_allclose(g["cam_t"], ref["cam_t"], atol=1e-9)
        np.testing.assert_allclose(g["intr"], ref["intr"], rtol=1e-10)
    pts = np.concatenate([g["pts"] for g in got])
    np.testing.assert_allclose(pts, ref["pts"], atol=1e-8)


def _one_rank(port, cam, shared, out):
    import torch.distributed as dist

    from vggsfm_amd import ba as BA
    from vggsfm_amd.dist import Collectives, ShardedBA, triangulate_tracks_sharded
    from vggsfm_amd.scene import make_scene, perturb_for_ba
    from vggsfm_amd.utils.triangulation_helpers import prepare_ba_options
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    sc = make_scene(96, 5000, cam, shared_camera=shared, seed=23)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=23)
    opts = prepare_ba_options()
    opts.solver_options.max_num_iterations = 10
    res = {}
    for name in ("plain", "rccl", "rccl_allreduce", "rccl_split"):
        BA.MERGED_TILE_MAX_OBS = 0 if name == "rccl_split" else 1_000_000     # (the split needs the two tile launches apart)
        pr, _, _ = BA.compile_problem(D(pts0), D(ext0), D(K0), D(sc.tracks), D(sc.mask), D(extra0), shared, cam, camera_split=True)
        co = None if name == "plain" else Collectives(1, plain_all_reduce=(name == "rccl_allreduce"))
        sb = ShardedBA(pr, opts, collectives=co, split_exchange=(name == "rccl_split"))
        assert sb._split == (name == "rccl_split")
        r = sb.solve()
        torch.cuda.synchronize()
        res[name] = dict(its=[(i["successful"], i["cost"]) for i in r["iterations"]], final=r["final_cost"], n_it=r["num_iterations"],
                         cam_t=pr.cam_t.cpu().numpy(), intr=pr.intr.cpu().numpy(), pts=pr.pts.cpu().numpy())
    out.put(res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("cam,shared", [("SIMPLE_RADIAL", True), ("SIMPLE_RADIAL", False)])
def test_rccl_one_rank_runs_the_exchange_sequence_on_the_solver_buffers(cam, shared):
    """What a 1-GPU box can show of the RCCL path: a REAL "nccl" communicator of one rank takes the solver's own buffers
    through every collective of an LM iteration -- all-reduce of the camera blocks, reduce-scatter of the zero-padded packed
    system into the slice buffer, all-gather into the buffer the unpack kernel reads, all-reduce of the step scalars -- on
    the kernel stream, between the phase launches.  With one rank every sum is the identity, so the solve has to reproduce
    the collective-free one bit for bit; the two-all-reduce form (A/B switch) as well."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_one_rank, args=(_free_port(), cam, shared, out))
    p.start()
    res = out.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    ref = res["plain"]
    # ("rccl_split", round 6: the system in two parts, the first one's reduce-scatter + all-gather enqueued on the communicator's
    #  stream before the diagonal tile launch -- async_op -- and waited for behind the second's)
    for name in ("rccl", "rccl_allreduce", "rccl_split"):
        g = res[name]
        assert g["n_it"] == ref["n_it"] and g["its"] == ref["its"] and g["final"] == ref["final"], name
        for k in ("cam_t", "intr", "pts"):
            assert np.array_equal(g[k], ref[k]), (name, k)
