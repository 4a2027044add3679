"""Same-box A/B of host-side launch choices on the bench workload (BASELINE configs[2] by default): the scene is built once,
every variant = a set of environment switches read by vggsfm_amd.ba.compile_problem (VGGSFM_TILE_BACKFILL, VGGSFM_TILE_WGS,
...) -> ms per LM iteration + per-kernel HIP-event times, variants interleaved over `--rounds` rounds (drift shows up as a
difference between the rounds of one variant).  A different library (built with other -D switches) is selected for the
whole process with VGGSFM_AMD_LIB.
A variant key RCCL=1 (RCCL=2) runs the iteration with the collectives of a ONE-RANK "nccl" communicator between the phases
(reduce-scatter + all-gather form; 2 = the two-all-reduce form; 3 = the split exchange, phases 7..11): what the exchange sequence costs on the kernel stream before
any link time.  --tracks overrides the track count (37500 = one rank's shard of configs[3] at 8 GPUs).
usage: python scripts/prof/ab_c3.py [--workload c3] [--tracks N] [--steps 25] [--rounds 2] name:ENV=VAL,ENV=VAL ..."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
from vggsfm_amd import _lib  # noqa: E402
from vggsfm_amd import ba as BA  # noqa: E402
from vggsfm_amd.ba_options import BundleAdjustmentOptions  # noqa: E402
from vggsfm_amd.dist import ShardedBA  # noqa: E402
from vggsfm_amd.scene import make_scene, make_scene_device, perturb_for_ba  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--steps", type=int, default=25)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--tracks", type=int, default=0)
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    S, N, cam, shared = B.WORKLOADS[a.workload]
    N = a.tracks or N
    if S >= 400:
        sc = make_scene_device(S, N, cam, shared_camera=shared, seed=0, track_seed=1000, device=dev)
        ext0, K0, xp0, _ = perturb_for_ba(B.sc_cameras_only(sc), seed=0)
        args = (sc.points3D_init, B.D(ext0, dev), B.D(K0, dev), sc.tracks, sc.mask, B.D(xp0, dev))
    else:
        sc = make_scene(S, N, cam, shared_camera=shared, seed=0, track_seed=1000)
        ext0, K0, xp0, pts0 = perturb_for_ba(sc, seed=0)
        args = tuple(B.D(x, dev) for x in (pts0, ext0, K0, sc.tracks, sc.mask, xp0))
    base_env = dict(os.environ)
    for rnd in range(a.rounds):
        for v in a.variants:
            name, _, envs = v.partition(":")
            os.environ.clear()
            os.environ.update(base_env)
            os.environ["VGGSFM_AMD_DEBUG_HOOKS"] = "1"       # (the product path reads the VGGSFM_* switches only behind this gate)
            L.vgg_ba_set_tile_rhs(2)
            L.vgg_ba_set_step_from_factors(0)
            L.vgg_ba_set_tile_dma(0)
            coll, split = None, False
            for kv in filter(None, envs.split(",")):
                k, _, val = kv.partition("=")
                if k == "TILE_RHS":                        # (process-wide library switch, not an environment variable here)
                    L.vgg_ba_set_tile_rhs(int(val))
                    continue
                if k == "TILE_DMA":                        # (LDS-DMA staging of the off-diagonal tile launch, round-6 A/B)
                    L.vgg_ba_set_tile_dma(int(val))
                    continue
                if k == "STEP_FACTORS":
                    L.vgg_ba_set_step_from_factors(int(val))
                    continue
                if k == "RCCL":
                    import torch.distributed as dist
                    from vggsfm_amd.dist import Collectives
                    if not dist.is_initialized():
                        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
                        torch.cuda.set_device(dev)
                        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
                    coll = Collectives(1, plain_all_reduce=(val == "2"))
                    split = val == "3"                     # (RCCL=3: the split exchange of round 6, phases 7..11)
                    continue
                os.environ[k] = val.replace(";", ",")
            prob, _, _ = BA.compile_problem(*args, shared, cam, camera_split=True)
            init = [t.clone() for t in (prob.cam_q, prob.cam_t, prob.intr, prob.pts)]
            opts = BundleAdjustmentOptions()
            so = opts.solver_options
            so.max_num_iterations = B.EPISODE
            so.function_tolerance = so.gradient_tolerance = so.parameter_tolerance = -1.0
            solver = ShardedBA(prob, opts, 0, 1, collectives=coll, split_exchange=split)
            assert solver._split == split
            cnt = [0]

            def run(n):
                for _ in range(n):
                    if cnt[0] % B.EPISODE == 0:
                        for dst, src in zip((prob.cam_q, prob.cam_t, prob.intr, prob.pts), init):
                            dst.copy_(src)
                        solver.begin()
                    solver.iteration()
                    cnt[0] += 1
            run(a.warmup)
            torch.cuda.synchronize()
            _lib.check(L.vgg_ba_profile(1, (a.steps + 4) * int(prob.batch_desc.shape[0])), "profile")
            t0 = time.perf_counter()
            run(a.steps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            fin = solver.finish()
            kms = {}
            for kid, kname in enumerate(B.KERNELS):
                tot, n = ctypes.c_double(), ctypes.c_int()
                _lib.check(L.vgg_ba_profile_read(kid, ctypes.byref(tot), ctypes.byref(n), 1), "read")
                kms[kname] = round(tot.value / a.steps, 4)
            L.vgg_ba_profile(0, 0)
            print(json.dumps(dict(variant=name, round=rnd, ms_per_iteration=round(1e3 * dt / a.steps, 4), it_per_s=round(a.steps / dt, 1),
                                  chunks=int(prob.chunk_desc.shape[0]), merged=bool(prob.merged_tile_launch), kernel_ms=kms,
                                  final_cost=fin["final_cost"])), flush=True)
            del solver, prob
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
