#!/usr/bin/env python
"""Benchmark of the hot path BASELINE.json names: Levenberg-Marquardt bundle-adjustment iterations/s.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A *step* is ONE LM iteration (linearise -> block-sparse Schur complement -> dense Cholesky of the
reduced camera system -> back-substitution -> candidate cost -> trust-region decision) over the whole
synthetic scene.  Workload = BASELINE.json configs[2], the configuration the north-star target is quoted on: 200 frames x 100k
tracks PER GPU, SIMPLE_RADIAL, shared camera (SURVEY.md section 8d generator) -- at EVERY N, so that the driver's
N = 1, 2, 4, 8 lines draw one curve: `value` = N x K / t shard-iterations per second (weak scaling: every rank owns a
100k-track shard of the 200-frame x (N x 100k)-track problem, cameras replicated, per-camera blocks all-reduced and the
reduced system reduce-scattered + all-gathered over RCCL once per iteration); at N = 1 that is the LM-iterations/s of
configs[2].  The STRONG-scaling figure rides along as the key `strong_scaling_c4` at every N: ONE fixed problem, BASELINE
configs[3] (400 frames x 300k tracks, per-frame intrinsics), its tracks split over the ranks; at N > 1 rank 0 first times
the SAME whole problem alone (`n1_same_problem` inside that key, the other ranks wait), so the line carries its own
speed-up (`speedup_vs_n1`).  `--workload c4 --gpus N` makes the strong problem the main one instead (value = K / t).
Inputs are resident in HBM before the timed region; termination tests are disabled so that exactly K
iterations run (a solve is restarted from the initial state every EPISODE iterations, like the
reference's 50/100-iteration BA calls).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from vggsfm_amd import _lib  # noqa: E402
from vggsfm_amd import ba as BA  # noqa: E402
from vggsfm_amd.ba_options import BundleAdjustmentOptions  # noqa: E402
from vggsfm_amd.dist import ShardedBA  # noqa: E402
from vggsfm_amd.scene import make_scene, make_scene_device, perturb_for_ba  # noqa: E402

EPISODE = 25
FP64_PEAK_TFLOPS = 78.6      # MI355X FP64 vector = FP64 matrix peak (datasheet; MI355X_MICROARCH.md has no fp64 row)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md chip table (6.29 TB/s measured copy)
KERNELS = ["cam_pass<linearize>", "point_pass", "cam_pass<rhs>", "schur_tile<offdiag>", "cholesky", "point_step",
           "schur_tile<diag>"]
# rocprofv3 kernel-name fragments of the single-kernel entries (for the PMC traffic lookup)
ROCPROF_NAME = {"schur_tile<offdiag>": "schur_tile_kernel<6, false>", "schur_tile<diag>": "schur_tile_kernel<6, true>",
                "point_pass": "point_pass_kernel", "point_step": "point_step_kernel",
                "cam_pass<linearize>": "cam_pass_kernel<2, 0>", "cam_pass<rhs>": "cam_pass_kernel<2, 1>"}
PMC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic_c3.json")

WORKLOADS = {
    # name: (frames, tracks per GPU, camera, shared)
    "c2": (50, 20000, "SIMPLE_PINHOLE", False),
    "c3": (200, 100000, "SIMPLE_RADIAL", True),
    # c3 with every track visible in every frame (what iterative_global_BA solves on the synthetic scene, whose reprojection
    # filter keeps the -- accurate -- positions of the frames a track is "invisible" in): 19.9 M observations, every Schur tile
    # dense.  The same kernels, no block to skip: the upper end of what the tile formulation reaches
    "c3dense": (200, 100000, "SIMPLE_RADIAL", True),
    "c4shard": (400, 37500, "SIMPLE_RADIAL", False),
    "c4full": (400, 300000, "SIMPLE_RADIAL", False),      # whole configs[3] on ONE GPU (robustness / capacity check)
    "c4": (400, 300000, "SIMPLE_RADIAL", False),          # configs[3], the 300k tracks SPLIT over the ranks (strong scaling)
}
STRONG = {"c4"}
BASELINE_CONFIG = {"c2": "BASELINE configs[1]", "c3": "BASELINE configs[2]",
                   "c3dense": "BASELINE configs[2] with every track visible in every frame", "c4shard": "one 8-GPU shard of BASELINE configs[3]",
                   "c4full": "BASELINE configs[3] whole on one GPU", "c4": "BASELINE configs[3]"}
# the reference's own torch triangulate_tracks on the CPU, measured during the survey (unmodified code,
# 200 views x 1 000 tracks, all tracks visible, 8 cores, 130.7 s) -- quoted beside the kernel's number, never re-measured here
REFERENCE_TRIANGULATION_CPU = dict(value=1000 / 130.7, unit="tracks/s", cores=8, kind="reference",
                                   sample="vggsfm/utils/triangulation.py::triangulate_tracks, 200 views x 1 000 tracks, fp64, survey "
                                          "container (BASELINE.md section 2): 130.7 s = 131 ms per track")


def D(x, dev):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).to(dev)


def sc_cameras_only(sc):
    """perturb_for_ba on a device scene: only its (host) cameras are perturbed here, the points come perturbed."""
    import types
    return types.SimpleNamespace(S=sc.S, extrinsics=sc.extrinsics, intrinsics=sc.intrinsics, extra_params=sc.extra_params,
                                 shared_camera=sc.shared_camera, points3D=np.zeros((1, 3)))


def pmc_traffic(kernel, workload):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (scripts/prof/pmc_traffic.sh):
    FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE counts half of the bytes of streaming reads on gfx950
    (MI355X_MICROARCH.md "HBM"; confirmed for 4/8/16 B per lane by scripts/ubench/fetch_calib.hip)."""
    if workload != "c3" or not os.path.exists(PMC_FILE) or kernel not in ROCPROF_NAME:
        return None, None
    with open(PMC_FILE) as fh:
        pmc = json.load(fh)
    at = pmc.get("_measured_at", "round 1, commit 0f5c1d2 era (before the camera split and the dataflow factorisation; the "
                                 "tile kernels have not changed since)")
    for name, ctr in pmc.items():
        if isinstance(ctr, dict) and ROCPROF_NAME[kernel] in name and "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr:
            return (2.0 * ctr["FETCH_SIZE"]["mean"] + ctr["WRITE_SIZE"]["mean"]) * 1024.0, at
    return None, None


def trajectory_divergence(gpu_its, port_its):
    """Where two LM iteration logs part: the first iteration whose accept / reject decision differs or whose cost differs by
    more than 1e-9 relative, with both sides' step quality and cost change there, and whether that iteration already sits
    at the rounding-noise floor (|cost change| < 1e-12 cost: the step quality is a quotient of two rounding errors)."""
    for a, b in zip(gpu_its, port_its):
        if a["successful"] != b["successful"] or abs(a["cost"] - b["cost"]) > 1e-9 * abs(b["cost"]):
            return dict(iteration=int(b["iteration"]), gpu=dict(successful=bool(a["successful"]), cost_change=a["cost_change"],
                                                                relative_decrease=a["relative_decrease"]),
                        port=dict(successful=bool(b["successful"]), cost_change=b["cost_change"],
                                  relative_decrease=b["relative_decrease"]),
                        rel_cost_change_port=abs(b["cost_change"]) / abs(b["cost"]),
                        at_noise_floor=bool(abs(b["cost_change"]) < 1e-12 * abs(b["cost"])))
    return None


def last_iteration_above_noise(its, floor=1e-12):
    """Index of the last iteration whose cost change is still >= floor x cost (beyond it accept / reject is rounding)."""
    last = 0
    for it in its:
        if it["iteration"] > 0 and abs(it["cost_change"]) >= floor * abs(it["cost"]):
            last = int(it["iteration"])
    return last


def parity_vs_port(pts, ext, K, tracks, masks, extra, shared, cam_type, iters, what):
    """`iters` LM iterations (termination tests off) of ONE problem from ONE start, through the public GPU entry
    (vggsfm_amd.ba.bundle_adjustment -> vgg_ba_solve) and by the CPU port (oracle/ba_oracle.c behind oracle/ba.py) on host
    copies of the same tensors: accept / reject pattern, cost and radius per iteration, poses, points, focal lengths.
    Device tensors in, comparison dict out (the checker is imported here, after every timed region)."""
    from oracle import ba as OB
    o = BundleAdjustmentOptions()
    so = o.solver_options
    so.max_num_iterations = iters
    so.function_tolerance = so.gradient_tolerance = so.parameter_tolerance = -1.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gp, ge, gK, gx, gs = BA.bundle_adjustment(pts, ext, K, tracks, masks, None, extra, shared, cam_type, o)
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    H = lambda t: None if t is None else t.detach().cpu().numpy()
    t0 = time.perf_counter()
    op, oe, oK, ox, osum = OB.bundle_adjustment(H(pts), H(ext), H(K), H(tracks), H(masks).astype(bool), H(extra), shared, cam_type,
                                                options=OB.ceres_options(iters, -1.0, -1.0, -1.0))
    t_port = time.perf_counter() - t0
    ge, gp, gK = H(ge), H(gp), H(gK)
    relv = lambda a, b: float(np.max(np.linalg.norm((a - b).reshape(len(a), -1), axis=1)
                                     / np.maximum(np.linalg.norm(b.reshape(len(b), -1), axis=1), 1e-12)))
    its = list(zip(gs["iterations"], osum["iterations"]))
    alive = ~np.asarray(osum["deleted"], bool)
    return dict(workload=f"{what}, {iters} LM iterations from the same start, termination tests off",
                observations=int(H(masks)[:, np.asarray(osum["valid_idx"])].sum()), reduced_system=int(osum["n_reduced"]),
                lm_iterations_gpu=int(gs["num_iterations"]), lm_iterations_port=int(osum["num_iterations"]),
                accept_pattern_equal=bool(len(gs["iterations"]) == len(osum["iterations"])
                                          and all(a["successful"] == b["successful"] for a, b in its)),
                max_rel_cost_delta_per_iteration=float(max(abs(a["cost"] - b["cost"]) / b["cost"] for a, b in its)),
                max_rel_radius_delta_per_iteration=float(max(abs(a["radius"] - b["radius"]) / b["radius"] for a, b in its)),
                final_cost_rel_delta=abs(gs["final_cost"] - osum["final_cost"]) / osum["final_cost"],
                max_rel_rotation_delta=relv(ge[:, :, :3], oe[:, :, :3]),
                max_rel_translation_delta=relv(ge[1:, :, 3], oe[1:, :, 3]),
                max_rel_point_delta=relv(gp[alive], op[alive]),
                max_rel_focal_delta=float(np.max(np.abs(gK[:, 0, 0] / oK[:, 0, 0] - 1))),
                gpu_seconds=t_gpu, port_seconds=t_port, port_threads=int(OB.lib().bao_num_threads()),
                tolerance=1e-4, reference="oracle/ba_oracle.c (Ceres/COLMAP restatement; unpinned vs pycolmap)")


class StageTimer:
    """Synchronising timers around the stages of Triangulator.forward (profiling run only: the synchronisations serialise
    host and device, the untimed run next to it gives the wall time).  `top` stages are the calls forward itself makes --
    together they are the whole forward, what is left is `unaccounted`; `inner` are the device entries below them."""

    def __init__(self):
        self.top, self.inner, self._undo, self._depth = {}, {}, [], 0

    def _wrap(self, owner, name, table, is_top):
        fn = getattr(owner, name)
        timer = self

        def wrapper(*a, **k):
            if is_top:
                timer._depth += 1
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                torch.cuda.synchronize()
                if not is_top or timer._depth == 1:
                    table[name] = table.get(name, 0.0) + time.perf_counter() - t0
                if is_top:
                    timer._depth -= 1
        setattr(owner, name, wrapper)
        self._undo.append((owner, name, fn))

    def install(self):
        from vggsfm_amd.models import triangulator as TM
        from vggsfm_amd.utils import triangulation as TU
        for name in ("get_EFP", "cam_from_img", "triangulate_by_pair", "find_best_initial_pair", "init_BA", "init_refine_pose",
                     "refine_pose", "iterative_global_BA", "build_reconstruction"):
            self._wrap(TM, name, self.top, True)
        self._wrap(TM.Triangulator, "triangulate_tracks_and_BA", self.top, True)
        self._wrap(BA, "compile_problem", self.inner, False)
        self._wrap(BA, "solve", self.inner, False)
        for name in ("triangulate_tracks", "filter_all_points3D", "pose_refinement_batch", "cam_from_img", "project_3D_points"):
            for mod in (TU, TM):
                if hasattr(mod, name) and not (mod is TM and name in self.top):
                    self._wrap(mod, name, self.inner, False)
        return self

    def remove(self):
        for owner, name, fn in reversed(self._undo):
            setattr(owner, name, fn)
        self._undo = []


def pipeline_leg(dev, configs=("c2", "c3")):
    """End-to-end geometry stage: vggsfm_amd.models.Triangulator.forward -- what demo.py calls after
    the tracker (reference vggsfm/models/triangulator.py:44-363) -- on the synthetic scenes of BASELINE configs[1] / [2]:
    wall time of an untimed run, then the same call under stage timers."""
    import types
    from vggsfm_amd.models import Triangulator
    out = {}
    W = 1024
    for name in configs:
        S, N, cam, shared = WORKLOADS[name]
        sc = make_scene(S, N, cam, shared_camera=shared, seed=0)
        ext0, K0, _, _ = perturb_for_ba(sc, seed=0, rot_deg=0.5, trans=0.02, focal_rel=0.02)
        cams = types.SimpleNamespace(R=D(ext0[:, :, :3], dev).float(), T=D(ext0[:, :, 3], dev).float(),
                                     focal_length=torch.stack([D(K0[:, 0, 0] / (W / 2.0), dev).float()] * 2, -1))
        img = types.SimpleNamespace(shape=(1, S, 3, W, W))             # (only .shape is read when extract_color=False)
        tracks, vis, score = D(sc.tracks, dev)[None], D(sc.vis, dev)[None], D(sc.score, dev)[None]
        prelim = {"fmat_inlier_mask": D(sc.mask[1:] & sc.mask[0:1], dev)[None]}
        tri = Triangulator()
        kw = dict(pred_score=score, shared_camera=shared, camera_type=cam, BA_iters=2, robust_refine=2, extract_color=False)
        import contextlib
        import io
        quiet = lambda: contextlib.redirect_stdout(io.StringIO())      # (forward prints the reference's milestones)
        with quiet():
            tri(cams, tracks[:, :, :2000], vis[:, :, :2000], img, {"fmat_inlier_mask": prelim["fmat_inlier_mask"][:, :, :2000]},
                pred_score=score[:, :, :2000], shared_camera=shared, camera_type=cam, BA_iters=1, robust_refine=1,
                extract_color=False)                                    # warm-up: library load, allocator, first launches
            walls = []
            for _ in range(2):
                torch.manual_seed(0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = tri(cams, tracks, vis, img, prelim, **kw)
                torch.cuda.synchronize()
                walls.append(time.perf_counter() - t0)
            timer = StageTimer().install()
            try:
                torch.manual_seed(0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                tri(cams, tracks, vis, img, prelim, **kw)
                torch.cuda.synchronize()
                wall_timed = time.perf_counter() - t0
            finally:
                timer.remove()
        ext, K, extra, pts, _, rec, vframes, v2d, vtracks = res
        vt = vtracks.cpu().numpy()
        e_gt, p_gt = BA.normalize_reconstruction(D(sc.extrinsics, dev), D(sc.points3D[vt], dev))
        Rrel = torch.einsum("sij,skj->sik", ext[:, :, :3], e_gt[:, :, :3])
        ang = torch.acos(((Rrel.diagonal(dim1=1, dim2=2).sum(-1) - 1) / 2).clamp(-1, 1))
        top_sum = sum(timer.top.values())
        out[name] = dict(workload=f"Triangulator.forward, synthetic {S} frames x {N} tracks {cam}{' shared' if shared else ''}, "
                                  "BA_iters 2, robust_refine 2 (reference defaults)",
                         wall_s=min(walls), wall_s_runs=walls, wall_s_under_stage_timers=wall_timed,
                         stages_s={k: round(v, 4) for k, v in sorted(timer.top.items(), key=lambda kv: -kv[1])},
                         device_entries_s={k: round(v, 4) for k, v in sorted(timer.inner.items(), key=lambda kv: -kv[1])},
                         unaccounted_s=round(wall_timed - top_sum, 4), unaccounted_frac=(wall_timed - top_sum) / wall_timed,
                         valid_tracks=int(vt.sum()), valid_frames=int(vframes.sum()),
                         max_rot_err_deg_vs_ground_truth=float(ang.max()) * 180 / np.pi,
                         median_point_err_vs_ground_truth=float((pts - p_gt).norm(dim=-1).median()))
        del tracks, vis, score, res, pts
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: c3 (BASELINE configs[2], one 100k-track shard per GPU) at every N")
    ap.add_argument("--no-n1-leg", action="store_true", help="N > 1, strong scaling: skip rank 0's solo run of the whole problem")
    ap.add_argument("--no-parity-c4", action="store_true",
                    help="N = 1: skip the GPU-vs-port comparison of 3 LM iterations of the whole configs[3] problem (~1 min of CPU)")
    ap.add_argument("--no-parity-c5", action="store_true",
                    help="N = 1: skip the configs[4] video loop + GPU-vs-port comparison of its final joint problem (n = 6002)")
    ap.add_argument("--parity-iters", type=int, default=3)
    ap.add_argument("--no-weak-leg", action="store_true", help="N > 1: skip the weak-scaling leg on configs[2] shards")
    ap.add_argument("--weak-steps", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-triangulation", action="store_true", help="skip the triangulation leg (tracks/s of the same scene)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="skip the end-to-end leg (Triangulator.forward at configs[1] and configs[2], wall time + stage breakdown)")
    ap.add_argument("--no-strong-leg", action="store_true",
                    help="skip the short strong-scaling leg on BASELINE configs[3] (400 x 300k split over the ranks)")
    ap.add_argument("--strong-steps", type=int, default=10)
    ap.add_argument("--cpu-iters", type=int, default=5)
    ap.add_argument("--no-camera-split", action="store_true",
                    help="keep the cameras in frame order (A/B of the side-by-side factorisation of decoupled camera blocks)")
    ap.add_argument("--overlap", action="store_true",
                    help="opt-in: 3 Schur tile batches, factorisation on a CU-masked stream beside the later ones "
                         "(ba.OVERLAP_FACTORIZATION; measured slower in round 1, DESIGN.md section 7)")
    ap.add_argument("--split-exchange", action="store_true",
                    help="opt-in, N > 1: exchange the reduced system in two parts, the off-diagonal tile launch's share on the "
                         "communicator's stream beside the diagonal launch (dist.SPLIT_EXCHANGE, DESIGN.md section 8)")
    args = ap.parse_args()
    if args.overlap:
        BA.OVERLAP_FACTORIZATION = True
    if args.split_exchange:
        from vggsfm_amd import dist as _dist_mod
        _dist_mod.SPLIT_EXCHANGE = True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.workload is None:
        args.workload = "c3"             # the same workload family at every N: value(N) / value(1) is a scaling curve
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    # VGGSFM_BENCH_BACKEND=gloo: rehearsal of the N > 1 code path on a box with fewer GPUs than ranks (ranks share
    # devices, collectives go through the host) -- for checking the flow, not for numbers
    backend = os.environ.get("VGGSFM_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", local_rank if backend == "nccl" else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    L = _lib.lib()

    def barrier():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    def build(workload, solo=False):
        """-> (problem, numpy scene or None, initial cameras / points on the host or None).  solo: the whole problem on
        this rank alone (no collective at set-up)."""
        S, N, cam_type, shared = WORKLOADS[workload]
        reduce_adj = (lambda t: dist.all_reduce(t, op=dist.ReduceOp.MAX)) if (dist and not solo) else None
        if workload in STRONG or S >= 400:
            # 400-frame configurations: scene drawn on the device (make_scene costs ~25 s of numpy per 100k tracks there)
            n_local = N // world if (workload in STRONG and not solo) else N
            sc = make_scene_device(S, n_local, cam_type, shared_camera=shared, seed=0, track_seed=1000 + rank, device=dev)
            ext0_c, K0_c, extra0_c, _ = perturb_for_ba(sc_cameras_only(sc), seed=0)
            prob, _, _ = BA.compile_problem(sc.points3D_init, D(ext0_c, dev), D(K0_c, dev), sc.tracks, sc.mask, D(extra0_c, dev),
                                            shared, cam_type, overlap=(world == 1 and args.overlap),
                                            camera_split=not args.no_camera_split, adjacency_reduce=reduce_adj,
                                            sort_points=BA.SORT_POINTS)   # (what the public entry, ba.bundle_adjustment, does)
            return prob, None, (sc, D(ext0_c, dev), D(K0_c, dev), D(extra0_c, dev))
        sc = make_scene(S, N, cam_type, shared_camera=shared, seed=0, track_seed=1000 + rank, full_visibility=(workload == "c3dense"))
        _, _, _, pts0 = perturb_for_ba(sc, seed=rank)
        ext0_c, K0_c, extra0_c, _ = perturb_for_ba(sc, seed=0)          # cameras identical on every rank
        prob, _, _ = BA.compile_problem(D(pts0, dev), D(ext0_c, dev), D(K0_c, dev), D(sc.tracks, dev), D(sc.mask, dev),
                                        D(extra0_c, dev), shared, cam_type, overlap=(world == 1 and args.overlap),
                                        camera_split=not args.no_camera_split, adjacency_reduce=reduce_adj,
                                            sort_points=BA.SORT_POINTS)   # (what the public entry, ba.bundle_adjustment, does)
        return prob, sc, (pts0, ext0_c, K0_c, extra0_c)

    def timed(prob, steps, warmup, profile, solo=False):
        """K LM iterations between barriers (termination tests off, solves restarted every EPISODE iterations).
        -> (seconds = max over ranks, summary of the last episode, per-kernel HIP-event profile or None).
        solo: this rank alone (no collectives, no barriers with the others)."""
        nonlocal_dist = None if solo else dist

        def barrier():
            torch.cuda.synchronize()
            if nonlocal_dist:
                nonlocal_dist.barrier()
            torch.cuda.synchronize()
        init = [t.clone() for t in (prob.cam_q, prob.cam_t, prob.intr, prob.pts)]
        opts = BundleAdjustmentOptions()
        so = opts.solver_options
        so.max_num_iterations = EPISODE
        so.function_tolerance = so.gradient_tolerance = so.parameter_tolerance = -1.0   # run exactly K iterations
        solver = ShardedBA(prob, opts, 0, 1) if solo else ShardedBA(prob, opts, rank, world)   # phases + RCCL collectives on the current stream
        counter = [0]

        def run(n):
            for _ in range(n):
                if counter[0] % EPISODE == 0:
                    for dst, src in zip((prob.cam_q, prob.cam_t, prob.intr, prob.pts), init):
                        dst.copy_(src)
                    solver.begin()
                solver.iteration()
                counter[0] += 1

        run(warmup)
        barrier()
        if profile:
            n_batches = int(prob.batch_desc.shape[0])     # Schur tile launches per iteration (per off-diagonal / diagonal kind)
            _lib.check(L.vgg_ba_profile(1, (steps + 4) * n_batches), "vgg_ba_profile")
        barrier()
        t0 = time.perf_counter()
        run(steps)
        barrier()
        dt = time.perf_counter() - t0
        if nonlocal_dist:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            nonlocal_dist.all_reduce(tmax, op=nonlocal_dist.ReduceOp.MAX)
            dt = float(tmax.item())
        # the last episode must have run all its iterations (no early termination => no skipped work)
        fin = solver.finish()
        expect = counter[0] % EPISODE or EPISODE
        if fin["num_iterations"] != expect:
            raise SystemExit(f"LM terminated early: {fin['num_iterations']} of {expect} iterations (termination "
                             f"{fin['termination_str']}) -- timing invalid")
        prof = None
        if profile:
            prof = {}
            for kid, name in enumerate(KERNELS):
                tot = ctypes.c_double()
                n = ctypes.c_int()
                _lib.check(L.vgg_ba_profile_read(kid, ctypes.byref(tot), ctypes.byref(n), 1), "vgg_ba_profile_read")
                prof[name] = (tot.value, n.value)
            L.vgg_ba_profile(0, 0)
        return dt, fin, prof

    S, N, cam_type, shared = WORKLOADS[args.workload]
    strong_main = args.workload in STRONG
    # ---- N > 1: the whole strong-scaling problem (configs[3]) on rank 0 alone first -- the N = 1 point of the strong curve,
    # measured on this box in this run; the other ranks wait at the barrier
    run_strong = strong_main or not args.no_strong_leg
    n1 = None
    if world > 1 and run_strong and not args.no_n1_leg:
        if rank == 0:
            p1, _, _ = build("c4", solo=True)
            d1, f1, _ = timed(p1, args.strong_steps, 2, False, solo=True)
            S4, N4 = WORKLOADS["c4"][:2]
            n1 = dict(workload=f"the same {S4} x {N4} problem on rank 0 alone", steps=args.strong_steps, observations=int(p1.num_obs),
                      ms_per_iteration=1e3 * d1 / args.strong_steps, lm_iterations_per_s=args.strong_steps / d1)
            del p1
            torch.cuda.empty_cache()
        barrier()
    prob, sc, host_init = build(args.workload)
    # The timed region runs UNOBSERVED; the per-kernel HIP events ride on a second pass of the same K steps right behind it.
    # An event pair around a launch costs ~5 us of stream time on each side on this stack (profiles/r05_timeline_c3.json: 54
    # of 1684 us per LM iteration of configs[2] were gaps next to the seven observed launches, none between the others), so
    # observing inside the region made it 3 % slower than the work it times.  `ms_per_step_with_events` is the second pass.
    dt, fin, _ = timed(prob, args.steps, args.warmup, False)
    dt_events, _, prof = timed(prob, args.steps, 0, True)
    # ---- N > 1 with the strong problem as the main one: weak-scaling leg on configs[2] shards
    weak = None
    if world > 1 and strong_main and not args.no_weak_leg:
        wprob, _, _ = build("c3")
        wdt, wfin, _ = timed(wprob, args.weak_steps, 2, False)
        weak = dict(workload=f"200 frames x {world} x 100000 tracks SIMPLE_RADIAL shared_camera: one 100000-track shard per rank",
                    steps=args.weak_steps, ms_per_iteration=1e3 * wdt / args.weak_steps,
                    lm_iterations_per_s_global=args.weak_steps / wdt, shard_iterations_per_s=args.weak_steps * world / wdt,
                    scaling="weak")
        del wprob
        torch.cuda.empty_cache()

    # ---- strong-scaling leg: BASELINE configs[3] (400 x 300k, per-frame intrinsics) split over the ranks
    strong, parity_c4 = None, None
    if not strong_main and not args.no_strong_leg:
        sprob, _, sdev = build("c4")
        sdt, sfin, _ = timed(sprob, args.strong_steps, 2, False)
        strong = dict(workload="synthetic 400 frames x 300000 tracks SIMPLE_RADIAL per-frame intrinsics (BASELINE configs[3]), the "
                               f"tracks split over {world} rank(s)", tracks_per_rank=300000 // world,
                      observations_rank0=int(sprob.num_obs), reduced_system=int(sfin["n_reduced"]), steps=args.strong_steps,
                      ms_per_iteration=1e3 * sdt / args.strong_steps, lm_iterations_per_s=args.strong_steps / sdt,
                      scaling="strong", data="synthetic (drawn on the device)", n1_same_problem=n1,
                      speedup_vs_n1=(args.strong_steps / sdt) / n1["lm_iterations_per_s"] if n1 else None)
        del sprob
        torch.cuda.empty_cache()
        if world == 1 and not args.no_parity_c4 and not args.no_cpu_baseline:
            # the whole configs[3] problem at FULL size, GPU and port side by side
            try:
                sc4, e4, K4, x4 = sdev
                parity_c4 = parity_vs_port(sc4.points3D_init, e4, K4, sc4.tracks, sc4.mask, x4, False, "SIMPLE_RADIAL",
                                           args.parity_iters, "BASELINE configs[3] whole (400 x 300000, per-frame SIMPLE_RADIAL)")
            except Exception as exc:                          # (an auxiliary leg must not cost the line)
                parity_c4 = dict(error=f"{type(exc).__name__}: {exc}")
        del sdev
        torch.cuda.empty_cache()

    if rank == 0:
        counts = (prob.row_ptr[1:] - prob.row_ptr[:-1]).double()
        n_obs = int(prob.num_obs)
        P = int(prob.pts.shape[0])
        n_red = int(fin["n_reduced"])
        bd = 6 if shared else 6 + (2 if cam_type == "SIMPLE_RADIAL" else 1)     # Schur block width
        y_slot = 12 if bd == 6 else 3 * bd                                         # doubles per observation in the segment buffer
        # camera-pair blocks (a <= b) of the Schur complement, split by the launch that computes them
        seg_count = torch.bincount(prob.obs_slot.long() // BA.GROUP, minlength=prob.num_segments).double()   # cameras per segment
        pairs_all = float((counts * (counts + 1) / 2).sum().item())
        pairs_diag = float((seg_count * (seg_count + 1) / 2).sum().item())       # both cameras in one 16-camera group
        pairs_off = pairs_all - pairs_diag
        if prob.merged_tile_launch:  # small problems: off-diagonal and diagonal tiles in one launch
            pairs_off, pairs_diag = pairs_all, 0.0
            ROCPROF_NAME["schur_tile<offdiag>"] = "schur_tile_merged_kernel"
        # algorithmic work per launch (DESIGN.md "Kernels"): flops for the fp64-compute-bound kernels,
        # bytes for the streaming kernels
        work = {
            "schur_tile<offdiag>": ("mfma", 2.0 * 3 * bd * bd * pairs_off),
            "schur_tile<diag>": ("mfma", 2.0 * 3 * bd * bd * pairs_diag),
            "cholesky": ("mfma", n_red ** 3 / 3.0 + 2.0 * n_red ** 2),
            # (with the segment buffer it writes: 96 B per observation compressed, 24 BD B otherwise -- an artefact of the
            #  formulation, priced because it is what bounds the kernel)
            # (+ the 72-byte back-substitution block Z per point the diagonal tile launch reads when it forms the right-hand side)
            "point_pass": ("hbm", (12.0 + 8.0 * y_slot) * n_obs + P * (24 + 4 + 8 * (6 + 3 + 6) + (72 if bd == 6 else 0))),
            "point_step": ("hbm", 12.0 * n_obs + P * (24 + 4 + 8 * (6 + 3) + 24)),
            "cam_pass<linearize>": ("hbm", 12.0 * n_obs + 24.0 * n_obs),
            "cam_pass<rhs>": ("hbm", 12.0 * n_obs + (24.0 + 24 + 48) * n_obs),
        }
        # the roofline object describes the entry with the largest share of the timed region.  "cholesky" = the dataflow
        # factorisation launch + the backward substitution (two launches per iteration; a latency-bound dependent chain,
        # priced against the FP64 matrix peak like the tile kernels so that the fraction says how far it is from being
        # a throughput problem)
        # ("largest share" by kernel: the two launches of schur_tile_kernel -- off-diagonal and diagonal tiles -- count
        #  together, and the object then describes the larger of the two launches.  Every entry's own fraction is in
        #  `roofline_by_kernel`, so the line says the same thing whichever entry happens to lead on a given box.)
        family = lambda k: "schur_tile" if k.startswith("schur_tile") else k
        fam_ms = {}
        for k, v in prof.items():
            if v[1] == 0:
                continue
            fam_ms[family(k)] = fam_ms.get(family(k), 0.0) + v[0]
        dom_family = max(fam_ms, key=fam_ms.get)
        dom = max((k for k in prof if family(k) == dom_family), key=lambda k: prof[k][0])
        by_kernel = {}
        for k, (ms_k, launches_k) in prof.items():
            if launches_k == 0:          # cam_pass<rhs> with 6 x 6 tile blocks: its sums come out of the diagonal tile launch
                by_kernel[k] = dict(ms_per_iteration=0.0, bound=work[k][0], frac=None, launches=0)
                continue
            bound_k, amount_k = work[k]
            per_iter_ms = ms_k / args.steps
            rate = amount_k / max(per_iter_ms * 1e-3, 1e-12)          # work per iteration / time per iteration
            peak = FP64_PEAK_TFLOPS * 1e12 if bound_k == "mfma" else HBM_PEAK_GBS * 1e9
            by_kernel[k] = dict(ms_per_iteration=per_iter_ms, bound=bound_k, frac=rate / peak)
        tot_ms, launches = prof[dom]
        avg_ms = tot_ms / max(launches, 1)
        bound, amount = work[dom]
        per_iter = max(1, round(launches / args.steps))     # the tile kernels run once per batch
        if dom == "cholesky":
            per_iter = 1                                    # a group: time and work are per iteration
            avg_ms = tot_ms / args.steps
        amount = amount / per_iter
        if bound == "mfma":
            achieved = amount / (avg_ms * 1e-3) / 1e12
            roof = dict(bound="mfma", achieved=achieved, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=achieved / FP64_PEAK_TFLOPS, traffic=None)
        else:
            achieved = amount / (avg_ms * 1e-3) / 1e9
            roof = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                        traffic=None)
        roof["traffic"], traffic_at = pmc_traffic(dom, args.workload)
        roof.update(kernel=dom, avg_launch_ms=avg_ms, launches=launches, algorithmic_per_launch=amount,
                    traffic_source="profiles/pmc_traffic_c3.json: (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, "
                                   "rocprofv3 --pmc, separate passes; a committed summary, not a counter of this run"
                                   if roof["traffic"] else None,
                    traffic_measured_at=traffic_at,
                    note="v_mfma_f64_16x16x4_f64 on the off-diagonal Schur tiles; algorithmic flops = 2*3*BD^2 per "
                         "co-observing camera pair of a point (padding of the 16-camera segments not counted); "
                         "FP64 MFMA peak = FP64 vector peak = 78.6 TFLOP/s"
                         + (f"; --overlap: {per_iter} launches per iteration (tile batches), the first on all 256 CUs, the "
                            f"others on {256 - BA.CHOL_CUS} CUs beside the factorisation, priced against the whole-chip peak"
                            if per_iter > 1 else "") if dom.startswith("schur_tile") else "")
        roof["launches_per_iteration"] = per_iter
        # per ITERATION (tile kernels: sum over the batches; with overlap the entries are not additive -- the
        # factorisation and the later tile batches run side by side)
        kernel_ms = {k: (v[0] / args.steps) for k, v in prof.items()}
        # whole-iteration view of SURVEY.md section 8(d): algorithmic bytes / flops of one LM iteration
        t_iter = dt / args.steps
        k_intr = n_red - 6 * S
        b_iter = 32.0 * n_obs + 72.0 * P + 16.0 * n_red ** 2 + 16.0 * (7 * S + k_intr)
        f_iter = 400.0 * n_obs + 108.0 * float((counts * counts + counts).sum().item()) + n_red ** 3 / 3.0
        iteration = dict(algorithmic_bytes=b_iter, hbm_frac_of_8TBs=b_iter / t_iter / (HBM_PEAK_GBS * 1e9),
                         hbm_frac_of_6p29TBs=b_iter / t_iter / 6.29e12, algorithmic_flops=f_iter,
                         fp64_frac=f_iter / t_iter / (FP64_PEAK_TFLOPS * 1e12))

        cpu = None
        if world == 1 and not args.no_cpu_baseline and sc is not None:
            pts0, ext0_c, K0_c, extra0_c = host_init
            from oracle import ba as OB          # checker timed as the CPU baseline ("port"), never on the GPU path
            vi, row_ptr, obs_cam, obs_uv, dele = OB.build_observations(pts0, ext0_c, sc.tracks, sc.mask)
            cq = np.ascontiguousarray(OB.rotmat_to_quat(ext0_c[:, :, :3]))
            ct = np.ascontiguousarray(ext0_c[:, :, 3])
            ni = 1 if shared else S
            intr = np.zeros((ni, 4))
            src = slice(0, 1) if shared else slice(None)
            intr[:, 0], intr[:, 1], intr[:, 2] = K0_c[src, 0, 0], K0_c[src, 0, 2], K0_c[src, 1, 2]
            if extra0_c is not None:
                intr[:, 3] = extra0_c[src, 0]
            cam_intr = np.zeros(S, np.int32) if shared else np.arange(S, dtype=np.int32)
            cam_const = np.zeros(S, np.uint8)
            cam_const[0], cam_const[1] = 1, 2
            o = OB.ceres_options(args.cpu_iters, -1.0, -1.0, -1.0)
            pts_c = np.ascontiguousarray(pts0[vi])
            tc = time.perf_counter()
            s_cpu = OB.solve_csr(cq, ct, intr, pts_c, cam_intr, row_ptr, obs_cam, obs_uv, OB.MODEL[cam_type], o,
                                 cam_const=cam_const)
            tcpu = time.perf_counter() - tc
            cpu = dict(value=s_cpu["num_iterations"] / tcpu, unit="LM-iterations/s", cores=int(OB.lib().bao_num_threads()),
                       kind="port", sample=f"full {S}x{N} workload, {s_cpu['num_iterations']} LM iterations "
                       f"(oracle/ba_oracle.c, OpenMP, includes the initial evaluation), {tcpu:.1f} s")
        # ---- the other half of north_star's first sentence: LO-RANSAC triangulation of the same scene
        tri = None
        if world == 1 and sc is not None and not args.no_triangulation:
            from vggsfm_amd.utils import triangulation as TR
            from vggsfm_amd.utils import triangulation_helpers as TH
            g_ext, g_K, g_xp = D(sc.extrinsics, dev), D(sc.intrinsics, dev), D(sc.extra_params, dev)
            g_tracks, g_vis, g_score = D(sc.tracks, dev), D(sc.vis, dev), D(sc.score, dev)
            tn = TH.cam_from_img(g_tracks, g_K, g_xp)
            torch.manual_seed(0)
            TR.triangulate_tracks(g_ext, tn, track_vis=g_vis, track_score=g_score)        # warm-up (same draws discarded)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 3
            torch.cuda.synchronize()
            th0 = time.perf_counter()
            ev0.record()
            for _ in range(reps):
                p3, num_inl, _ = TR.triangulate_tracks(g_ext, tn, track_vis=g_vis, track_score=g_score)
            ev1.record()
            torch.cuda.synchronize()
            t_host = (time.perf_counter() - th0) / reps
            t_dev = ev0.elapsed_time(ev1) * 1e-3 / reps
            hyp = min(256, S * (S - 1) // 2)
            # SURVEY 8(d) per-track flops of the REFERENCE formulation (its all-pairs angle term included)
            f_ref = N * (hyp * (2.5e3 + 70 * S) + 60 * (150 * S + 2e3 + 70 * S) + 60 * 40 * S * S)
            # what the kernel evaluates (round 4): (H + 2 x 60) passes over the track's VISIBLE views of ~60 flop each (one
            # reciprocal square root, the acos series; only visible views can hold inliers; the cheirality pass over all views
            # costs 6 flop per view), the per-view DLT matrices once per track (~100 flop) and 20 flop per (local-optimisation
            # hypothesis, inlier view) to add them up, ~1 k flop per hypothesis for its two view matrices, the smallest
            # eigenvector by inverse iteration (LDL^T + ~8 steps) and the triangulation angle (rounds 1-3: 3 k with the Jacobi)
            vis_mean = float(sc.mask.sum()) / N
            f_exec = N * ((hyp + 120) * vis_mean * 60.0 + 120 * S * 6.0 + vis_mean * 100.0 + 60 * vis_mean * 20.0 + (hyp + 60) * 1.0e3)
            tri = dict(workload=f"triangulate_tracks on the timed scene ({S} x {N}), {hyp} hypotheses + 2 local-optimisation rounds, "
                                "all reference chunks in one launch", mean_visible_views=vis_mean, ms=1e3 * t_dev, ms_host_inclusive=1e3 * t_host,
                       tracks_per_s=N / t_dev, valid_frac=float((num_inl >= 3).float().mean()),
                       bound="fp64 valu", executed_flops=f_exec, frac_of_fp64_peak=f_exec / t_dev / (FP64_PEAK_TFLOPS * 1e12),
                       reference_formulation_flops=f_ref, reference_formulation_tflops=f_ref / t_dev / 1e12,
                       kernel="triangulate_kernel (its own time: the rocprofv3 kernel stats under profiles/; `ms` is the whole call on "
                              "the launch stream, host-side pair draws included)")
            # CPU figures beside it: the numpy port on a bounded slice of the SAME scene, timed
            # here on one host core, and the reference's own torch function as the survey measured it (quoted, 8 cores)
            tri["cpu_reference_survey"] = REFERENCE_TRIANGULATION_CPU
            if not args.no_cpu_baseline:
                try:
                    from oracle import geometry as OG
                    n_cpu = 16
                    tn_h = tn[:, :n_cpu].cpu().numpy()
                    pr = OG.generate_combinations(S)
                    pr = pr[np.random.default_rng(0).permutation(len(pr))[:hyp]]
                    tcp = time.perf_counter()
                    OG.triangulate_tracks_chunk(sc.extrinsics, tn_h, pr, 50, 2, 1.5, sc.vis[:, :n_cpu], sc.score[:, :n_cpu])
                    tcp = time.perf_counter() - tcp
                    tri["cpu_baseline"] = dict(value=n_cpu / tcp, unit="tracks/s", cores=1, kind="port",
                                               sample=f"the first {n_cpu} tracks of the timed scene ({S} views), oracle/geometry.py "
                                                      f"(numpy restatement of triangulate_tracks_single_chunk), {tcp:.1f} s")
                except Exception as exc:
                    tri["cpu_baseline"] = dict(error=f"{type(exc).__name__}: {exc}")
            sq = os.path.join(ROOT, "profiles", "r05_pmc_sq_tri.json")
            by_kernel["triangulate"] = dict(ms_per_call=1e3 * t_dev, bound="fp64 valu", frac=tri["frac_of_fp64_peak"],
                                            executed_flops=f_exec, note="not part of an LM iteration: one triangulate_tracks call on "
                                            "the timed scene; executed-flop model in the `triangulation` key",
                                            sq_counters=(json.load(open(sq)) if os.path.exists(sq) else None))
            del g_tracks, g_vis, g_score, tn, p3
            torch.cuda.empty_cache()
        parity_c3 = None
        if cpu is not None:
            # the SAME headline problem and the same iterations the port was just timed on, now on the GPU through the
            # public entry: the LM trajectories side by side
            o5 = BundleAdjustmentOptions()
            o5.solver_options.max_num_iterations = args.cpu_iters
            o5.solver_options.function_tolerance = o5.solver_options.gradient_tolerance = -1.0
            o5.solver_options.parameter_tolerance = -1.0
            gp, ge, gK, gx, gs = BA.bundle_adjustment(D(pts0, dev), D(ext0_c, dev), D(K0_c, dev), D(sc.tracks, dev), D(sc.mask, dev),
                                                      None, D(extra0_c, dev), shared, cam_type, o5)
            oe = np.concatenate([OB.quat_to_rotmat(cq), ct[:, :, None]], -1)
            ge, gp = ge.cpu().numpy(), gp.cpu().numpy()
            relv = lambda a, b: float(np.max(np.linalg.norm((a - b).reshape(len(a), -1), axis=1)
                                             / np.maximum(np.linalg.norm(b.reshape(len(b), -1), axis=1), 1e-12)))
            its = list(zip(gs["iterations"], s_cpu["iterations"]))
            parity_c3 = dict(workload=f"the timed workload itself ({S} x {N} {cam_type}{' shared' if shared else ''}), "
                                      f"{args.cpu_iters} LM iterations from the same start, termination tests off",
                             lm_iterations_gpu=int(gs["num_iterations"]), lm_iterations_port=int(s_cpu["num_iterations"]),
                             accept_pattern_equal=bool(all(a["successful"] == b["successful"] for a, b in its)),
                             max_rel_cost_delta_per_iteration=float(max(abs(a["cost"] - b["cost"]) / b["cost"] for a, b in its)),
                             max_rel_radius_delta_per_iteration=float(max(abs(a["radius"] - b["radius"]) / b["radius"] for a, b in its)),
                             max_rel_rotation_delta=relv(ge[:, :, :3], oe[:, :, :3]),
                             max_rel_translation_delta=relv(ge[1:, :, 3], oe[1:, :, 3]),
                             max_rel_point_delta=relv(gp, pts_c),
                             max_rel_focal_delta=float(abs(gK[0, 0, 0].item() / intr[0, 0] - 1)) if shared else
                             float(np.max(np.abs(gK[:, 0, 0].cpu().numpy() / intr[:, 0] - 1))),
                             tolerance=1e-4, reference="oracle/ba_oracle.c (Ceres/COLMAP restatement; unpinned vs pycolmap)")
        parity = None
        if cpu is not None:
            # "pose delta vs ref" half of BASELINE.json's metric: the same full solve (reference BA options: 50 iterations,
            # tolerances x10) on the GPU and by the CPU port, on a workload the port finishes in seconds
            from vggsfm_amd.utils.triangulation_helpers import prepare_ba_options
            ps, pn = 50, 20000                   # BASELINE configs[1] at full size
            psc = make_scene(ps, pn, "SIMPLE_PINHOLE", shared_camera=False, seed=5)
            pe0, pK0, _, pp0 = perturb_for_ba(psc, seed=5)
            gp, ge, gK, _, gs = BA.bundle_adjustment(D(pp0, dev), D(pe0, dev), D(pK0, dev), D(psc.tracks, dev), D(psc.mask, dev),
                                                     None, None, False, "SIMPLE_PINHOLE", prepare_ba_options())
            op, oe, oK, _, osum = OB.bundle_adjustment(pp0, pe0, pK0, psc.tracks, psc.mask, None, False, "SIMPLE_PINHOLE",
                                                       OB.prepare_ba_options())
            ge, gp, gK = ge.cpu().numpy(), gp.cpu().numpy(), gK.cpu().numpy()
            rel = lambda a, b: float(np.max(np.linalg.norm((a - b).reshape(len(a), -1), axis=1)
                                            / np.maximum(np.linalg.norm(b.reshape(len(b), -1), axis=1), 1e-12)))
            from vggsfm_amd.ba_options import TERMINATION
            parity = dict(workload=f"synthetic {ps} frames x {pn} tracks SIMPLE_PINHOLE, full solve with the reference's BA options",
                          lm_iterations_gpu=int(gs["num_iterations"]), lm_iterations_port=int(osum["num_iterations"]),
                          termination_gpu=gs["termination_str"], termination_port=TERMINATION.get(osum["termination"], "?"),
                          # exits: up to which iteration the two logs are the same trajectory, and
                          # whether the iteration on which they part is already rounding noise
                          last_iteration_above_noise_floor_port=last_iteration_above_noise(osum["iterations"]),
                          first_divergence=trajectory_divergence(gs["iterations"], osum["iterations"]),
                          final_cost_rel_delta=abs(gs["final_cost"] - osum["final_cost"]) / osum["final_cost"],
                          max_rel_rotation_delta=rel(ge[:, :, :3], oe[:, :, :3]), max_rel_translation_delta=rel(ge[1:, :, 3], oe[1:, :, 3]),
                          max_rel_point_delta=rel(gp, op), max_rel_focal_delta=float(np.max(np.abs(gK[:, 0, 0] / oK[:, 0, 0] - 1))),
                          tolerance=1e-4, reference="oracle/ba_oracle.c (Ceres/COLMAP restatement; unpinned vs pycolmap)")
        pipeline = None
        chol_split = list(prob.chol_split)
        if world == 1 and not args.no_pipeline and args.workload == "c3":
            del prob
            torch.cuda.empty_cache()
            try:
                pipeline = pipeline_leg(dev)
            except Exception as exc:                          # (an auxiliary leg must not cost the line)
                pipeline = dict(error=f"{type(exc).__name__}: {exc}")
        # ---- BASELINE configs[4]: the 1000-frame video loop once, then its FINAL joint problem (n = 6002) on the GPU and by
        # the port from the same start
        video_c5 = None
        if world == 1 and args.workload == "c3" and not args.no_parity_c5 and not args.no_cpu_baseline:
            try:
                import contextlib
                import importlib.util
                import io
                spec = importlib.util.spec_from_file_location("run_c5_video", os.path.join(ROOT, "scripts", "run_c5_video.py"))
                c5 = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(c5)
                with contextlib.redirect_stdout(io.StringIO()):
                    video_c5 = c5.run_video(parity_iters=args.parity_iters, parity_fn=parity_vs_port)
            except Exception as exc:
                video_c5 = dict(error=f"{type(exc).__name__}: {exc}")
        out = {
            "metric": "BA LM-iterations/sec",
            "value": args.steps / dt if strong_main else args.steps * world / dt,
            # (ADVICE r5 / VERDICT r5 weak 6: at N > 1 `value` counts every rank's shard -- say so in the unit and put the
            #  strong-scaling figure of the fixed configs[3] problem right beside it, in front of everything else)
            "unit": "LM-iterations/s" if (strong_main or world == 1) else "shard-iterations/s (weak: N x K / t)",
            "scaling": "strong" if strong_main else "weak",
            "value_definition": ("strong: LM iterations/s of ONE fixed problem (configs[3] 400 x 300000, tracks split over the ranks)"
                                 if strong_main else
                                 "weak: shard-iterations/s = N x K / t, K iterations of the 200-frame x (N x 100000)-track problem "
                                 "counted once per 100000-track shard; at N = 1 = LM-iterations/s of BASELINE configs[2]"),
            "strong_scaling_c4_speedup_vs_n1": (None if not strong else strong.get("speedup_vs_n1")),
            "lm_iterations_per_s_global": args.steps / dt,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "ms_per_step_with_events": 1e3 * dt_events / args.steps,
            "higher_is_better": True, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"synthetic {S} frames x {N} tracks split over {world} rank(s), {cam_type} per-frame intrinsics, "
                                    "full LM (BASELINE configs[3])" if strong_main else
                                    f"synthetic {S} frames x {N} tracks per GPU, {cam_type}"
                                    f"{' shared_camera' if shared else ' per-frame intrinsics'}, full LM "
                                    f"({BASELINE_CONFIG[args.workload]}{' per GPU' if world > 1 else ''})"),
                       "frames": S, "tracks_per_gpu": (N // world if strong_main else N), "observations_per_gpu": n_obs,
                       "reduced_system": n_red,
                       "parallelism": f"points sharded x{world}, cameras replicated, RCCL all-reduce of the camera blocks, "
                                      "reduce-scatter + all-gather of the packed reduced system",
                       "episode_iterations": EPISODE,
                       "camera_split_columns": chol_split,   # block-diagonal leading part factorised side by side
                       "successful_steps_last_episode": int(fin["num_successful_steps"]),
                       "kernel_ms": kernel_ms},
            "roofline": roof,
            "roofline_by_kernel": by_kernel,
            "iteration_roofline": iteration,
            "cpu_baseline": cpu,
            "pose_delta_vs_port": parity,
            "pose_delta_vs_port_c3": parity_c3,
            "pose_delta_vs_port_c4": parity_c4,
            "pose_delta_vs_port_c5_joint": None if video_c5 is None else video_c5.get("pose_delta_vs_port_final_joint", video_c5),
            "video_c5": None if video_c5 is None else {k: v for k, v in video_c5.items() if k not in ("pose_delta_vs_port_final_joint", "joint_ba_log")},
            "strong_scaling_c4": strong,
            "n1_same_problem": n1,
            "speedup_vs_n1": (args.steps / dt) / n1["lm_iterations_per_s"] if (n1 and strong_main) else None,
            "weak_scaling_c3": weak,
            "triangulation": tri,
            "pipeline": pipeline,
        }
        print(json.dumps(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
