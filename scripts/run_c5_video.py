#!/usr/bin/env python
"""BASELINE configs[4] once, end to end on one GPU: ``VideoGeometry.run`` (the geometric half of the reference's
``VideoRunner.run``, vggsfm/runners/video_runner.py:64-247, 640-905, 494-541) over a synthetic 1000-frame translating
trajectory with a simulated camera predictor (noisy ground truth in a fresh similarity gauge per call) and a simulated
tracker (ground-truth projections + pixel noise).  The learned parts are out of scope; what is measured is
register -> triangulate -> window BA -> table update, the joint BA over all frames so far every few windows, and the
size / time of the final joint BA (all 1000 frames, n = 6002 reduced unknowns).

    python scripts/run_c5_video.py [--frames 1000] [--points 500000] [--out profiles/r02_c5_video.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vggsfm_amd import ba as BA  # noqa: E402
from vggsfm_amd import video as V  # noqa: E402


def rodrigues(w):
    th = np.linalg.norm(w)
    if th < 1e-12:
        return np.eye(3)
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def run_video(frames=1000, points=500000, init=32, window=16, new_tracks=4096, max_query_pts=2048, joint_interval=6,
              parity_iters=0, parity_fn=None):
    """The loop once -> the record dict (`joint_ba_log` included).  parity_iters > 0: the inputs of the LAST joint BA of the
    loop (all frames, n = 6 frames + 2 unknowns) are kept, and after the loop the same problem is solved for `parity_iters` LM
    iterations from the same start on the GPU and by the CPU port (`parity_fn` = bench.parity_vs_port; the checker is
    only ever called from there) -> `pose_delta_vs_port_final_joint`."""
    import types
    args = types.SimpleNamespace(frames=frames, points=points, init=init, window=window, new_tracks=new_tracks,
                                 max_query_pts=max_query_pts, joint_interval=joint_interval)
    dev = "cuda"
    T, N, W, f, k1 = args.frames, args.points, 1024.0, 1000.0, 0.02
    rng = np.random.default_rng(3)
    ext = np.zeros((T, 3, 4))
    for t in range(T):
        R = rodrigues(np.array([0.02 * np.sin(0.3 * t), 0.06 * np.sin(0.15 * t), 0.01 * np.cos(0.2 * t)]))
        c = np.array([0.11 * t, 0.05 * np.sin(0.25 * t), 0.03 * np.cos(0.2 * t)])
        ext[t, :, :3], ext[t, :, 3] = R, -R @ c
    pts = np.stack([rng.uniform(-2.5, 0.11 * T + 2.5, N), rng.uniform(-1.5, 1.5, N), rng.uniform(3.0, 6.0, N)], 1)
    E = torch.from_numpy(ext).to(dev)
    P = torch.from_numpy(pts).to(dev)
    gen = torch.Generator(device=dev).manual_seed(4)
    order = torch.argsort(P[:, 0])                         # points sorted along the path: a frame sees a contiguous x-range
    P = P[order]

    def observe(f0, f1, idx):
        """ground-truth projections of points idx into frames f0..f1-1 (+ 0.3 px noise, fixed per (frame, point))"""
        Xc = torch.einsum("sij,nj->sni", E[f0:f1, :, :3], P[idx]) + E[f0:f1, None, :, 3]
        u, v = Xc[..., 0] / Xc[..., 2], Xc[..., 1] / Xc[..., 2]
        d = 1.0 + k1 * (u * u + v * v)
        uv = torch.stack([f * u * d + W / 2, f * v * d + W / 2], -1)
        vis = (uv[..., 0] > 8) & (uv[..., 0] < W - 8) & (uv[..., 1] > 8) & (uv[..., 1] < W - 8) & (Xc[..., 2] > 0.5)
        # deterministic pseudo-noise from (frame, point): the same observation is the same pixel in every call
        fr = torch.arange(f0, f1, device=dev)[:, None].double()
        h = torch.sin(fr * 12.9898 + idx[None].double() * 78.233) * 43758.5453
        n1 = (h - torch.floor(h)) - 0.5
        h2 = torch.sin(fr * 39.3468 + idx[None].double() * 11.135) * 24634.6345
        n2 = (h2 - torch.floor(h2)) - 0.5
        uv = uv + 1.04 * torch.stack([n1, n2], -1)          # uniform noise with sigma 0.3 px
        # a tracker has nothing to follow outside the image: those pixels are junk (they must not pass the reprojection
        # filters, which -- as in the reference -- do not look at the visibility)
        uv = torch.where(vis[..., None], uv, torch.full_like(uv, -4000.0))
        return uv.float(), vis

    def candidates(fr):
        c = -ext[fr, :, :3].T @ ext[fr, :, 3]
        lo = int(torch.searchsorted(P[:, 0].contiguous(), torch.tensor(c[0] - 5.0, device=dev, dtype=P.dtype)))
        hi = int(torch.searchsorted(P[:, 0].contiguous(), torch.tensor(c[0] + 5.0, device=dev, dtype=P.dtype)))
        return torch.arange(lo, hi, device=dev)

    prng = np.random.default_rng(9)

    def camera_prior(f0, f1):
        e = ext[f0:f1].copy()
        Rg, s, tg = rodrigues(prng.normal(0, 0.5, 3)), float(prng.uniform(0.5, 2.0)), prng.normal(0, 1.0, 3)
        out = np.zeros_like(e)
        for i in range(len(e)):
            Rn = rodrigues(prng.normal(0, 0.01, 3)) @ e[i, :, :3]
            tn = e[i, :, 3] + prng.normal(0, 0.02, 3)
            out[i, :, :3] = Rn @ Rg.T
            out[i, :, 3] = s * tn - out[i, :, :3] @ tg
        return torch.from_numpy(out).to(dev)

    def track_existing(f0, f1, uv):
        # the query pixel identifies the ground-truth point: exact match among the points visible in f0
        cand = candidates(f0)
        tr0, v0 = observe(f0, f0 + 1, cand)
        cand = cand[v0[0]]
        ref = tr0[0][v0[0]]
        # nearest ground-truth observation (the projection is recomputed with another batch shape: last-bit differences)
        idx = torch.empty(len(uv), dtype=torch.long, device=dev)
        for a in range(0, len(uv), 512):
            dmat = torch.cdist(uv[a:a + 512].double(), ref.double())
            dist, idx[a:a + 512] = dmat.min(dim=1)
            assert float(dist.max()) < 1e-2, "query pixel not found among the ground-truth observations"
        ids = cand[idx]
        tr, vis = observe(f0, f1, ids)
        return tr, vis.float()

    def track_new(f0, f1):
        cand = candidates(f0)
        _, v0 = observe(f0, f0 + 1, cand)
        cand = cand[v0[0]]
        sel = cand[torch.randperm(cand.numel(), device=dev, generator=gen)[:args.new_tracks]]
        tr, vis = observe(f0, f1, sel)
        return tr, vis.float(), torch.ones_like(vis, dtype=torch.float32)

    # initial window: a perturbed reconstruction of frames 0..init (what sparse_reconstruct hands over)
    INIT = args.init
    cand = candidates(INIT // 2)
    tr, vis = observe(0, INIT, cand)
    keep = vis.sum(0) >= 4
    idx0 = cand[keep][:4 * args.new_tracks]
    tr, vis = observe(0, INIT, idx0)
    pred = {"extrinsics_opencv": E[:INIT], "pred_track": tr, "pred_vis": vis.float(), "valid_2D_mask": vis,
            "valid_tracks": torch.ones(len(idx0), dtype=torch.bool, device=dev),
            "points3D": P[idx0] + 0.01 * torch.randn(len(idx0), 3, device=dev, dtype=P.dtype, generator=gen), "points3D_rgb": None}
    K = torch.tensor([[f, 0, W / 2], [0, f, W / 2], [0, 0, 1.0]], device=dev, dtype=torch.float64)[None]
    vg = V.VideoGeometry(K, torch.tensor([[k1]], device=dev, dtype=torch.float64), "SIMPLE_RADIAL",
                         max_query_pts=args.max_query_pts, device=dev, generator=gen)
    vg.add_initial_window(pred, 0, INIT)

    # instrument the two BA entry points
    log = {"window": [], "joint": []}
    wba, jba = V.window_bundle_adjustment, V.joint_bundle_adjustment

    last_joint = {}

    def timed(kind, fn):
        def wrapper(*a, **k):
            if kind == "joint" and parity_iters > 0:
                last_joint["args"] = a                      # (references only: the loop builds fresh tensors per call)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = fn(*a, **k)
            torch.cuda.synchronize()
            summ = out[4] if kind == "window" else out[6]
            log[kind].append(dict(ms=1e3 * (time.perf_counter() - t0), iterations=summ["num_iterations"], n_reduced=summ["n_reduced"],
                                  points=int(a[0].shape[0]), frames=int(a[1].shape[0]),
                                  observations=int(a[4].sum()) if kind == "joint" else int(a[4].sum())))
            return out
        return wrapper
    V.window_bundle_adjustment = timed("window", wba)
    V.joint_bundle_adjustment = timed("joint", jba)

    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        table = vg.run(T, INIT, args.window, camera_prior, track_existing, track_new, joint_BA_interval=args.joint_interval)
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
    finally:
        V.window_bundle_adjustment, V.joint_bundle_adjustment = wba, jba

    parity = None
    if parity_iters > 0 and "args" in last_joint:
        # video.joint_bundle_adjustment(points3d, extrinsics, intrinsics (1,3,3), tracks, masks, extra_params (1,1), camera_type, ...)
        # normalises, then hands (pts, ext, K per frame, tracks, masks, extra per frame, shared = True) to the solver: the same here
        jp, je, jK, jtr, jm, jx, jcam = last_joint["args"][:7]
        je, jp = BA.normalize_reconstruction(je.to(torch.float64), jp.to(torch.float64))
        Sj = je.shape[0]
        if parity_fn is None:
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            import bench
            parity_fn = bench.parity_vs_port
        parity = parity_fn(jp, je, jK.expand(Sj, -1, -1).contiguous(), jtr, jm, jx.expand(Sj, -1).contiguous(), True, jcam, parity_iters,
                           f"the final joint BA of the {T}-frame video loop (BASELINE configs[4]: {Sj} frames, shared {jcam})")
        last_joint.clear()

    # kernel breakdown of one LM iteration of the final joint problem (HIP events, as in bench.py)
    import ctypes
    from vggsfm_amd import _lib
    from vggsfm_amd.ba_options import BundleAdjustmentOptions
    from vggsfm_amd.dist import ShardedBA
    xyz, e_all, trk, msk, _ = table.window_tensors(0, T)
    prob, _, _ = BA.compile_problem(xyz, e_all, vg.intrinsics.double().expand(T, -1, -1), trk, msk,
                                    vg.extra_params.double().expand(T, -1), True, "SIMPLE_RADIAL", camera_split=True)
    o = BundleAdjustmentOptions()
    o.solver_options.max_num_iterations = 12
    o.solver_options.function_tolerance = o.solver_options.gradient_tolerance = o.solver_options.parameter_tolerance = -1.0
    sb = ShardedBA(prob, o)
    sb.begin()
    for _ in range(2):
        sb.iteration()
    L = _lib.lib()
    L.vgg_ba_profile(1, 64)
    torch.cuda.synchronize()
    tk = time.perf_counter()
    for _ in range(8):
        sb.iteration()
    torch.cuda.synchronize()
    it_ms = 1e3 * (time.perf_counter() - tk) / 8
    names = ["cam_pass<linearize>", "point_pass", "cam_pass<rhs>", "schur_tile<offdiag>", "cholesky", "point_step", "schur_tile<diag>"]
    kernel_ms = {}
    for kid, nm in enumerate(names):
        tot, cnt = ctypes.c_double(), ctypes.c_int()
        L.vgg_ba_profile_read(kid, ctypes.byref(tot), ctypes.byref(cnt), 1)
        kernel_ms[nm] = tot.value / 8
    L.vgg_ba_profile(0, 0)
    order_info = dict(k_way_envelope=prob.chol_first_blk is not None, chol_split=list(prob.chol_split),
                      envelope_blocks=None if prob.chol_first_blk is None else prob.chol_first_blk.cpu().tolist())

    est = table.extri[:T]
    aR, aT, a_s = V.align_camera_extrinsics(est, E)
    al = V.apply_transformation(est, aR, aT, a_s)
    Rrel = torch.bmm(al[:, :, :3], E[:, :, :3].transpose(1, 2))
    ang = torch.acos(((Rrel.diagonal(dim1=1, dim2=2).sum(1) - 1) / 2).clamp(-1, 1))
    c_est = -torch.bmm(al[:, :, :3].transpose(1, 2), al[:, :, 3:])[..., 0]
    c_gt = -torch.bmm(E[:, :, :3].transpose(1, 2), E[:, :, 3:])[..., 0]
    path = float((c_gt[-1] - c_gt[0]).norm())
    last = log["joint"][-1]
    out = dict(workload=f"synthetic video, {T} frames, {N} scene points, init window {INIT}, window {args.window}, "
                        f"{args.new_tracks} new tracks + <= {args.max_query_pts} carried-over points per window, joint BA every "
                        f"{args.joint_interval} windows (BASELINE configs[4] path on ONE GPU; predictor / tracker simulated)",
               total_seconds=total, windows=len(log["window"]),
               window_ba_ms_mean=float(np.mean([w["ms"] for w in log["window"]])),
               window_ba_iterations_mean=float(np.mean([w["iterations"] for w in log["window"]])),
               joint_ba_calls=len(log["joint"]), joint_ba_seconds_total=sum(j["ms"] for j in log["joint"]) / 1e3,
               final_joint_ba=dict(frames=last["frames"], points=last["points"], observations=last["observations"],
                                   n_reduced=last["n_reduced"], iterations=last["iterations"], ms=last["ms"],
                                   ms_per_iteration=last["ms"] / max(last["iterations"], 1)),
               final_joint_problem_iteration_ms=it_ms, final_joint_problem_kernel_ms=kernel_ms, camera_order=order_info,
               table_points=int(table.num_points), table_observations=int(table.num_observations),
               max_rotation_error_rad=float(ang.max()), max_centre_error=float((c_est - c_gt).norm(dim=1).max()), path_length=path,
               focal=float(vg.intrinsics[0, 0, 0]), k1=float(vg.extra_params[0, 0]), pose_delta_vs_port_final_joint=parity,
               joint_ba_log=log["joint"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--points", type=int, default=500000)
    ap.add_argument("--init", type=int, default=32)
    ap.add_argument("--window", type=int, default=16)
    ap.add_argument("--new-tracks", type=int, default=4096)
    ap.add_argument("--max-query-pts", type=int, default=2048)
    ap.add_argument("--joint-interval", type=int, default=6)
    ap.add_argument("--parity-iters", type=int, default=0,
                    help="> 0: the final joint problem also on the CPU port for this many LM iterations (bench.parity_vs_port)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    out = run_video(a.frames, a.points, a.init, a.window, a.new_tracks, a.max_query_pts, a.joint_interval, a.parity_iters)
    print(json.dumps({k: v for k, v in out.items() if k != "joint_ba_log"}))
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
