"""Generate tests/golden/*.npz by running the REFERENCE's own torch functions on the CPU.

Run in the build container only (needs /root/reference):
    python -m oracle.gen_golden
The vectors are committed; the GPU box and the test-suite only read the .npz files.
Tie-break normalisation: ``torch.sort`` is forced to ``stable=True`` while the reference
runs (SURVEY §7 hard part 2); RNG: ``torch.manual_seed(seed)`` right before each
``triangulate_tracks`` call, and the hypothesis pairs the reference drew are recorded.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_harness  # noqa: E402
from vggsfm_amd.scene import make_scene, perturb_for_ba  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


class _StableSort:
    def __enter__(self):
        self._orig = torch.sort

        def stable_sort(x, *a, **kw):
            kw["stable"] = True
            return self._orig(x, *a, **kw)

        torch.sort = stable_sort
        return self

    def __exit__(self, *exc):
        torch.sort = self._orig


class _RecordRandperm:
    """Record every torch.randperm draw the reference makes (one per chunk)."""

    def __enter__(self):
        self._orig = torch.randperm
        self.draws = []

        def rp(n, *a, **kw):
            out = self._orig(n, *a, **kw)
            self.draws.append(out.clone().numpy())
            return out

        torch.randperm = rp
        return self

    def __exit__(self, *exc):
        torch.randperm = self._orig


def T(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x))


def tc(x):
    """layout the reference's .view at triangulation.py:817 needs (SURVEY §7 hard part 3)."""
    return x.transpose(0, 1).contiguous().transpose(0, 1)


def main():
    tri, helpers, dist = ref_harness.load()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    only = sys.argv[1:]                 # e.g. `python -m oracle.gen_golden s200_chunked s400_chunked`: just those LO-RANSAC cases
    if not only:
        geometry_goldens(tri, helpers)
    if not only or any(o.startswith("geom_") for o in only):
        distortion_model_goldens(tri, helpers, only)
    if not only or not all(o.startswith("geom_") for o in only):
        triangulation_goldens(tri, helpers, only)


def geometry_goldens(tri, helpers):

    # ---------------- projection / normalisation / filter ----------------
    for name, cam_type, S, N in [("pinhole", "SIMPLE_PINHOLE", 8, 256), ("radial", "SIMPLE_RADIAL", 12, 200)]:
        sc = make_scene(S, N, cam_type, shared_camera=(cam_type == "SIMPLE_RADIAL"), seed=3)
        ext, K, extra, pts = perturb_for_ba(sc, seed=3)
        # put a few points behind cameras / far away to exercise the guards
        pts[0] = [0.0, 0.0, -1.0]
        pts[1] = [500.0, 0.0, 4.0]
        pts[2] = ext[0, :, :3].T @ (-ext[0, :, 3])          # at camera-0 centre: z = 0 -> inf/nan
        p2, pc = helpers.project_3D_points(T(pts), T(ext), T(K), T(extra), return_points_cam=True)
        tn = helpers.cam_from_img(T(sc.tracks), T(K), T(extra))
        out = dict(points3D=pts, extrinsics=ext, intrinsics=K, tracks=sc.tracks,
                   proj2D=p2.numpy(), proj_cam=pc.numpy(), tracks_normalized=tn.numpy())
        if extra is not None:
            out["extra_params"] = extra
        for chk in (False, True):
            for thr in (4, 1):
                m, d = helpers.filter_all_points3D(T(pts), T(sc.tracks), T(ext), T(K), T(extra),
                                                   max_reproj_error=thr, check_triangle=chk, return_detail=True)
                out[f"filter_mask_chk{int(chk)}_thr{thr}"] = m.numpy()
                out[f"filter_detail_chk{int(chk)}_thr{thr}"] = d.numpy().astype(bool)
        np.savez_compressed(os.path.join(OUT, f"geom_{name}.npz"), **out)
        print("wrote geom", name)

    # ---------------- triangulate_by_pair ----------------
    sc = make_scene(8, 300, "SIMPLE_PINHOLE", seed=5)
    ext, K, extra, _ = perturb_for_ba(sc, seed=5, rot_deg=0.2, trans=0.01)
    tn = helpers.cam_from_img(T(sc.tracks), T(K))
    p3, che, ang = tri.triangulate_by_pair(T(ext)[None], tn[None])
    np.savez_compressed(os.path.join(OUT, "tri_by_pair.npz"), extrinsics=ext, tracks_normalized=tn.numpy(),
                        points=p3.numpy(), cheirality=che.numpy(), angle=ang.numpy())
    print("wrote tri_by_pair")



def distortion_model_goldens(tri, helpers, only=()):
    """The 2- and 4-parameter distortion models of the reference (vggsfm/utils/distortion.py:102-159: RADIAL k1, k2 and
    OPENCV k1, k2, p1, p2) through its own project_3D_points / cam_from_img (iterative_undistortion) / filter_all_points3D
    (VERDICT r2 item 7).  Per-frame coefficients; the observations are the reference's own projections of the ground-truth
    scene + 0.5 px noise (float32, as the tracker delivers them), so that the filters have inliers to find."""
    for name, kparams, S, N, seed in [("radial2", 2, 10, 220, 31), ("opencv4", 4, 9, 240, 32)]:
        if only and f"geom_{name}" not in only:
            continue
        rng = np.random.default_rng(seed)
        sc = make_scene(S, N, "SIMPLE_PINHOLE", shared_camera=False, seed=seed)
        extra = np.zeros((S, kparams))
        extra[:, 0] = 0.05 + 0.02 * rng.standard_normal(S)             # k1
        extra[:, 1] = 0.01 * rng.standard_normal(S)                    # k2
        if kparams == 4:
            extra[:, 2:] = 2e-3 * rng.standard_normal((S, 2))          # p1, p2
        clean = helpers.project_3D_points(T(sc.points3D), T(sc.extrinsics), T(sc.intrinsics), T(extra)).numpy()
        tracks = (clean + 0.5 * rng.standard_normal(clean.shape)).astype(np.float32)
        ext, K, _, pts = perturb_for_ba(sc, seed=seed, rot_deg=0.05, trans=0.002, focal_rel=0.001, point=0.005)
        pts[0] = [0.0, 0.0, -1.0]
        pts[1] = [500.0, 0.0, 4.0]
        p2, pc = helpers.project_3D_points(T(pts), T(ext), T(K), T(extra), return_points_cam=True)
        tn = helpers.cam_from_img(T(tracks), T(K), T(extra))
        out = dict(points3D=pts, extrinsics=ext, intrinsics=K, tracks=tracks, extra_params=extra,
                   proj2D=p2.numpy(), proj_cam=pc.numpy(), tracks_normalized=tn.numpy())
        for chk in (False, True):
            for thr in (4, 1):
                m, d = helpers.filter_all_points3D(T(pts), T(tracks), T(ext), T(K), T(extra), max_reproj_error=thr,
                                                   check_triangle=chk, return_detail=True)
                out[f"filter_mask_chk{int(chk)}_thr{thr}"] = m.numpy()
                out[f"filter_detail_chk{int(chk)}_thr{thr}"] = d.numpy().astype(bool)
        np.savez_compressed(os.path.join(OUT, f"geom_{name}.npz"), **out)
        print("wrote geom", name, "filter thr4 keeps", int(out["filter_mask_chk0_thr4"].sum()), "of", N)


def triangulation_goldens(tri, helpers, only=()):
    # ---------------- triangulate_tracks (LO-RANSAC) ----------------
    cases = [
        # name, S, N, camera, max_ransac_iters, max_tri_points_num, seed
        ("s8_all_pairs", 8, 256, "SIMPLE_PINHOLE", 256, 819200, 11),      # C(8,2)=28 < 256: no randperm
        ("s30_randperm", 30, 160, "SIMPLE_PINHOLE", 256, 819200, 12),      # 435 pairs -> 256 sampled
        ("s30_it128", 30, 96, "SIMPLE_RADIAL", 128, 819200, 13),           # iterative_global_BA setting
        ("s24_chunked", 24, 200, "SIMPLE_PINHOLE", 256, 2000, 14),         # 3 chunks, one randperm each
        # the view counts of BASELINE configs[2] / [3] (the kernel's large-S regime: 2 wavefronts/SIMD, wave-uniform view
        # loops, several reference chunks in one launch); ~1-3 min of reference CPU time each
        ("s200_chunked", 200, 480, "SIMPLE_RADIAL", 256, 200 * 160, 15),   # 3 chunks of 160 tracks
        ("s400_chunked", 400, 320, "SIMPLE_RADIAL", 256, 400 * 160, 16),   # 2 chunks of 160 tracks
    ]
    for name, S, N, cam_type, iters, max_pts, seed in cases:
        if only and name not in only:
            continue
        sc = make_scene(S, N, cam_type, shared_camera=(cam_type == "SIMPLE_RADIAL"), seed=seed,
                        outlier_frac=0.10)
        ext, K, extra, _ = perturb_for_ba(sc, seed=seed, rot_deg=0.1, trans=0.005, focal_rel=0.002)
        vis = sc.vis.copy()
        score = sc.score.copy()
        score[:, ::17] = 0.4          # whole tracks with low score
        vis[0, ::13] = 0.03           # low visibility entries
        tn = helpers.cam_from_img(T(sc.tracks), T(K), T(extra))
        torch.manual_seed(seed)
        with _StableSort(), _RecordRandperm() as rec:
            p3, num, msk = tri.triangulate_tracks(T(ext), tc(tn), max_ransac_iters=iters, track_vis=tc(T(vis)),
                                                  track_score=tc(T(score)), max_tri_points_num=max_pts)
        out = dict(extrinsics=ext, tracks_normalized=tn.numpy(), vis=vis, score=score, points=p3.numpy(),
                   inlier_num=num.numpy(), inlier_mask=msk.numpy(), max_ransac_iters=iters,
                   max_tri_points_num=max_pts, seed=seed, n_draws=len(rec.draws))
        for i, d in enumerate(rec.draws):
            out[f"randperm_{i}"] = d
        np.savez_compressed(os.path.join(OUT, f"tri_tracks_{name}.npz"), **out)
        print("wrote tri_tracks", name, "valid>=3:", int((num >= 3).sum()), "/", N, "draws", len(rec.draws))


if __name__ == "__main__":
    main()
