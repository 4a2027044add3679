"""End-to-end geometry stage at BASELINE configs[1] / configs[2]: vggsfm_amd.models.Triangulator.forward
(triangulation + pose refinement + iterative bundle adjustment) on synthetic tracks, wall clock and accuracy
against the ground-truth scene (after the same normalisation).  Not the bench.py metric; one JSON line per config."""
import json
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vggsfm_amd import ba as BA  # noqa: E402
from vggsfm_amd.models import Triangulator  # noqa: E402
from vggsfm_amd.scene import make_scene, perturb_for_ba  # noqa: E402

TIMES = {}


def _timed(mod, name):
    """Wrap mod.name with a synchronising timer (profiling aid: the syncs serialise the host with the device)."""
    fn = getattr(mod, name)

    def wrapper(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn(*a, **k)
        torch.cuda.synchronize()
        TIMES[name] = TIMES.get(name, 0.0) + time.perf_counter() - t0
        return out

    setattr(mod, name, wrapper)


def install_timers():
    from vggsfm_amd import pose as PO
    from vggsfm_amd.models import triangulator as TM
    from vggsfm_amd.utils import triangulation as TU
    _timed(BA, "compile_problem")
    _timed(BA, "solve")
    for mod in (TU, TM):
        for name in ("triangulate_tracks", "filter_all_points3D", "cam_from_img", "pose_refinement_batch",
                     "triangulate_by_pair"):
            if hasattr(mod, name):
                _timed(mod, name)
    del PO


CONFIGS = {"c2": (50, 20000, "SIMPLE_PINHOLE", False), "c3": (200, 100000, "SIMPLE_RADIAL", True)}


def D(x):
    return None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()


for name, (S, N, cam, shared) in CONFIGS.items():
    if any(a in CONFIGS for a in sys.argv[1:]) and name not in sys.argv[1:]:
        continue
    W = 1024
    sc = make_scene(S, N, cam, shared_camera=shared, seed=0)
    ext0, K0, _, _ = perturb_for_ba(sc, seed=0, rot_deg=0.5, trans=0.02, focal_rel=0.02)
    cams = types.SimpleNamespace(R=D(ext0[:, :, :3]).float(), T=D(ext0[:, :, 3]).float(),
                                 focal_length=torch.stack([D(K0[:, 0, 0] / (W / 2.0)).float()] * 2, -1))
    images = torch.zeros(1, S, 3, 8, 8, device="cuda")                  # colours are not extracted here
    images = images.expand(1, S, 3, 8, 8)

    class _Img:                                                        # only .shape is read when extract_color=False
        shape = (1, S, 3, W, W)

    prelim = {"fmat_inlier_mask": D(sc.mask[1:] & sc.mask[0:1])[None]}
    tri = Triangulator()
    if not TIMES and "--breakdown" in sys.argv:
        install_timers()
    # warm-up on the first 2000 tracks (library load, allocator, first-launch costs)
    tri(cams, D(sc.tracks[:, :2000])[None], D(sc.vis[:, :2000])[None], _Img,
        {"fmat_inlier_mask": D(sc.mask[1:, :2000] & sc.mask[0:1, :2000])[None]}, pred_score=D(sc.score[:, :2000])[None],
        shared_camera=shared, camera_type=cam, BA_iters=1, robust_refine=1, extract_color=False)
    TIMES.clear()
    torch.manual_seed(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = tri(cams, D(sc.tracks)[None], D(sc.vis)[None], _Img, prelim, pred_score=D(sc.score)[None], shared_camera=shared,
              camera_type=cam, BA_iters=2, robust_refine=2, extract_color=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ext, K, extra, pts, rgb, rec, vframes, v2d, vtracks = out
    # accuracy: normalise the ground truth the same way and compare
    vt = vtracks.cpu().numpy()
    e_gt, p_gt = BA.normalize_reconstruction(D(sc.extrinsics), D(sc.points3D[vt]))
    Rrel = torch.einsum("sij,skj->sik", ext[:, :, :3], e_gt[:, :, :3])
    ang = torch.acos(((Rrel.diagonal(dim1=1, dim2=2).sum(-1) - 1) / 2).clamp(-1, 1))
    perr = (pts - p_gt).norm(dim=-1)
    print(json.dumps({"config": name, "frames": S, "tracks": N, "wall_s": dt, "valid_tracks": int(vt.sum()),
                      "valid_frames": int(vframes.sum()), "max_rot_err_deg": float(ang.max()) * 180 / np.pi,
                      "median_point_err": float(perr.median()), "focal_median": float(K[:, 0, 0].median()),
                      "breakdown_s": {k: round(v, 4) for k, v in TIMES.items()}}))
