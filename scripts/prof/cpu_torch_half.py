#!/usr/bin/env python
"""CPU baseline of the TORCH half of the path: ``triangulate_tracks``, ``filter_all_points3D``,
``project_3D_points``, ``cam_from_img`` timed on the host cores, beside the GPU numbers of scripts/prof/bench_geometry.py.

* where the reference tree exists (build container): the REFERENCE's own functions through oracle/ref_harness.py
  (``kind: "reference"``), torch threads = all cores;
* elsewhere (GPU box): the numpy restatement oracle/geometry.py (``kind: "port"``).
Both on a bounded slice of the workload (default 1000 tracks) -- the cost is linear in the tracks, the whole-workload
figure is the slice time x tracks / slice, labelled extrapolated.

    python scripts/prof/cpu_torch_half.py [--config c2|c3] [--slice 1000] [--out profiles/...json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import geometry as OG  # noqa: E402
from oracle import ref_harness  # noqa: E402
from vggsfm_amd.scene import make_scene  # noqa: E402

CONFIGS = {"c2": (50, 20000, "SIMPLE_PINHOLE", False), "c3": (200, 100000, "SIMPLE_RADIAL", True)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--slice", type=int, default=1000)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    S, N, cam, shared = CONFIGS[a.config]
    n = min(a.slice, N)
    sc = make_scene(S, n, cam, shared_camera=shared, seed=2, outlier_frac=0.05)
    cores = os.cpu_count()
    torch.set_num_threads(cores)
    res = {}
    if ref_harness.available():
        import warnings
        kind = "reference"
        tri, helpers, _ = ref_harness.load()
        T = torch.from_numpy
        ext, K = T(sc.extrinsics), T(sc.intrinsics)
        extra = None if sc.extra_params is None else T(sc.extra_params)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t0 = time.perf_counter(); tn = helpers.cam_from_img(T(sc.tracks), K, extra); res["cam_from_img"] = time.perf_counter() - t0
            tn = tn.transpose(0, 1).contiguous().transpose(0, 1)
            torch.manual_seed(0)
            t0 = time.perf_counter()
            pts, num, msk = tri.triangulate_tracks(ext, tn, track_vis=T(sc.vis).transpose(0, 1).contiguous().transpose(0, 1),
                                                   track_score=T(sc.score).transpose(0, 1).contiguous().transpose(0, 1))
            res["triangulate_tracks"] = time.perf_counter() - t0
            t0 = time.perf_counter(); helpers.filter_all_points3D(pts, T(sc.tracks), ext, K, extra_params=extra); res["filter_all_points3D"] = time.perf_counter() - t0
            t0 = time.perf_counter(); helpers.project_3D_points(pts, ext, K, extra); res["project_3D_points"] = time.perf_counter() - t0
    else:
        kind = "port"
        cores = 1                                                  # numpy: one thread for everything that matters here
        tr = sc.tracks.astype(np.float64)
        t0 = time.perf_counter(); tn = OG.cam_from_img(tr, sc.intrinsics, sc.extra_params); res["cam_from_img"] = time.perf_counter() - t0
        comb = OG.generate_combinations(S)
        torch.manual_seed(0)
        pairs = comb if len(comb) <= 256 else comb[torch.randperm(len(comb))[:256].numpy()]
        t0 = time.perf_counter(); pts, num, msk = OG.triangulate_tracks_chunk(sc.extrinsics, tn, pairs, track_vis=sc.vis, track_score=sc.score); res["triangulate_tracks"] = time.perf_counter() - t0
        t0 = time.perf_counter(); OG.filter_all_points3D(pts, tr, sc.extrinsics, sc.intrinsics, sc.extra_params); res["filter_all_points3D"] = time.perf_counter() - t0
        t0 = time.perf_counter(); OG.project_3D_points(pts, sc.extrinsics, sc.intrinsics, sc.extra_params); res["project_3D_points"] = time.perf_counter() - t0
    out = dict(config=a.config, frames=S, tracks=N, slice_tracks=n, kind=kind, cores=cores,
               what=("the reference's own torch functions (vggsfm/utils/triangulation.py, triangulation_helpers.py) on the CPU"
                     if kind == "reference" else "numpy restatement oracle/geometry.py on the CPU"),
               seconds_slice=res, seconds_full_extrapolated={k: v * N / n for k, v in res.items()},
               tracks_per_second={k: n / v for k, v in res.items()})
    print(json.dumps(out))
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
