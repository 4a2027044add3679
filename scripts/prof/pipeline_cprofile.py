"""cProfile of ONE vggsfm_amd.models.Triangulator.forward at BASELINE configs[2] (host view: where the Python side spends the
wall time -- device work shows up at the synchronisation points).  usage: python scripts/prof/pipeline_cprofile.py [c2|c3]"""
import contextlib
import cProfile
import io
import os
import pstats
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
from vggsfm_amd.models import Triangulator  # noqa: E402
from vggsfm_amd.scene import make_scene, perturb_for_ba  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
S, N, cam, shared = B.WORKLOADS[name]
dev = torch.device("cuda", 0)
W = 1024
sc = make_scene(S, N, cam, shared_camera=shared, seed=0)
ext0, K0, _, _ = perturb_for_ba(sc, seed=0, rot_deg=0.5, trans=0.02, focal_rel=0.02)
cams = types.SimpleNamespace(R=B.D(ext0[:, :, :3], dev).float(), T=B.D(ext0[:, :, 3], dev).float(),
                             focal_length=torch.stack([B.D(K0[:, 0, 0] / (W / 2.0), dev).float()] * 2, -1))
img = types.SimpleNamespace(shape=(1, S, 3, W, W))
tracks, vis, score = B.D(sc.tracks, dev)[None], B.D(sc.vis, dev)[None], B.D(sc.score, dev)[None]
prelim = {"fmat_inlier_mask": B.D(sc.mask[1:] & sc.mask[0:1], dev)[None]}
tri = Triangulator()
kw = dict(pred_score=score, shared_camera=shared, camera_type=cam, BA_iters=2, robust_refine=2, extract_color=False)
with contextlib.redirect_stdout(io.StringIO()):
    torch.manual_seed(0)
    tri(cams, tracks, vis, img, prelim, **kw)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    torch.manual_seed(0)
    pr.enable()
    tri(cams, tracks, vis, img, prelim, **kw)
    torch.cuda.synchronize()
    pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumtime").print_stats(40)
# device view of the same call: which kernels the forward spends its GPU time in (torch ops of compile_problem and the
# drop-in's own launches side by side)
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with contextlib.redirect_stdout(io.StringIO()):
    torch.manual_seed(0)
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        tri(cams, tracks, vis, img, prelim, **kw)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=70))
