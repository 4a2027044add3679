import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ba as OB
from vggsfm_amd import ba as BA
from vggsfm_amd.scene import make_scene, perturb_for_ba
D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
cam, shared = sys.argv[1], True
S, N, n_exist = 17, 900, 500
sc = make_scene(S, N, cam, shared_camera=shared, seed=12, full_visibility=False, outlier_frac=0.0)
ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=12)
ext0[0] = sc.extrinsics[0]
valid = sc.mask.sum(0) >= 2
order = np.cumsum(valid) - 1
constant = valid & (order < n_exist)
pts0[constant] = sc.points3D[constant]
po, eo, Ko, xo, so = OB.bundle_adjustment(pts0, ext0, K0, sc.tracks, sc.mask, extra0, shared, cam, options=OB.ceres_options(100),
                                          constant_points=constant, constant_pose_frames=[0], filter_negative_depth=False,
                                          refine_focal=False, refine_extra=False)
pts, ext, K, extra, sg = BA.window_bundle_adjustment(D(pts0), D(ext0), D(K0), D(sc.tracks), D(sc.mask), n_exist, D(extra0), shared, cam)
for a, b in zip(so["iterations"], sg["iterations"]):
    print(a["iteration"], f"{a['cost']:.9e} {b['cost']:.9e} | grad {a['gradient_max_norm']:.3e} {b['gradient_max_norm']:.3e} | step {a['step_norm']:.3e} {b['step_norm']:.3e} | rad {a['radius']:.2e} {b['radius']:.2e} | {a['successful']} {b['successful']}")
print(so["termination"], sg["termination"], sg.get("termination_str"), so["n_reduced"], sg["n_reduced"])
for b in sg["iterations"][len(so["iterations"]):]:
    print(b["iteration"], f"{b['cost']:.9e} grad {b['gradient_max_norm']:.3e} step {b['step_norm']:.3e} rad {b['radius']:.2e} {b['successful']}")
