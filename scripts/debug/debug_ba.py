import sys
import numpy as np, torch
sys.path.insert(0, '.')
from oracle import ba as OB
from vggsfm_amd import ba as BA
from vggsfm_amd.ba_options import BundleAdjustmentOptions
from vggsfm_amd.scene import make_scene, perturb_for_ba
from vggsfm_amd.utils.triangulation_helpers import prepare_ba_options
D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
for (S, N, cam, shared, kind) in [(6, 60, "SIMPLE_PINHOLE", False, "prep")]:
    sc = make_scene(S, N, cam, shared_camera=shared, seed=S + N)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=S + N)
    oo = OB.prepare_ba_options() if kind == "prep" else OB.ceres_options()
    go = prepare_ba_options() if kind == "prep" else BundleAdjustmentOptions()
    po, eo, Ko, xo, so = OB.bundle_adjustment(pts0, ext0, K0, sc.tracks, sc.mask, extra0, shared, cam, oo)
    pts, ext, K, extra, sg = BA.bundle_adjustment(D(pts0), D(ext0), D(K0), D(sc.tracks), D(sc.mask), None, D(extra0), shared, cam, go)
    print("CASE", S, N, cam, shared, "oracle term", so["termination"], "its", so["num_iterations"], "gpu term", sg["termination"], "its", sg["num_iterations"])
    print("max ext diff", np.abs(ext.cpu().numpy() - eo).max(), "pts", np.abs(pts.cpu().numpy() - po).max(), "f", K[0,0,0].item(), Ko[0,0,0])
    for a, b in zip(sg["iterations"], so["iterations"]):
        print(f"{a['iteration']:3d} g:{a['successful']:d} o:{b['successful']:d} cost {a['cost']:.12e} {b['cost']:.12e} rad {a['radius']:.6e} {b['radius']:.6e} rel {a['relative_decrease']:.4e} {b['relative_decrease']:.4e} gmax {a['gradient_max_norm']:.4e} {b['gradient_max_norm']:.4e} step {a['step_norm']:.4e} {b['step_norm']:.4e}")
