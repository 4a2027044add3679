"""Where does an LM iteration of a SMALL problem go?  (video window BA: 17 frames, ~1.5 k points, n = 102: 0.83 ms per iteration
in round 5.)  Wall time per iteration of whole solves (compile excluded), per-kernel HIP-event sums, for a few sizes."""
import ctypes, json, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vggsfm_amd import _lib, ba as BA
from vggsfm_amd.ba_options import BundleAdjustmentOptions
from vggsfm_amd.scene import make_scene, perturb_for_ba
import bench as B
D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
L = _lib.lib()
SIZES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(17, 1500), (17, 6000), (33, 3000), (50, 20000)]
for (S, N) in SIZES:
    sc = make_scene(S, N, "SIMPLE_RADIAL", shared_camera=True, seed=3)
    ext0, K0, xp0, pts0 = perturb_for_ba(sc, seed=3)
    opt = BundleAdjustmentOptions()
    opt.refine_focal_length = opt.refine_extra_params = False
    so = opt.solver_options
    so.max_num_iterations = 20
    so.function_tolerance = so.gradient_tolerance = so.parameter_tolerance = -1.0
    args = [D(x) for x in (pts0, ext0, K0, sc.tracks, sc.mask, xp0)]
    prob, _, _ = BA.compile_problem(*args, True, "SIMPLE_RADIAL", gauge="config", camera_split=True, refine_focal_length=False, refine_extra_params=False, sort_points=True)
    prob.cam_const[0] = 1
    init = [t.clone() for t in (prob.cam_q, prob.cam_t, prob.intr, prob.pts)]
    ws = None
    res = {}
    for prof in (0, 1):
        L.vgg_ba_profile(prof, 4096)
        for rep in range(3):
            for dst, src in zip((prob.cam_q, prob.cam_t, prob.intr, prob.pts), init):
                dst.copy_(src)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            summ, ws = BA.solve(prob, opt, ws)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        its = summ["num_iterations"] + 1
        res["wall_ms_per_iteration" + ("_with_events" if prof else "")] = round(1e3 * dt / its, 4)
        if prof:
            kms = {}
            for kid, kname in enumerate(B.KERNELS):
                tot, n = ctypes.c_double(), ctypes.c_int()
                L.vgg_ba_profile_read(kid, ctypes.byref(tot), ctypes.byref(n), 1)
                kms[kname] = round(tot.value / 3 / its, 4)
            res["kernel_ms"] = kms
            res["kernel_sum_ms"] = round(sum(kms.values()), 4)
    L.vgg_ba_profile(0, 0)
    print(json.dumps(dict(frames=S, tracks=N, obs=int(prob.num_obs), n_reduced=summ["n_reduced"], iterations=its, merged=bool(prob.merged_tile_launch), **res)), flush=True)
