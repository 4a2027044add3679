"""A/B of the CU-masked overlap (factorisation beside the later Schur tile batches) on BASELINE configs[2]:
same start, K LM iterations, (a) one batch / one stream, (b) three batches + overlap, (b) twice.
Checks: (b) is bit-reproducible run to run (no race between the producers of S2 and the panels that read it);
(a) and (b) agree to rounding (the lazily added S2 changes the summation order only)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vggsfm_amd import ba as BA
from vggsfm_amd.ba_options import BundleAdjustmentOptions
from vggsfm_amd.dist import ShardedBA
from vggsfm_amd.scene import make_scene, perturb_for_ba

D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
S, N, K = int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 100000, 12
sc = make_scene(S, N, "SIMPLE_RADIAL", shared_camera=True, seed=0)
ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=0)


def run(overlap, cus=None):
    if cus is not None:
        BA.CHOL_CUS = cus
    prob, _, _ = BA.compile_problem(D(pts0), D(ext0), D(K0), D(sc.tracks), D(sc.mask), D(extra0), True, "SIMPLE_RADIAL",
                                    overlap=overlap)
    opts = BundleAdjustmentOptions()
    opts.solver_options.max_num_iterations = K
    for name in ("function_tolerance", "gradient_tolerance", "parameter_tolerance"):
        setattr(opts.solver_options, name, -1.0)
    s = ShardedBA(prob, opts)
    s.begin()
    for _ in range(3):
        s.iteration()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K - 3):
        s.iteration()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / (K - 3) * 1e3
    fin = s.finish(K + 2)
    return dt, fin, [t.clone() for t in (prob.cam_q, prob.cam_t, prob.intr, prob.pts)], prob.batch_desc.shape[0]


ta, fa, xa, _ = run(False)
tb, fb, xb, nb = run(True)
tc, fc, xc, _ = run(True)
print(f"{S}x{N}: one batch {ta:.3f} ms/iter | {nb} batches + overlap ({BA.CHOL_CUS} CUs) {tb:.3f} / {tc:.3f} ms/iter")
print("final cost   a", fa["final_cost"], "b", fb["final_cost"], "c", fc["final_cost"], "iters", fa["num_iterations"], fb["num_iterations"])
print("bit-identical overlap runs:", all(torch.equal(p, q) for p, q in zip(xb, xc)))
for name, p, q in zip(("cam_q", "cam_t", "intr", "pts"), xa, xb):
    print(f"  a vs b {name}: max abs diff {float((p - q).abs().max()):.3e} (scale {float(p.abs().max()):.3e})")
assert abs(fa["final_cost"] - fb["final_cost"]) <= 1e-9 * abs(fa["final_cost"])
assert all(torch.equal(p, q) for p, q in zip(xb, xc))
if len(sys.argv) > 3:
    for cus in (64,):
        td, fd, xd, _ = run(True, cus)
        print(f"  factorisation on {cus} CUs: {td:.3f} ms/iter, cost {fd['final_cost']}")
