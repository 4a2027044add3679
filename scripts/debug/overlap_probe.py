"""Feasibility probe: can the (latency-bound, ~20-CU) Cholesky run on a second stream WHILE the Schur phase occupies the
chip?  Times phase 1 alone, the factorisation alone, both back to back, and both concurrently, with the tile launches
sized at 100 % and at 90 % of the resident capacity."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vggsfm_amd import _lib, ba as BA
from vggsfm_amd.ba_options import BundleAdjustmentOptions
from vggsfm_amd.dist import ShardedBA
from vggsfm_amd.scene import make_scene, perturb_for_ba
D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
L = _lib.lib()
sc = make_scene(200, 100000, "SIMPLE_RADIAL", shared_camera=True, seed=0)
ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=0)
n = 1202
rng = np.random.default_rng(0)
M = rng.normal(size=(n, n + 8)); A0 = M @ M.T + 1e-3 * np.eye(n)
Abuf0 = D(np.concatenate([np.tril(A0).ravel(), rng.normal(size=n)]))
Abuf = Abuf0.clone()
ws_ch = torch.empty(int(L.vgg_cholesky_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
fail = torch.zeros(1, dtype=torch.int32, device="cuda")
orig_build = BA.build_schur_tiles
hip = ctypes.CDLL("libamdhip64.so")


class RawStream:
    """HIP stream restricted to a set of CUs (hipExtStreamCreateWithCUMask); quacks like torch.cuda.Stream here."""

    def __init__(self, mask_words):
        self.h = ctypes.c_void_p()
        arr = (ctypes.c_uint32 * len(mask_words))(*mask_words)
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(self.h), len(mask_words), arr)
        assert rc == 0, rc
        self.cuda_stream = self.h.value


MASK_CHOL = [0xFFFFFFFF] + [0] * 7                     # 32 CUs for the factorisation
MASK_REST = [0] + [0xFFFFFFFF] * 7                     # 224 CUs for everything else

for frac in (1.0, 0.875):
    def patched(row_ptr, obs_cam, group=BA.GROUP, chunk=BA.CHUNK, max_chunks=None, _f=frac):
        if max_chunks is not None:
            max_chunks = tuple(int(m * _f) for m in max_chunks)
        return orig_build(row_ptr, obs_cam, group, chunk, max_chunks)
    BA.build_schur_tiles = patched
    prob, _, _ = BA.compile_problem(D(pts0), D(ext0), D(K0), D(sc.tracks), D(sc.mask), D(extra0), True, "SIMPLE_RADIAL")
    opts = BundleAdjustmentOptions(); opts.solver_options.max_num_iterations = 50
    s = ShardedBA(prob, opts)
    s.begin()
    for _ in range(3):
        s.iteration()
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def phase1(stream):
        _lib.check(L.vgg_ba_phase(ctypes.byref(s.cp), ctypes.byref(s.co), _lib.ptr(s.ws), 1, ctypes.c_void_p(stream.cuda_stream)), "p1")
    def chol(stream):
        _lib.check(L.vgg_cholesky_solve(_lib.ptr(Abuf), _lib.ptr(Abuf[n * n:]), n, _lib.ptr(ws_ch), _lib.ptr(fail), ctypes.c_void_p(stream.cuda_stream)), "ch")
    def timed(fn, reps=8):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    t_p1 = timed(lambda: phase1(s1))
    def chol_only():
        Abuf.copy_(Abuf0); chol(s2)
    t_ch = timed(lambda: (Abuf.copy_(Abuf0), torch.cuda.synchronize(), chol(s2)))
    def both_seq():
        Abuf.copy_(Abuf0); torch.cuda.synchronize(); phase1(s1); chol(s1)
    def both_conc():
        Abuf.copy_(Abuf0); torch.cuda.synchronize(); phase1(s1); chol(s2)
    m1, m2 = RawStream(MASK_REST), RawStream(MASK_CHOL)
    def both_masked():
        Abuf.copy_(Abuf0); torch.cuda.synchronize(); phase1(m1); chol(m2)
    t_p1m = timed(lambda: phase1(m1))
    t_chm = timed(lambda: (Abuf.copy_(Abuf0), torch.cuda.synchronize(), chol(m2)))
    print(f"tile slots x{frac}: phase1 {t_p1:.3f} ms, cholesky (+copy,sync) {t_ch:.3f} ms, sequential {timed(both_seq):.3f} ms, "
          f"concurrent {timed(both_conc):.3f} ms | CU-masked: phase1 on 224 CUs {t_p1m:.3f} ms, cholesky on 32 CUs {t_chm:.3f} ms, "
          f"concurrent {timed(both_masked):.3f} ms")
