import sys, ctypes
import numpy as np, torch
sys.path.insert(0, '.')
from oracle import ba as OB
from vggsfm_amd import ba as BA, _lib
from vggsfm_amd.scene import make_scene, perturb_for_ba
from vggsfm_amd.utils.triangulation_helpers import prepare_ba_options
D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
for (S, N, cam, shared) in [(6, 60, "SIMPLE_PINHOLE", False)]:
    sc = make_scene(S, N, cam, shared_camera=shared, seed=S + N)
    ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=S + N)
    kd = (1 if cam == "SIMPLE_PINHOLE" else 2)
    n = 6 * S + kd * (1 if shared else S)
    lhs = np.zeros((n, n)); rhs = np.zeros(n)
    OB.lib().bao_debug_dump_system(lhs.ctypes.data_as(ctypes.c_void_p), rhs.ctypes.data_as(ctypes.c_void_p))
    OB.bundle_adjustment(pts0, ext0, K0, sc.tracks, sc.mask, extra0, shared, cam, OB.ceres_options(1))
    prob, vi, dele = BA.compile_problem(D(pts0), D(ext0), D(K0), D(sc.tracks), D(sc.mask), D(extra0), shared, cam)
    L = _lib.lib()
    opts = prepare_ba_options(); cp = prob.c_struct(); co = BA._c_options(opts)
    nbytes = int(L.vgg_ba_workspace_bytes(ctypes.byref(cp), ctypes.byref(co)))
    ws = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    st = _lib.stream_ptr()
    _lib.check(L.vgg_ba_begin(ctypes.byref(cp), ctypes.byref(co), _lib.ptr(ws), ctypes.c_size_t(nbytes), 0, 1, st), "begin")
    for ph in (0, 1):
        _lib.check(L.vgg_ba_phase(ctypes.byref(cp), ctypes.byref(co), _lib.ptr(ws), ph, st), "phase")
    # phase 2 starts with fix_constant; emulate by comparing only active rows
    p = ctypes.POINTER(ctypes.c_double)(); cnt = ctypes.c_size_t()
    L.vgg_ba_reduce_buffer(ctypes.byref(cp), ctypes.byref(co), _lib.ptr(ws), 1, ctypes.byref(p), ctypes.byref(cnt))
    off = ctypes.addressof(p.contents) - ws.data_ptr()
    sys_buf = ws[off:off + 8 * cnt.value].view(torch.float64).cpu().numpy()
    Sg = sys_buf[:n * n].reshape(n, n); rg = sys_buf[n * n:]
    Sl = np.tril(Sg); So = np.tril(lhs)
    act = np.ones(n, bool); act[:6] = False; act[9] = False
    M = np.outer(act, act)
    diff = np.abs(Sl - So) * M
    rel = diff / (np.abs(So) + 1e-300)
    print("CASE", S, N, cam, shared, "n", n, "max abs diff", diff.max(), "at", np.unravel_index(diff.argmax(), diff.shape), "rhs diff", np.abs((rg - rhs) * act).max())
    bad = np.argwhere((diff > 1e-6 * np.abs(So).max()))
    print("  num bad", len(bad), "all", bad.tolist())
    for (i, j) in bad[:6]:
        print("   ", i, j, Sl[i, j], So[i, j])
