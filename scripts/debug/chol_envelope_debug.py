"""Per-block error of the envelope Cholesky solve against LAPACK (debug aid for tests/test_gpu_ba.py::test_cholesky_envelope_*)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vggsfm_amd import _lib

def D(x): return torch.from_numpy(np.ascontiguousarray(x)).cuda()

def case(n, band, arrow, dense=False):
    rng = np.random.default_rng(n); nb = n - arrow
    M = np.zeros((n, n))
    for i in range(nb):
        lo = max(0, i - band); M[i, lo:i] = rng.normal(size=i - lo) * 0.05
    if arrow: M[nb:, :] = np.tril(rng.normal(size=(arrow, n)) * 0.05)
    A = M + M.T; A[np.arange(n), np.arange(n)] = np.abs(A).sum(1) + 1.0
    k = max(2, min(5, nb // (3 * band))); rem = nb - (k - 1) * band
    sizes = [rem // k + (1 if j < rem % k else 0) for j in range(k)]
    interiors, seps, pos = [], [], 0
    for j in range(k):
        interiors += list(range(pos, pos + sizes[j])); pos += sizes[j]
        if j < k - 1: seps += list(range(pos, pos + band)); pos += band
    order = np.array(interiors + seps + list(range(nb, n)))
    Ap = A[order][:, order]
    first_col = np.array([np.nonzero(Ap[i, :i + 1])[0][0] for i in range(n)])
    nbk = (n + 63) // 64
    fc = np.concatenate([first_col, np.full(nbk * 64 - n, n)])
    fb = (fc.reshape(nbk, 64).min(1) // 64).astype(np.int32)
    if dense: fb[:] = 0
    b = rng.normal(size=n)
    L = _lib.lib()
    ws = torch.empty(int(L.vgg_cholesky_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
    for rep in range(3):
        buf = D(np.concatenate([np.tril(Ap).ravel(), b])); fail = torch.zeros(1, dtype=torch.int32, device="cuda")
        At, bt = buf[:n * n], buf[n * n:]
        rc = L.vgg_cholesky_solve_envelope(_lib.ptr(At), _lib.ptr(bt), n, _lib.ptr(D(fb)), _lib.ptr(ws), _lib.ptr(fail), _lib.stream_ptr())
        torch.cuda.synchronize()
        x = bt.cpu().numpy(); xr = np.linalg.solve(Ap, b)
        Lr = np.linalg.cholesky(Ap); Lg = np.tril(At.cpu().numpy().reshape(n, n))
        z = np.linalg.solve(Lr, b)
        errx = [float(np.abs(x[64*c:64*c+64] - xr[64*c:64*c+64]).max()) for c in range(nbk)]
        errL = float(np.abs(Lg - Lr).max())
        print(n, band, arrow, "dense" if dense else "env", "rep", rep, "rc", rc, "fail", int(fail), "errL %.1e" % errL, "first bad block", next((c for c in range(nbk-1, -1, -1) if errx[c] > 1e-8), None), ["%.0e" % e for e in errx])
    print("first_blk", fb.tolist())

for c in [(1202, 300, 2), (900, 100, 0), (2000, 200, 130)]:
    case(*c); case(*c, dense=True)
