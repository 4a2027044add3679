"""Every n from 1 to 200 (+ a few larger) through the dataflow factorisation with the right-hand side stored behind the matrix
(the BA path), twice per workspace (the hand-off buffers are reused), against numpy."""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from vggsfm_amd import _lib
L = _lib.lib()
bad = 0
for n in list(range(1, 201)) + [255, 256, 257, 320, 511, 513, 1000]:
    rng = np.random.default_rng(n)
    ws = torch.empty(int(L.vgg_cholesky_workspace_bytes(n)), dtype=torch.uint8, device="cuda")
    for rep in range(2):
        M = rng.normal(size=(n, n + 8)); A = M @ M.T + 1e-3 * np.eye(n); b = rng.normal(size=n)
        buf = torch.from_numpy(np.concatenate([np.tril(A).ravel(), b])).cuda()
        fail = torch.zeros(1, dtype=torch.int32, device="cuda")
        rc = L.vgg_cholesky_solve(_lib.ptr(buf[:n * n]), _lib.ptr(buf[n * n:]), n, _lib.ptr(ws), _lib.ptr(fail), _lib.stream_ptr())
        x = buf[n * n:].cpu().numpy(); xr = np.linalg.solve(A, b)
        ok = rc == 0 and int(fail.item()) == 0 and np.allclose(x, xr, rtol=1e-8, atol=1e-10 * np.abs(xr).max())
        if not ok: bad += 1; print("FAIL n", n, "rep", rep, rc, int(fail.item()), float(np.abs(x - xr).max()))
print("bad", bad)
