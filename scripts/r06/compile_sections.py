import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from vggsfm_amd import ba as BA
from vggsfm_amd.scene import make_scene, perturb_for_ba
D = lambda x: None if x is None else torch.from_numpy(np.ascontiguousarray(x)).cuda()
sc = make_scene(17, 3000, "SIMPLE_RADIAL", shared_camera=True, seed=3)
ext0, K0, xp0, pts0 = perturb_for_ba(sc, seed=3)
args = [D(x) for x in (pts0, ext0, K0, sc.tracks, sc.mask, xp0)]
acc = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter(); out = fn(*a, **k); torch.cuda.synchronize()
        acc[name] = acc.get(name, 0.0) + 1e3 * (time.perf_counter() - t); return out
    return w
BA.build_schur_tiles = timed("build_schur_tiles", BA.build_schur_tiles)
for n in ("find_camera_order", "find_camera_split", "_group_adjacency"):
    if hasattr(BA, n): setattr(BA, n, timed(n, getattr(BA, n)))
kw = dict(gauge="config", camera_split=True, refine_focal_length=False, refine_extra_params=False, sort_points=True, filter_negative_depth=False)
for rep in range(5):
    acc.clear()
    torch.cuda.synchronize(); t = time.perf_counter()
    BA.compile_problem(*args, True, "SIMPLE_RADIAL", **kw)
    torch.cuda.synchronize(); tot = 1e3 * (time.perf_counter() - t)
print(json.dumps(dict(total_ms=round(tot, 3), **{k: round(v, 3) for k, v in acc.items()})))
kw["sort_points"] = False
for rep in range(5):
    torch.cuda.synchronize(); t = time.perf_counter()
    BA.compile_problem(*args, True, "SIMPLE_RADIAL", **kw)
    torch.cuda.synchronize(); tot2 = 1e3 * (time.perf_counter() - t)
print("without sort_points", round(tot2, 3))
