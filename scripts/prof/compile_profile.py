"""cProfile of ba.compile_problem on a window-sized problem (host overhead of the torch ops and their synchronisations)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from vggsfm_amd import ba as BA
from vggsfm_amd.scene import make_scene, perturb_for_ba

dev = torch.device("cuda:0")
S_, N_ = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (33, 8000)
sc = make_scene(S_, N_, "SIMPLE_RADIAL", shared_camera=True, seed=1)
ext0, K0, extra0, pts0 = perturb_for_ba(sc, seed=1)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
args = (T(pts0), T(ext0), T(K0), T(sc.tracks), T(sc.mask), T(extra0), True, "SIMPLE_RADIAL")
for _ in range(2):
    BA.compile_problem(*args, camera_split=True)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
REPS = 20 if N_ < 50000 else 5
for _ in range(REPS):
    BA.compile_problem(*args, camera_split=True)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
