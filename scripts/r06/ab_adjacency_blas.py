"""configs[4] loop with the camera-group adjacency from a matrix product (rounds 4-5) or from packed group sets (round 6):
one variant per process (the BLAS library's code objects load once per process).  usage: ab_adjacency_blas.py mm|packed"""
import contextlib, importlib.util, io, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vggsfm_amd import ba as BA
if sys.argv[1] == "mm":
    def adjacency_mm(masks, group, adjacency_reduce):
        S = masks.shape[0]; G = (S + group - 1) // group; pad = G * group - S
        m = torch.cat([masks, masks.new_zeros((pad, masks.shape[1]))]) if pad else masks
        V = m.reshape(G, group, -1).any(1).to(torch.float32)
        adj = ((V @ V.t()) > 0).to(torch.float32)
        if adjacency_reduce is not None:
            adjacency_reduce(adj)
        return adj.cpu() > 0
    BA._group_adjacency = adjacency_mm
spec = importlib.util.spec_from_file_location("run_c5_video", os.path.join(ROOT, "scripts", "run_c5_video.py"))
c5 = importlib.util.module_from_spec(spec); spec.loader.exec_module(c5)
with contextlib.redirect_stdout(io.StringIO()):
    out = c5.run_video()
print(json.dumps(dict(variant=sys.argv[1], total_seconds=round(out["total_seconds"], 3), window_ba_ms_mean=round(out["window_ba_ms_mean"], 2),
                      joint_ba_seconds_total=round(out["joint_ba_seconds_total"], 3), joint_ms=[round(j["ms"], 1) for j in out["joint_ba_log"]])))
