#!/bin/bash
# SQ counters of triangulate_kernel at BASELINE configs[2] size (one rocprofv3 --pmc pass per counter group, kernel trace only)
# -> gpurun_out/pmc/r05_pmc_sq_tri.json (committed copy: profiles/r05_pmc_sq_tri.json, read by bench.py's roofline_by_kernel).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/tri_once.py <<PY
import sys, numpy as np, torch
sys.path.insert(0, "$ROOT")
from vggsfm_amd.scene import make_scene
from vggsfm_amd.utils.triangulation import triangulate_tracks
from vggsfm_amd.utils.triangulation_helpers import cam_from_img
dev = torch.device("cuda:0")
sc = make_scene(200, 100000, "SIMPLE_RADIAL", shared_camera=True, seed=0)
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
tn = cam_from_img(T(sc.tracks), T(sc.intrinsics), T(sc.extra_params))
for _ in range(2):
    torch.manual_seed(0)
    triangulate_tracks(T(sc.extrinsics), tn, track_vis=T(sc.vis), track_score=T(sc.score))
torch.cuda.synchronize()
PY
i=0
for CTRS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
            "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_WAVES"; do
  i=$((i+1)); rm -rf /tmp/pmc_tri_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/pmc_tri_$i -- python /tmp/tri_once.py > $OUT/tri_pass_$i.log 2>&1
done
python $ROOT/scripts/prof/pmc_aggregate.py /tmp/pmc_tri_all.json /tmp/pmc_tri_1 /tmp/pmc_tri_2 > /dev/null 2>&1
python - <<PY
import json
d = json.load(open("/tmp/pmc_tri_all.json"))
k = [n for n in d if "triangulate_kernel" in n]
out = {"kernel": k[0] if k else None, "workload": "triangulate_tracks, 200 views x 100000 tracks (BASELINE configs[2] scene), 25 reference chunks in 5 launches",
       "units": "per call (sum over its launches would be x5: values are means per LAUNCH); SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* in quad-cycles summed over waves or SIMDs (MI355X_MICROARCH.md)",
       "counters": {c: v["mean"] for c, v in (d[k[0]].items() if k else [])}, "launches_seen": {c: v["n"] for c, v in (d[k[0]].items() if k else [])}}
c = out["counters"]
if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c:
    out["valu_active_quadcycles_per_wave_quadcycle"] = c["SQ_ACTIVE_INST_VALU"] / max(c.get("SQ_WAVE_CYCLES", 1), 1)
json.dump(out, open("$OUT/r05_pmc_sq_tri.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY
