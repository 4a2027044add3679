#!/bin/bash
# rocprofv3 --kernel-trace --stats of the BA loop of the bench workload alone: average duration of every vgg:: launch of an
# LM iteration (the small ones are not in bench.py's own HIP-event list).  usage: gpurun -- bash scripts/prof/kernel_stats.sh [workload]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-c3}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o ks -- python $ROOT/bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-strong-leg --no-pipeline --no-triangulation > /tmp/ks_line.json 2>/dev/null
python - <<PY
import csv, glob, json
f = glob.glob("/tmp/prof_ks/**/*kernel_stats.csv", recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    if "vgg::" in r["Name"] and int(r["Calls"]) >= 20:
        per = float(r["AverageNs"]) * int(r["Calls"]) / 25 / 1e3
        tot += per
        print(f"{r['Name'][:64]:64s} {int(r['Calls']):4d} x {float(r['AverageNs'])/1e3:8.1f} us")
print("sum of vgg launches per iteration: %.1f us" % tot)
d = json.load(open("/tmp/ks_line.json"))
print("bench line under rocprof: %.4f ms per iteration" % d["ms_per_step"])
PY
