#!/bin/bash
# SQ-side counters of the bench workload (one pass, kernel-trace only), aggregated per kernel.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq
CTRS=${1:-"SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"}
(cd $ROOT && rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/pmc_sq -- \
    python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-strong-leg --no-triangulation --no-pipeline > $OUT/bench_sq.log 2>&1)
python $ROOT/scripts/prof/pmc_aggregate.py $OUT/pmc_sq.json /tmp/pmc_sq > /dev/null
python - <<PY
import json
d = json.load(open("$OUT/pmc_sq.json"))
for k, v in d.items():
    if "vgg::" in k:
        print(k[:70], {c: round(x["mean"]) for c, x in v.items()})
PY
