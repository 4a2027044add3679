#!/bin/bash
# Same-box A/B of LIBRARY builds on the bench workload: every variant = a .so under vggsfm_amd/_variants/ (or "product" = the
# in-tree library), interleaved over ROUNDS rounds.   usage: ab_lib.sh OUT.jsonl name[:path] ...      (on the GPU box)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; shift
ROUNDS=${ROUNDS:-2}
ARGS=${BENCH_ARGS:---steps 20 --warmup 5 --no-cpu-baseline --no-strong-leg --no-triangulation --no-pipeline}
: > $OUT
cd $ROOT
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    name=${v%%:*}; path=${v#*:}
    if [ "$name" = "product" ]; then unset VGGSFM_AMD_LIB; else export VGGSFM_AMD_LIB=$ROOT/${path}; fi
    python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print(json.dumps(dict(variant='$name', round=$r, ms_per_step=d['ms_per_step'], value=d['value'], kernel_ms=d['config']['kernel_ms'], accepted=d['config'].get('successful_steps_last_episode'))))" | tee -a $OUT
  done
done
unset VGGSFM_AMD_LIB
