#!/bin/bash
# Kernel timeline of the bench workload: every launch of a few LM iterations with start / end timestamps, to see what the
# time OUTSIDE the seven profiled kernels is made of (small launches vs gaps between dependent launches).
#   -> gpurun_out/timeline/kernel_trace.csv (+ timeline_summary.json via scripts/prof/timeline_summary.py)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/timeline
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl
(cd $ROOT && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-strong-leg --no-triangulation --no-pipeline > $OUT/bench.json 2> $OUT/bench.err)
cp $(find /tmp/tl -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv
python $ROOT/scripts/prof/timeline_summary.py $OUT/kernel_trace.csv $OUT/timeline_summary.json
