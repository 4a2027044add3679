#!/bin/bash
# The measurements a round commits under profiles/ (run on the GPU box: gpurun -- bash scripts/prof/record.sh r03):
#   <tag>_bench_c3.json                 the default bench line
#   <tag>_bench_c3_kernel_stats.csv     rocprofv3 --kernel-trace --stats of the same command (triangulation leg included)
#   <tag>_pmc_fetch_write_c3.json       FETCH_SIZE / WRITE_SIZE per kernel, separate --pmc passes (scripts/prof/pmc_traffic.sh)
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/${TAG}_bench_c3.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_rec
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rec -o rec -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strong-leg > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/prof_rec -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_c3_kernel_stats.csv
bash $ROOT/scripts/prof/pmc_traffic.sh > $OUT/pmc.log 2>&1
cp $ROOT/gpurun_out/pmc/pmc_summary.json $OUT/${TAG}_pmc_fetch_write_c3.json
