#!/bin/bash
# round 4, first GPU call: the whole gpu suite, the default bench line (with the new pipeline leg), configs[3] whole on one GPU
# with its per-kernel record (HIP events in the line + rocprofv3 kernel stats)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04a
mkdir -p $OUT
cd $ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -25 > $OUT/tests.log
python bench.py > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python bench.py --workload c4full --no-cpu-baseline --no-strong-leg --no-pipeline --no-triangulation --steps 10 --warmup 3 > $OUT/bench_c4full.json 2> $OUT/bench_c4full.err
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_c4
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -o c4 -- python $ROOT/bench.py --workload c4full --no-cpu-baseline --no-strong-leg --no-pipeline --no-triangulation --steps 10 --warmup 3 > $OUT/bench_c4full_under_rocprof.json 2>/dev/null
cp $(find /tmp/prof_c4 -name "*kernel_stats.csv" | head -1) $OUT/c4full_kernel_stats.csv
