#!/bin/bash
# The measurements a round commits under profiles/ (run on the GPU box: gpurun -- bash scripts/prof/record.sh r04):
#   <tag>_gputest_tail.txt              tail of pytest -m gpu
#   <tag>_bench_c3.json                 the default bench line
#   <tag>_bench_c3_kernel_stats.csv     rocprofv3 --kernel-trace --stats of the same command (triangulation leg included)
#   <tag>_pmc_fetch_write_c3.json       FETCH_SIZE / WRITE_SIZE per kernel, separate --pmc passes (scripts/prof/pmc_traffic.sh)
#   <tag>_bench_c4full.json (+ _kernel_stats.csv)   configs[3] whole on one GPU
#   <tag>_c5_video.json, <tag>_c1_kitchen.json      configs[4] (1000-frame video loop) and configs[0] (kitchen plumbing) once
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python -m pytest tests -q -m gpu 2>&1 | tail -6 > $OUT/${TAG}_gputest_tail.txt
python bench.py > $OUT/${TAG}_bench_c3.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_rec
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rec -o rec -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strong-leg --no-pipeline > $OUT/${TAG}_bench_c3_under_rocprof.json 2>/dev/null
cp $(find /tmp/prof_rec -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_c3_kernel_stats.csv
cd $ROOT
bash $ROOT/scripts/prof/pmc_traffic.sh > $OUT/pmc.log 2>&1
cp $ROOT/gpurun_out/pmc/pmc_summary.json $OUT/${TAG}_pmc_fetch_write_c3.json
python bench.py --workload c4full --no-cpu-baseline --no-strong-leg --no-pipeline --no-triangulation --steps 10 --warmup 3 > $OUT/${TAG}_bench_c4full.json 2> $OUT/bench_c4full.err
cd /tmp && rm -rf /tmp/prof_c4
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -o c4 -- python $ROOT/bench.py --workload c4full --no-cpu-baseline --no-strong-leg --no-pipeline --no-triangulation --steps 10 --warmup 3 > /dev/null 2>&1
cp $(find /tmp/prof_c4 -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_c4full_kernel_stats.csv
cd $ROOT
#   <tag>_bench_c4shard.json (+ _kernel_stats.csv)  one 8-GPU shard of configs[3] (37 500 of the 300 000 tracks): the T_shard of DESIGN.md section 8
python bench.py --workload c4shard --no-cpu-baseline --no-strong-leg --no-pipeline --no-triangulation --steps 20 --warmup 5 > $OUT/${TAG}_bench_c4shard.json 2> $OUT/bench_c4shard.err
cd /tmp && rm -rf /tmp/prof_c4s
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4s -o c4s -- python $ROOT/bench.py --workload c4shard --no-cpu-baseline --no-strong-leg --no-pipeline --no-triangulation --steps 10 --warmup 3 > /dev/null 2>&1
cp $(find /tmp/prof_c4s -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_c4shard_kernel_stats.csv
cd $ROOT
python scripts/prof/compile_profile.py > $OUT/compile_profile.txt 2>&1
python scripts/run_c5_video.py --out $OUT/${TAG}_c5_video.json > $OUT/c5.log 2>&1
python scripts/run_c1_kitchen.py --stage run --out $OUT/${TAG}_c1_kitchen.json > $OUT/c1.log 2>&1
#   <tag>_timeline_c3.json              every launch of an LM iteration with its duration and the gap in front of it
#   <tag>_pmc_sq_tri.json               SQ counters of triangulate_kernel (bench.py reads the committed copy)
bash $ROOT/scripts/prof/timeline.sh > $OUT/timeline.log 2>&1
cp $ROOT/gpurun_out/timeline/timeline_summary.json $OUT/${TAG}_timeline_c3.json
bash $ROOT/scripts/prof/pmc_sq_tri.sh > $OUT/pmc_sq_tri.log 2>&1
cp $(ls $ROOT/gpurun_out/pmc/*pmc_sq_tri.json | head -1) $OUT/${TAG}_pmc_sq_tri.json
