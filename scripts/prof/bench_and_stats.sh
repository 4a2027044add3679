# the default bench line + the rocprofv3 kernel-trace summary of the same workload (profiles/r02_*), and the shader clock
# seen by a latency-bound loop right after the run (scripts/ubench/clock_probe)
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
scripts/ubench/clock_probe > gpurun_out/final_clock_probe.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_final
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -o r02 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strong-leg > $GRAFT_REPO_ROOT/gpurun_out/final_bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/prof_final -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/final_kernel_stats.csv
