set -x
python -m pytest tests -q -x -m gpu 2>&1 | tail -3 > gpurun_out/final_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1
python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_final
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -o r02 -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/final_bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/prof_final -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/final_kernel_stats.csv
