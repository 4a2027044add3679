# depth-ordered updates: standalone factorisation, its tests, c3 / c4shard bench and the c5 video against the previous build
# (the "head" legs need vggsfm_amd/_variants/lib_head.so: scripts/prof/build_ref_variant.sh <git-ref>)
scripts/ubench/chol_bench 1202 384 288 | grep "mode=0\|residual"
scripts/ubench/chol_bench 3200 1024 960 | grep "mode=0\|residual"
scripts/ubench/chol_bench_trace 1202 384 288 | tail -12
python -m pytest tests/test_gpu_ba.py -q -x -m gpu -k "cholesky or kway or camera_split" 2>&1 | tail -2
for v in new head; do
  if [ $v = head ]; then export VGGSFM_AMD_LIB=$PWD/vggsfm_amd/_variants/lib_head.so; else unset VGGSFM_AMD_LIB; fi
  for wl in c3 c4shard; do
    python bench.py --workload $wl --no-cpu-baseline --no-strong-leg --steps 30 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v $wl', round(d['value'],1), round(d['ms_per_step'],4), round(d['config']['kernel_ms']['cholesky'],4))"
  done
  python scripts/run_c5_video.py 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v c5', round(d['total_seconds'],3), round(d['final_joint_problem_iteration_ms'],3), round(d['final_joint_problem_kernel_ms']['cholesky'],3))"
done
