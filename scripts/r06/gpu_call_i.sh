mkdir -p gpurun_out/r06i
python -m pytest tests/test_gpu_ba.py -x -q -k "cholesky or reproducible or trajectory or oracle" 2>&1 | tail -2
cd scripts/ubench
for r in 1 2; do for b in chol_bench_nosplit chol_bench chol_bench_occ2; do for args in "1202 384 288" "3200" "3200 768 672" "6002"; do echo "$b $args: $(timeout 60 ./$b $args | grep mode=0)"; done; done; done
./chol_trace 3200 | sed -n 15,40p
cd ../..
for v in default dfocc2 default dfocc2; do
  if [ $v = default ]; then unset VGGSFM_AMD_LIB; else export VGGSFM_AMD_LIB=$PWD/vggsfm_amd/_variants/lib_dfocc2.so; fi
  python scripts/run_c5_video.py --out gpurun_out/r06i/c5_$v.json > /dev/null 2>&1
  python - <<PY
import json; d=json.load(open("gpurun_out/r06i/c5_$v.json")); print("$v", round(d["final_joint_problem_iteration_ms"],4), "chol", round(d["final_joint_problem_kernel_ms"]["cholesky"],4))
PY
done
unset VGGSFM_AMD_LIB
python bench.py --workload c4shard --no-strong-leg --no-cpu-baseline --no-triangulation --no-pipeline > gpurun_out/r06i/bench_c4shard.json 2>/dev/null
python - <<PY
import json; d=json.loads(open("gpurun_out/r06i/bench_c4shard.json").read().strip().splitlines()[-1]); print("c4shard", d["ms_per_step"], d["config"]["kernel_ms"])
PY
