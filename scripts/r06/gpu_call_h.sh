mkdir -p gpurun_out/r06h
python -m pytest tests/test_gpu_ba.py -x -q -k "cholesky or reproducible" 2>&1 | tail -2
cd scripts/ubench
for r in 1 2; do for b in chol_bench_nopf chol_bench; do for args in "1202 384 288" "3200" "3200 768 672" "6002"; do echo "$b $args: $(timeout 60 ./$b $args | grep mode=0)"; done; done; done
cd ../..
python bench.py --workload c4shard --no-strong-leg --no-cpu-baseline --no-triangulation --no-pipeline > gpurun_out/r06h/bench_c4shard.json 2>/dev/null
python - <<PY
import json; d=json.loads(open("gpurun_out/r06h/bench_c4shard.json").read().strip().splitlines()[-1]); print("c4shard", d["ms_per_step"], d["config"]["kernel_ms"])
PY
