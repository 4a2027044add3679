mkdir -p gpurun_out/r06c
python -m pytest tests/test_gpu_ba.py -x -q -k "tile_dma or cholesky" > gpurun_out/r06c/test_dma.log 2>&1; tail -5 gpurun_out/r06c/test_dma.log
bash scripts/r06/ab_tile_dma.sh gpurun_out/r06c/ab_tile_dma.jsonl
python scripts/run_c5_video.py --out gpurun_out/r06c/c5_video.json > gpurun_out/r06c/c5.log 2>&1
(cd scripts/ubench; for ch in 1 0; do echo "== CHAIN=$ch"; VGG_CHOL_CHAIN=$ch ./chol_bench 3200 | grep mode=0; VGG_CHOL_CHAIN=$ch ./chol_bench 1202 384 288 | grep mode=0; done) > gpurun_out/r06c/chol_chain.log 2>&1; cat gpurun_out/r06c/chol_chain.log
python - <<PY
import json
for l in open("gpurun_out/r06c/ab_tile_dma.jsonl"):
    d=json.loads(l); k=d["kernel_ms"]; print(d["variant"], d["round"], d["ms_per_iteration"], "off", k["schur_tile<offdiag>"], "diag", k["schur_tile<diag>"], "expand", k["cam_pass<rhs>"], "pp", k["point_pass"], "cost", d["final_cost"])
d=json.load(open("gpurun_out/r06c/c5_video.json")); print(d["final_joint_problem_iteration_ms"], d["final_joint_problem_kernel_ms"])
PY
tail -3 gpurun_out/r06c/ab_tile_dma.jsonl.err
