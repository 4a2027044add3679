mkdir -p gpurun_out/r06d
python -m pytest tests -m gpu -x -q > gpurun_out/r06d/gputest.log 2>&1; tail -4 gpurun_out/r06d/gputest.log
for v in default dfocc1 default dfocc1; do
  if [ $v = default ]; then unset VGGSFM_AMD_LIB; else export VGGSFM_AMD_LIB=$PWD/vggsfm_amd/_variants/lib_dfocc1.so; fi
  python scripts/run_c5_video.py --out gpurun_out/r06d/c5_$v.json > /dev/null 2>&1
  python - <<PY
import json; d=json.load(open("gpurun_out/r06d/c5_$v.json")); print("$v", round(d["final_joint_problem_iteration_ms"],4), {k: round(x,4) for k,x in d["final_joint_problem_kernel_ms"].items()}, "wall", {k: d[k] for k in d if "seconds" in k or "wall" in k})
PY
done
unset VGGSFM_AMD_LIB
