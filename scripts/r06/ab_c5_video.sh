# configs[4] video loop under the previous-commit library, the in-tree one, and with the simple work list of small problems off
mkdir -p gpurun_out/r06p
for v in head default; do
  if [ $v = default ]; then unset VGGSFM_AMD_LIB; else export VGGSFM_AMD_LIB=$PWD/vggsfm_amd/_variants/lib_$v.so; fi
  python scripts/run_c5_video.py --out gpurun_out/r06p/c5_$v.json > gpurun_out/r06p/c5_$v.log 2>&1
  python - <<PY
import json
d=json.load(open("gpurun_out/r06p/c5_$v.json"))
print("$v", d["total_seconds"], d["window_ba_ms_mean"], d["window_ba_iterations_mean"], d["final_joint_ba"]["iterations"], [j["iterations"] for j in d["joint_ba_log"]])
PY
done
unset VGGSFM_AMD_LIB
python -c "import sys; sys.argv=['x','--out','gpurun_out/r06p/c5_nosimple.json']; sys.path.insert(0,'scripts'); sys.path.insert(0,'.'); import vggsfm_amd.ba as BA; BA.SIMPLE_WORKLIST_MAX_OBS=0; import run_c5_video as R; R.main()" > gpurun_out/r06p/c5_nosimple.log 2>&1
python - <<PY
import json
d=json.load(open("gpurun_out/r06p/c5_nosimple.json"))
print("nosimple?", d["total_seconds"], d["window_ba_ms_mean"], d["window_ba_iterations_mean"], d["final_joint_ba"]["iterations"], [j["iterations"] for j in d["joint_ba_log"]])
PY
