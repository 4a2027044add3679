mkdir -p gpurun_out/r06f
python -m pytest tests/test_gpu_ba.py tests/test_gpu_video.py tests/test_gpu_video_golden.py tests/test_gpu_video_sequence.py tests/test_gpu_triangulator_golden.py -x -q 2>&1 | tail -4
python scripts/run_c5_video.py --out gpurun_out/r06f/c5_video.json > /dev/null 2>&1
python - <<PY
import json; d=json.load(open("gpurun_out/r06f/c5_video.json")); print({k: d[k] for k in d if "seconds" in k or "window" in k or "ms_mean" in k})
PY
