"""configs[4] loop with different thresholds of the simple work list (ba.SIMPLE_WORKLIST_MAX_OBS), one process, interleaved."""
import contextlib, importlib.util, io, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vggsfm_amd import ba as BA
spec = importlib.util.spec_from_file_location("run_c5_video", os.path.join(ROOT, "scripts", "run_c5_video.py"))
c5 = importlib.util.module_from_spec(spec); spec.loader.exec_module(c5)
with contextlib.redirect_stdout(io.StringIO()):
    c5.run_video()                                   # warm-up (code-object loads)
for rnd in range(2):
    for lim in (100_000, 200_000, 400_000, 1_000_000):
        BA.SIMPLE_WORKLIST_MAX_OBS = lim
        with contextlib.redirect_stdout(io.StringIO()):
            out = c5.run_video()
        print(json.dumps(dict(simple_below=lim, round=rnd, total_seconds=round(out["total_seconds"], 3), window_ba_ms_mean=round(out["window_ba_ms_mean"], 2),
                              window_ba_iterations_mean=round(out["window_ba_iterations_mean"], 2), joint_ba_seconds_total=round(out["joint_ba_seconds_total"], 3))), flush=True)
