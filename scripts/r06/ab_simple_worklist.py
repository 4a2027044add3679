"""Same-box A/B of the small-problem work list (ba.SIMPLE_WORKLIST_MAX_OBS / SMALL_GRID_CELLS, round 6) on the configs[4] video
loop: the full construction for every window problem against the simple one, interleaved.  One JSON line per run."""
import contextlib, importlib.util, io, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vggsfm_amd import ba as BA  # noqa: E402
spec = importlib.util.spec_from_file_location("run_c5_video", os.path.join(ROOT, "scripts", "run_c5_video.py"))
c5 = importlib.util.module_from_spec(spec)
spec.loader.exec_module(c5)
defaults = (BA.SIMPLE_WORKLIST_MAX_OBS, BA.SMALL_GRID_CELLS)
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for name, vals in (("full work list (round 5)", (0, 0)), ("simple work list below 100 k observations", defaults)):
        BA.SIMPLE_WORKLIST_MAX_OBS, BA.SMALL_GRID_CELLS = vals
        with contextlib.redirect_stdout(io.StringIO()):
            out = c5.run_video()
        print(json.dumps(dict(variant=name, round=rnd, total_seconds=round(out["total_seconds"], 3), window_ba_ms_mean=round(out["window_ba_ms_mean"], 3),
                              window_ba_iterations_mean=round(out["window_ba_iterations_mean"], 2), joint_ba_seconds_total=round(out["joint_ba_seconds_total"], 3),
                              max_rotation_error_rad=out["max_rotation_error_rad"], focal=out["focal"])), flush=True)
