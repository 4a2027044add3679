"""torch profiler around the FIRST joint BA call and the FIRST window BA call of the configs[4] loop (one-off costs)."""
import contextlib, importlib.util, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from vggsfm_amd import video as V
seen = {"joint": 0, "window": 0}
def wrap(kind, fn):
    def w(*a, **k):
        seen[kind] += 1
        if seen[kind] == 1:
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                out = fn(*a, **k); torch.cuda.synchronize()
            sys.stderr.write(f"== first {kind} BA\n" + prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=8, max_name_column_width=40) + "\n")
            return out
        return fn(*a, **k)
    return w
V.joint_bundle_adjustment = wrap("joint", V.joint_bundle_adjustment)
V.window_bundle_adjustment = wrap("window", V.window_bundle_adjustment)
spec = importlib.util.spec_from_file_location("run_c5_video", os.path.join(ROOT, "scripts", "run_c5_video.py"))
c5 = importlib.util.module_from_spec(spec); spec.loader.exec_module(c5)
with contextlib.redirect_stdout(io.StringIO()):
    c5.run_video()
