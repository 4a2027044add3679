"""Where the one-off 45 / 170 ms of the first two joint BAs of the configs[4] loop go: compile_problem / solve timed with
synchronisation inside every bundle_adjustment call of scripts/run_c5_video.py."""
import contextlib, importlib.util, io, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vggsfm_amd import ba as BA
rows = []
cp0, sv0 = BA.compile_problem, BA.solve
def cp(*a, **k):
    if int(a[1].shape[0]) == 139 and os.environ.get("PROFILE_SECOND"):
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            out = cp0(*a, **k); torch.cuda.synchronize()
        sys.stderr.write(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=12) + "\n")
        return out
    torch.cuda.synchronize(); t = time.perf_counter(); out = cp0(*a, **k); torch.cuda.synchronize()
    rows.append(dict(what="compile", ms=round(1e3 * (time.perf_counter() - t), 2), obs=int(out[0].num_obs), frames=int(a[1].shape[0]))); return out
def sv(*a, **k):
    torch.cuda.synchronize(); t = time.perf_counter(); out = sv0(*a, **k); torch.cuda.synchronize()
    rows.append(dict(what="solve", ms=round(1e3 * (time.perf_counter() - t), 2), iterations=out[0]["num_iterations"], n=out[0]["n_reduced"])); return out
BA.compile_problem, BA.solve = cp, sv
spec = importlib.util.spec_from_file_location("run_c5_video", os.path.join(ROOT, "scripts", "run_c5_video.py"))
c5 = importlib.util.module_from_spec(spec); spec.loader.exec_module(c5)
with contextlib.redirect_stdout(io.StringIO()):
    out = c5.run_video()
big = [r for r in rows if r.get("frames", 0) > 40 or r.get("n", 0) > 250]
for r in big[:16]: print(json.dumps(r))
print(json.dumps(dict(total_seconds=out["total_seconds"], compile_ms_total=sum(r["ms"] for r in rows if r["what"] == "compile"), solve_ms_total=sum(r["ms"] for r in rows if r["what"] == "solve"), calls=len(rows) // 2)))
