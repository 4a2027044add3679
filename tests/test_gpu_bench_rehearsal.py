"""the driver's multi-GPU command line, rehearsed end to end before its first real run.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 2 --steps K --warmup W

Only one GPU is visible on the test box, so the two ranks SHARE it and the collectives go over gloo, staged through the host
(``VGGSFM_BENCH_BACKEND=gloo`` -- bench.py's rehearsal switch; the measured path is RCCL).  Everything else is the code the
8-GPU run executes: RANK / LOCAL_RANK / WORLD_SIZE handling, the agreed camera order, the sharded solve with its three
exchanges per iteration (camera blocks; reduce-scatter + all-gather of the packed reduced system on the solver's own buffers,
phases 4 / 6; step scalars), barrier + max-over-ranks timing, rank 0's solo run of the whole configs[3] problem and the
strong-scaling leg on it, the ONE JSON line on rank 0.

Round 5: `value` means the same thing at every N -- configs[2] shard-iterations per second, one
100k-track shard per GPU -- so value(N) / value(1) is a scaling curve; the strong-scaling figure on configs[3] rides along as
the key `strong_scaling_c4` with its own N = 1 point.  Asserted here for the N = 2 line against the N = 1 line's fields."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_command_line_over_gloo():
    env = dict(os.environ, VGGSFM_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--strong-steps", "3"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                         # ONE JSON line, printed by rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["unit"].startswith("shard-iterations/s")
    assert out["scaling"] == "weak" and out["higher_is_better"] is True and out["dtype"] == "f64"
    # whole-job aggregate: N x K / t shard-iterations per second
    assert out["value"] > 0 and abs(out["value"] - 2 * 1e3 / out["ms_per_step"]) <= 1e-6 * out["value"]
    cfg = out["config"]
    # the SAME workload family as the N = 1 line (bench.py's default there: configs[2], 200 x 100k per GPU, n = 1202)
    assert cfg["frames"] == 200 and cfg["tracks_per_gpu"] == 100000 and cfg["reduced_system"] == 1202
    assert "configs[2]" in cfg["workload"] and "shared_camera" in cfg["workload"]
    strong = out["strong_scaling_c4"]
    assert strong["scaling"] == "strong" and strong["tracks_per_rank"] == 150000 and strong["reduced_system"] == 3200
    n1 = strong["n1_same_problem"]
    assert n1["observations"] > 2 * strong["observations_rank0"] * 0.9 and n1["lm_iterations_per_s"] > 0
    assert strong["speedup_vs_n1"] == pytest.approx(strong["lm_iterations_per_s"] / n1["lm_iterations_per_s"])
    # the two scaling figures side by side at the head of the line, where a reader of SCALE_rNN.json cannot miss either
    head = lines[0][:700]
    assert '"value_definition": "weak: shard-iterations/s' in head and '"strong_scaling_c4_speedup_vs_n1"' in head
    assert out["strong_scaling_c4_speedup_vs_n1"] == strong["speedup_vs_n1"]
    assert out["roofline"]["bound"] in ("mfma", "hbm") and 0 < out["roofline"]["frac"] < 1
    assert out["cpu_baseline"] is None                                # rank 0 at N = 1 only
    assert out["pose_delta_vs_port_c4"] is None and out["pose_delta_vs_port_c5_joint"] is None
