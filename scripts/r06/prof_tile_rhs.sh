# per-kernel times (rocprofv3 --kernel-trace --stats) of one c4 shard iteration with the right-hand side from cam_pass<RHS>
# (VGG_TILE_RHS=1) and from the diagonal tile launch (2)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06n
for m in 1 2; do
  VGG_TILE_RHS=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rhs$m -o p -- python $R/bench.py --workload c4shard --no-cpu-baseline --no-strong-leg --no-pipeline --no-triangulation --steps 10 --warmup 3 > /dev/null 2>&1
  cp $(find /tmp/prof_rhs$m -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r06n/c4shard_tile_rhs${m}_kernel_stats.csv
done
cd $R
python - <<'PY'
import csv
for m in (1, 2):
    rows = list(csv.DictReader(open(f"gpurun_out/r06n/c4shard_tile_rhs{m}_kernel_stats.csv")))
    print("== mode", m)
    for r in rows[:14]:
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.2f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
