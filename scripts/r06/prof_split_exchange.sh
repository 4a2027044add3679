cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_split
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_split -o p -- python $R/scripts/prof/ab_c3.py --workload c4shard --rounds 1 --steps 10 split:RCCL=3 > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_split/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.2f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}")
PY
