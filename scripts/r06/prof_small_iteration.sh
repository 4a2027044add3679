# rocprofv3 kernel stats of the LM iterations of one video-window-sized problem (17 frames x 3000 tracks, n = 102)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_small
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_small -o p -- python $GRAFT_REPO_ROOT/scripts/r06/small_ba_timing.py 17x3000 > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_small/**/*kernel_stats.csv", recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    if "vgg::" in r["Name"]:
        print("%-72s calls %5s avg_us %8.2f" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
