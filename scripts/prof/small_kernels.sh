# per-kernel averages of the c3 bench under rocprofv3 (kernel trace only), for the library in $VGGSFM_AMD_LIB (or the in-tree one)
cd /tmp && export TMPDIR=/tmp
tag=${1:-cur}
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-strong-leg --steps 20 > /tmp/prof_$tag.log 2>&1
python - <<PY
import csv, glob
f = glob.glob('/tmp/prof_$tag/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = 0
for r in rows:
    n = r['Name']
    if 'vgg::' in n or 'fillBuffer' in n:
        print('$tag', n[:60].ljust(60), r['Calls'].rjust(5), round(float(r['AverageNs'])/1e3, 2))
PY
