# LM iteration of small problems (video windows) with the multi-launch factorisation below 128 unknowns (VGG_DF_MIN_N=128, rounds
# 3-5) against the dataflow launch at every size (default since round 6): per-iteration wall time and the factorisation's share
OUT=${1:-gpurun_out/r06u/ab_chol_small_n.jsonl}
mkdir -p $(dirname $OUT); : > $OUT
for m in 128 1 128 1; do
  VGG_DF_MIN_N=$m python scripts/r06/small_ba_timing.py 5x1500 9x1500 11x3000 17x3000 17x6000 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(json.dumps(dict(dataflow_from_n=$m, frames=d['frames'], n_reduced=d['n_reduced'], obs=d['obs'], wall_ms_per_iteration=d['wall_ms_per_iteration'], cholesky_ms=d['kernel_ms']['cholesky'])))
" >> $OUT
done
cat $OUT
