#!/bin/bash
# HBM-side traffic per kernel launch of the bench workload (run on the GPU box via gpurun).
# Separate passes per counter, kernel-trace only (no sys/hip/hsa trace together with --pmc).
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  (cd $ROOT && rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$C -- \
      python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-strong-leg --no-triangulation --no-pipeline > $OUT/bench_$C.log 2>&1)
done
# calibration: known-size streaming reads with the kernels' access widths
make -s -C $ROOT/scripts/ubench fetch_calib >/dev/null 2>&1
if [ -x $ROOT/scripts/ubench/fetch_calib ]; then
  rm -rf /tmp/pmc_calib
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_calib -- $ROOT/scripts/ubench/fetch_calib > $OUT/calib.log 2>&1
  python $ROOT/scripts/prof/pmc_aggregate.py $OUT/pmc_calib.json /tmp/pmc_calib
fi
python $ROOT/scripts/prof/pmc_aggregate.py $OUT/pmc_summary.json /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE
