# which part of the full-factor tile_rhs path costs the diagonal tile launch its time: rocprofv3 kernel stats of one configs[3]
# shard, library variants built with -DVGG_TRF_ABLATE=1|2|3 (no row products | no z loads | neither), tile_rhs modes 1 and 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in ${VARIANTS:-default trfabl1 trfabl2 trfabl3}; do
for m in 1 2; do
  if [ $v = default ]; then unset VGGSFM_AMD_LIB; else export VGGSFM_AMD_LIB=$R/vggsfm_amd/_variants/lib_$v.so; fi
  rm -rf /tmp/prof_$v
  VGG_TILE_RHS=$m rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -o p -- python $R/bench.py --workload c4shard --no-cpu-baseline --no-strong-leg --no-pipeline --no-triangulation --steps 10 --warmup 3 > /dev/null 2>&1
  echo "== $v mode $m"; grep -E "schur_tile_kernel<8, true>" $(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1) | cut -d, -f6-9
done; done
