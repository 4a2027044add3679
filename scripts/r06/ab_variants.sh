# same-box A/B of library variants (vggsfm_amd/_variants/lib_<name>.so, "default" = the in-tree build) on bench workloads:
#   VARIANTS="default d3" WORKLOADS="c4shard c2" bash scripts/r06/ab_variants.sh out.jsonl
OUT=${1:-gpurun_out/r06o/ab_variants.jsonl}
mkdir -p $(dirname $OUT); : > $OUT
for r in 1 2; do
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then unset VGGSFM_AMD_LIB; else export VGGSFM_AMD_LIB=$PWD/vggsfm_amd/_variants/lib_$v.so; fi
  for wl in ${WORKLOADS:-c4shard}; do
    steps=25; [ $wl = c4full ] && steps=10
    python scripts/prof/ab_c3.py --workload $wl --steps $steps --rounds 1 $v:TILE_RHS=${TILE_RHS:-2} >> $OUT 2>/dev/null
  done
done; done
unset VGGSFM_AMD_LIB
python - $OUT <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); k = d["kernel_ms"]
    print(d["variant"], d.get("workload", ""), d["ms_per_iteration"], "diag", k["schur_tile<diag>"], "off", k["schur_tile<offdiag>"], "lin", k["cam_pass<linearize>"], "pp", k["point_pass"], "chol", k["cholesky"], "camrhs", k["cam_pass<rhs>"], "cost", d["final_cost"])
PY
