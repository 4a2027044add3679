# needs vggsfm_amd/_variants/lib_head.so: scripts/prof/build_ref_variant.sh <git-ref of the build to compare with>
for i in 1 2; do
  for v in new head nochain; do
    if [ $v = head ]; then export VGGSFM_AMD_LIB=$PWD/vggsfm_amd/_variants/lib_head.so; else unset VGGSFM_AMD_LIB; fi
    if [ $v = nochain ]; then export VGG_CHOL_CHAIN=0; else unset VGG_CHOL_CHAIN; fi
    python bench.py --no-cpu-baseline --no-strong-leg --steps 40 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['config']['kernel_ms']['cholesky'],4))"
  done
done
