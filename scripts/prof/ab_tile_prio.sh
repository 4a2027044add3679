for rep in 1 2; do for v in base prio1 prio2; do
  if [ $v = base ]; then unset VGGSFM_AMD_LIB; else export VGGSFM_AMD_LIB=$PWD/vggsfm_amd/_variants/lib_$v.so; fi
  timeout 100 python bench.py --no-strong-leg --no-cpu-baseline --no-triangulation 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$v', round(d['value'],1), {k: round(x,4) for k,x in d['config']['kernel_ms'].items() if 'schur' in k})"
done; done
