for r in 1 2; do for v in head default; do
  if [ $v = default ]; then unset VGGSFM_AMD_LIB; else export VGGSFM_AMD_LIB=$PWD/vggsfm_amd/_variants/lib_$v.so; fi
  echo "== $v"; python scripts/r06/window_ba_breakdown.py 2>/dev/null | grep '"rep": [23]'
done; done
unset VGGSFM_AMD_LIB
