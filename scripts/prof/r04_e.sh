#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04i
mkdir -p $OUT
cd $ROOT
python scripts/prof/tri_ab.py > $OUT/tri_default.jsonl 2> $OUT/tri_default.err
for v in tri_nocontract tri_fallback tri_fallback_nocontract; do
  VGGSFM_AMD_LIB=$ROOT/vggsfm_amd/_variants/lib_$v.so python scripts/prof/tri_ab.py > $OUT/$v.jsonl 2> $OUT/$v.err
done
