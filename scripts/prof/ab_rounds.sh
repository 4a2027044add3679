#!/bin/bash
# Same box, same bench.py: the round-4 library (vggsfm_amd/_variants/lib_r04.so = scripts/prof/build_ref_variant.sh b861de0)
# with the round-4 scheduling of the tile work list and point numbering (environment hooks of ba.build_schur_tiles /
# ba.compile_problem) against HEAD, interleaved.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-strong-leg --no-triangulation --no-pipeline"
emit() { python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print(json.dumps(dict(variant='$1', round=$2, ms_per_step=d['ms_per_step'], ms_per_step_with_events=d['ms_per_step_with_events'], value=d['value'], kernel_ms=d['config']['kernel_ms'])))"; }
for r in 1 2 3; do
  VGGSFM_AMD_LIB=$ROOT/vggsfm_amd/_variants/lib_r04.so VGGSFM_AMD_DEBUG_HOOKS=1 VGGSFM_TILE_ORDER=plain VGGSFM_TILE_FIXED_COST=18,18 VGGSFM_TILE_TOP_UP=0 VGGSFM_SORT_POINTS=0 python bench.py $ARGS 2>/dev/null | emit round4_library_and_schedule $r
  python bench.py $ARGS 2>/dev/null | emit round5_head $r
done
