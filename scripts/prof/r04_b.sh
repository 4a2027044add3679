#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04b
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_pycolmap_compat.py tests/test_gpu_triangulator_golden.py tests/test_gpu_runner.py -q -m gpu 2>&1 | tail -8 > $OUT/tests.log
python scripts/prof/ab_c3.py --rounds 2 "base:VGGSFM_TILE_BACKFILL=0" "bf3_3:VGGSFM_TILE_BACKFILL=3;3" "bf3_4:VGGSFM_TILE_BACKFILL=3;4" "bf3_6:VGGSFM_TILE_BACKFILL=3;6" "bf3_2:VGGSFM_TILE_BACKFILL=3;2" "bf2.9_4:VGGSFM_TILE_BACKFILL=2.9;4" > $OUT/ab_backfill.jsonl 2> $OUT/ab_backfill.err
python bench.py --no-cpu-baseline --no-strong-leg --no-triangulation > $OUT/bench_pipeline.json 2> $OUT/bench_pipeline.err
