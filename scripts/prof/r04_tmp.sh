#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04o
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_triangulation.py tests/test_gpu_triangulator_golden.py tests/test_gpu_dist.py -q -m gpu 2>&1 | tail -4 > $OUT/tests.log
python scripts/prof/tri_ab.py > $OUT/tri_ordered.jsonl 2> $OUT/tri_ordered.err
VGGSFM_TRI_ORDER=0 python scripts/prof/tri_ab.py > $OUT/tri_unordered.jsonl 2> $OUT/tri_unordered.err
