#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04k
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_ba.py -q -m gpu -x -k "trajectory or exit or reproducible" 2>&1 | tail -4 > $OUT/tests.log
python scripts/prof/ab_c3.py --rounds 1 --steps 40 "prep:" > $OUT/ab.jsonl 2> $OUT/ab.err
python scripts/prof/pipeline_cprofile.py c3 > $OUT/pipeline_cprofile.txt 2>&1
