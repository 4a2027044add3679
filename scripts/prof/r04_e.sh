#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04j
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_ba.py -q -m gpu -x -k "trajectory or tile_rhs or reproducible or launch_variants or camera_split" 2>&1 | tail -6 > $OUT/tests.log
python scripts/prof/ab_c3.py --rounds 2 "interleave:" > $OUT/ab_int.jsonl 2> $OUT/ab_int.err
VGGSFM_AMD_LIB=$ROOT/vggsfm_amd/_variants/lib_nointerleave.so python scripts/prof/ab_c3.py --rounds 2 "nointerleave:" > $OUT/ab_noint.jsonl 2> $OUT/ab_noint.err
python scripts/prof/ab_c3.py --workload c4full --steps 8 --warmup 2 --rounds 1 "c4_interleave:" > $OUT/ab_c4_int.jsonl 2> $OUT/ab_c4_int.err
VGGSFM_AMD_LIB=$ROOT/vggsfm_amd/_variants/lib_nointerleave.so python scripts/prof/ab_c3.py --workload c4full --steps 8 --warmup 2 --rounds 1 "c4_nointerleave:" > $OUT/ab_c4_noint.jsonl 2> $OUT/ab_c4_noint.err
