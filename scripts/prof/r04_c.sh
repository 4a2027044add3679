#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04d
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_ba.py tests/test_gpu_dist.py tests/test_gpu_video.py tests/test_gpu_pycolmap_compat.py tests/test_gpu_video_sequence.py tests/test_gpu_triangulator_golden.py -q -m gpu 2>&1 | tail -15 > $OUT/tests.log
python scripts/prof/ab_c3.py --rounds 2 "rhs_cam:TILE_RHS=0" "rhs_tile:TILE_RHS=1" > $OUT/ab_tile_rhs.jsonl 2> $OUT/ab_tile_rhs.err
VGGSFM_AMD_LIB=$ROOT/vggsfm_amd/_variants/lib_diag_occ3.so python scripts/prof/ab_c3.py --rounds 1 "occ3_rhs_cam:TILE_RHS=0" "occ3_rhs_tile:TILE_RHS=1" > $OUT/ab_tile_rhs_occ3.jsonl 2> $OUT/ab_tile_rhs_occ3.err
