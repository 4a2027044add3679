# Round-6 A/B: reduced right-hand side of per-camera-intrinsics problems (7 x 7 / 8 x 8 blocks) from the diagonal tile launch
# (vgg_ba_set_tile_rhs 2, default) against the camera pass cam_pass<RHS> (1 = the round-5 behaviour); same box, interleaved
OUT=${1:-gpurun_out/r06l/ab_tile_rhs_full.jsonl}
mkdir -p $(dirname $OUT); : > $OUT
python scripts/prof/ab_c3.py --workload c4shard --rounds 2 rhs_cam:TILE_RHS=1 rhs_tile:TILE_RHS=2 >> $OUT 2> $OUT.err
python scripts/prof/ab_c3.py --workload c2 --rounds 2 rhs_cam:TILE_RHS=1 rhs_tile:TILE_RHS=2 >> $OUT 2>> $OUT.err
python scripts/prof/ab_c3.py --workload c4full --steps 10 --rounds 2 rhs_cam:TILE_RHS=1 rhs_tile:TILE_RHS=2 >> $OUT 2>> $OUT.err
