# Round-6 A/B, second formulation: producer / consumer wave specialisation of the off-diagonal tile launch (vgg_ba_set_tile_dma 3 / 4)
OUT=${1:-gpurun_out/r06g/ab_tile_pc.jsonl}
mkdir -p $(dirname $OUT)
python scripts/prof/ab_c3.py --rounds 2 base: pc3:TILE_DMA=3 pc4:TILE_DMA=4 "dma1_occ4:TILE_DMA=1,VGGSFM_TILE_WGS=4;4" > $OUT 2> $OUT.err
