# Round-6 A/B of the off-diagonal Schur tile launch: register staging from compressed records (base) against LDS-DMA staging
# from the expanded image (vgg_ba_set_tile_dma 1 / 2), at three and at four workgroups per CU; lib_dmapipe = the DMA kernel with
# the last K step of a batch behind its barrier.  cam_pass<rhs> in the kernel table = the expansion kernel (DMA variants).
OUT=${1:-gpurun_out/r06c/ab_tile_dma.jsonl}
mkdir -p $(dirname $OUT)
python scripts/prof/ab_c3.py --rounds 2 base: dma1:TILE_DMA=1 "dma1_occ4:TILE_DMA=1,VGGSFM_TILE_WGS=4;4" dma2:TILE_DMA=2 "dma2_occ4:TILE_DMA=2,VGGSFM_TILE_WGS=4;4" > $OUT 2> $OUT.err
VGGSFM_AMD_LIB=$PWD/vggsfm_amd/_variants/lib_dmapipe.so python scripts/prof/ab_c3.py --rounds 2 pipe_base: pipe_dma1:TILE_DMA=1 "pipe_dma1_occ4:TILE_DMA=1,VGGSFM_TILE_WGS=4;4" "pipe_dma2_occ4:TILE_DMA=2,VGGSFM_TILE_WGS=4;4" >> $OUT 2>> $OUT.err
