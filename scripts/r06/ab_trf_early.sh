# Round-6 A/B: where the full-factor tile_rhs row products sit in a batch of the diagonal tile launch (VGG_TRF_EARLY:
# 0 = behind the batch's last matrix instruction, 1 / 2 = behind K step 0 / 1); variant libraries under vggsfm_amd/_variants
OUT=${1:-gpurun_out/r06m/ab_trf_early.jsonl}
mkdir -p $(dirname $OUT); : > $OUT
for r in 1 2; do
for v in trf0 trf1 default; do
  if [ $v = default ]; then unset VGGSFM_AMD_LIB; else export VGGSFM_AMD_LIB=$PWD/vggsfm_amd/_variants/lib_$v.so; fi
  python scripts/prof/ab_c3.py --workload c4shard --rounds 1 $v:TILE_RHS=2 >> $OUT 2>/dev/null
  python scripts/prof/ab_c3.py --workload c4full --steps 10 --rounds 1 $v:TILE_RHS=2 >> $OUT 2>/dev/null
done; done
unset VGGSFM_AMD_LIB
