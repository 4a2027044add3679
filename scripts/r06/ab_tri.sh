# Round-6 A/B of triangulate_kernel: view-loop pipeline (VGG_TRI_PIPE) and estimate-based reciprocals in the eigen solve
# (VGG_TRI_FASTDIV); variant libraries under vggsfm_amd/_variants (see the commands in profiles/r06_ab_tri_c3.jsonl)
OUT=${1:-gpurun_out/r06e/ab_tri.jsonl}
mkdir -p $(dirname $OUT); : > $OUT
for r in 1 2; do
for v in base pipe fdiv default; do
  if [ $v = default ]; then unset VGGSFM_AMD_LIB; else export VGGSFM_AMD_LIB=$PWD/vggsfm_amd/_variants/lib_tri_$v.so; fi
  python scripts/prof/tri_ab.py c2 c3 >> $OUT 2>/dev/null
done; done
unset VGGSFM_AMD_LIB
