# A/B of the chained diagonal (VGG_CHOL_CHAIN) in the dataflow factorisation: chol_bench n [split_a split_b]
cd "$(dirname "$0")/../ubench"
for args in ${CHOL_AB_CASES:-"1202 384 288" "3200" "4096" "4800" "6002"}; do
  for ch in 1 0; do
    echo "== VGG_CHOL_CHAIN=$ch: $args"; VGG_CHOL_CHAIN=$ch timeout 60 ./chol_bench $args | grep "mode=0"
  done
done
