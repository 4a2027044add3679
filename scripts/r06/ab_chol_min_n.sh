# configs[4] loop with the dataflow factorisation from n = 128 (rounds 3-5), 65 or 1 unknowns (VGG_DF_MIN_N); one process each
for m in 128 1 65 128 1 65; do
  echo -n "min_n $m "
  VGG_DF_MIN_N=$m python scripts/r06/ab_adjacency_blas.py packed 2>/dev/null | tail -1 | cut -c1-160
done
