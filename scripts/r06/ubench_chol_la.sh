cd scripts/ubench
for b in factor64_bench_la0_sq0 factor64_bench_la1_sq0 factor64_bench_la1_sq1 factor64_bench_la2_sq1 factor64_bench_la4_sq1; do echo "== $b"; timeout 60 ./$b 1; done
for b in chol_bench_la0_sq0 chol_bench_la1_sq1 chol_bench_la2_sq1; do for args in "1202 384 288" "3200" "6002"; do echo "== $b $args"; timeout 60 ./$b $args | grep -v "mode=1"; done; done
echo "== trace la0"; timeout 60 ./chol_trace_la0_sq0 1202 384 288
echo "== trace la2"; timeout 60 ./chol_trace_la2_sq1 1202 384 288
