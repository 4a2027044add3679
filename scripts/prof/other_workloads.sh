# c2 / c4 shard / whole c4 on one GPU with the in-tree library; the traced factorisation at n = 1202 (camera split of c3)
for wl in c2 c4shard c4full; do
  python bench.py --workload $wl --no-cpu-baseline --no-strong-leg --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/wl_$wl.json
done
scripts/ubench/chol_bench_trace 1202 384 288 > gpurun_out/chol_trace_n1202.txt 2>&1
