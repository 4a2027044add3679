#!/bin/bash
# Profiling only: times schur_tile_kernel without its MFMAs (1) / without its global loads (2) / without the per-batch
# barrier (3) / without the exchange + cross products of the compressed staging (4) / without LDS writes (5) / without LDS
# operand reads (6, off-diagonal launch); NOSKIP=1 adds -DVGG_NO_SKIP=1 (every matrix instruction runs).
# Results of these builds are wrong by construction; the product library is restored at the end.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT/vggsfm_amd/csrc
cp ../libvggsfm_amd.so /tmp/lib_product.so
for A in ${ABLATES:-1 2 4 5 6 0}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DVGG_ABLATE=$A ${EXTRA} -c ba.hip -o /tmp/ba_ablate.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvggsfm_amd.so /tmp/ba_ablate.o $(ls _obj/*.o | grep -v "/ba.o")
  echo "ABLATE=$A ${EXTRA}"
  (cd $ROOT && python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-strong-leg --no-triangulation 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['config']['kernel_ms'], d['config'].get('successful_steps_last_episode'))")
done
cp /tmp/lib_product.so ../libvggsfm_amd.so
