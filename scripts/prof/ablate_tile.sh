#!/bin/bash
# Profiling only: times schur_tile_kernel without its MFMAs (1) / without its global loads (2).
# Results of these builds are wrong by construction; the product library is rebuilt at the end.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT/vggsfm_amd/csrc
for A in ${ABLATES:-1 2 0}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DVGG_ABLATE=$A -c ba.hip -o _obj/ba.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvggsfm_amd.so _obj/*.o
  echo "ABLATE=$A"
  (cd $ROOT && python bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['config']['kernel_ms'], d['config'].get('successful_steps_last_episode'))")
done
