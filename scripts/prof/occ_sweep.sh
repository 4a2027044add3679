#!/bin/bash
# Profiling only: occupancy (min waves per SIMD) sweep of the observation-pass kernels.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT/vggsfm_amd/csrc
for O in "2 2 2" "3 3 3" "4 4 4"; do
  set -- $O
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DVGG_PP_OCC=$1 -DVGG_PS_OCC=$2 -DVGG_CP_OCC=$3 -c ba.hip -o _obj/ba.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvggsfm_amd.so _obj/*.o
  echo "OCC=$O"
  (cd $ROOT && python bench.py --steps 6 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['config']['kernel_ms'])")
done
