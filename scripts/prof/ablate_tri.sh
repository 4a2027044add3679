#!/bin/bash
# Profiling only: phase shares of the LO-RANSAC kernel (results of the ablated builds are wrong by construction).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT/vggsfm_amd/csrc
for A in 1 2 3 0; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=off -DVGG_TRI_ABLATE=$A -c triangulate.hip -o _obj/triangulate.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvggsfm_amd.so _obj/*.o
  echo "TRI_ABLATE=$A"; (cd $ROOT && python scripts/prof/tri_angle_cost.py 2>/dev/null | tail -2 | tr "\n" " "; echo)
done
