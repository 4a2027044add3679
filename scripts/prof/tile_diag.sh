#!/bin/bash
# Round 5 diagnosis of the Schur tile launches at HEAD (run on the GPU box: gpurun -- bash scripts/prof/tile_diag.sh):
#   (1) SQ instruction-mix counters of the bench workload (which pipe the non-MFMA 38 % of the launch goes to),
#   (2) the compile-time ablations of schur_tile_kernel (scripts/prof/ablate_tile.sh) + two occupancy builds, all on one box.
# Output: gpurun_out/tile_diag/{counters_*.json, ablations.jsonl}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/tile_diag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $OUT/sq_counter_names.txt
BENCH="python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-strong-leg --no-triangulation --no-pipeline"
pass() {   # name, counters
  rm -rf /tmp/pmc_$1
  (cd $ROOT && timeout 300 rocprofv3 --kernel-trace --pmc $2 --output-format csv -d /tmp/pmc_$1 -- $BENCH > $OUT/pass_$1.log 2>&1)
  python $ROOT/scripts/prof/pmc_aggregate.py $OUT/counters_$1.json /tmp/pmc_$1 > /dev/null 2>&1
  python - <<PY
import json, os
p = "$OUT/counters_$1.json"
if os.path.exists(p):
    d = json.load(open(p))
    for k, v in d.items():
        if "schur_tile" in k or "point_pass" in k or "point_step" in k:
            print(k[:60], {c: round(x["mean"]) for c, x in v.items()})
else:
    print("pass $1 failed:", open("$OUT/pass_$1.log").read()[-400:])
PY
}
pass a "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES"
pass b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"
pass c "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
# ---- ablations (results of these builds are wrong by construction; the product library is restored at the end)
cd $ROOT/vggsfm_amd/csrc
cp ../libvggsfm_amd.so /tmp/lib_product.so
: > $OUT/ablations.jsonl
run() {   # label, extra -D flags
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $2 -c ba.hip -o /tmp/ba_ablate.o 2>/dev/null || { echo "build failed: $1"; return; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libvggsfm_amd.so /tmp/ba_ablate.o $(ls _obj/*.o | grep -v "/ba.o")
  (cd $ROOT && $BENCH 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print(json.dumps(dict(label='$1', flags='$2', ms_per_step=d['ms_per_step'], kernel_ms=d['config']['kernel_ms'], accepted=d['config'].get('successful_steps_last_episode'))))" | tee -a $OUT/ablations.jsonl)
}
run base ""
run base_again ""
run no_mfma "-DVGG_ABLATE=1"
run no_global_loads "-DVGG_ABLATE=2"
run no_barrier "-DVGG_ABLATE=3"
run no_staging_arith "-DVGG_ABLATE=4"
run no_lds_writes "-DVGG_ABLATE=5"
run no_lds_operand_reads "-DVGG_ABLATE=6"
run no_skip "-DVGG_NO_SKIP=1"
VGGSFM_AMD_DEBUG_HOOKS=1 VGGSFM_TILE_WGS=4,4 run offdiag_occ4 "-DVGG_OFFDIAG_OCC=4"
cp /tmp/lib_product.so ../libvggsfm_amd.so
