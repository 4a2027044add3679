"""A/B on one box: off-diagonal Schur tile kernel at 3 / 4 workgroups per CU (variant library built with
-DVGG_OFFDIAG_OCC=4 as vggsfm_amd/_variants/lib_occ4.so).  usage: python scripts/prof/ab_tile_occ.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for name, lib, wgs in (("occ3", None, "3,4"), ("occ4", "vggsfm_amd/_variants/lib_occ4.so", "4,4"), ("occ3", None, "3,4"),
                       ("occ4", "vggsfm_amd/_variants/lib_occ4.so", "4,4")):
    env = dict(os.environ, VGGSFM_TILE_WGS=wgs)
    if lib:
        env["VGGSFM_AMD_LIB"] = os.path.join(ROOT, lib)
    code = ("import os, sys, runpy; sys.argv = ['bench.py', '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-strong-leg', '--no-triangulation'];"
            "import vggsfm_amd.ba as BA; BA.TILE_WGS_PER_CU = tuple(int(v) for v in os.environ['VGGSFM_TILE_WGS'].split(','));"
            "runpy.run_path('bench.py', run_name='__main__')")
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    print(name, round(d["value"], 1), {k: round(v, 4) for k, v in d["config"]["kernel_ms"].items()}, flush=True)
