"""VGPRs / scratch / occupancy / LDS of every kernel of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python scripts/prof/kernel_resources.py vggsfm_amd/csrc/ba.hip [name filter] [extra hipcc flags...]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-c", src, "-o",
       "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur).replace("void vgg::", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" ")[0]] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print(f"{k:70s} vgpr {v.get('VGPRs', 0):4d} agpr {v.get('AGPRs', 0):4d} scratch {v.get('ScratchSize', 0):4d} occ {v.get('Occupancy', 0)} lds {v.get('LDS', 0)}")
