#!/bin/bash
# Registers / spills / occupancy of every kernel of libtfx_hip.so, as the compiler reports them.
# usage: tools/kernel_resources.sh [name filter]
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off \
  -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -fno-fast-math -Iinclude \
  --cuda-device-only -S -o /dev/null traffic-env_amd/csrc/tfx_hip.hip -Rpass-analysis=kernel-resource-usage 2>&1 |
python3 -c '
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    m = re.search(r"remark: (?:\S+ )?\s*Function Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"\s+(VGPRs|AGPRs|SGPRs|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|ScratchSize \[bytes/lane\]): (\d+)", line)
    if m and cur is not None: cur[m.group(1)] = int(m.group(2))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
print("%-64s %5s %5s %6s %6s %4s %6s" % ("kernel", "VGPR", "SGPR", "vspill", "sspill", "occ", "LDS"))
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("tfx::", "")
    if flt in n:
        print("%-64s %5d %5d %6d %6d %4d %6d" % (n[:64], r.get("VGPRs", -1), r.get("SGPRs", -1), r.get("VGPRs Spill", -1),
              r.get("SGPRs Spill", -1), r.get("Occupancy [waves/SIMD]", -1), r.get("LDS Size [bytes/block]", -1)))
' "$1"
