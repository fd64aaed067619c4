#!/bin/bash
# register / spill / LDS figures of every instantiation of the resident kernels (compile only): tools/wg_resource_usage.sh [file.hip]
cd "$(dirname "$0")/.."
F=${1:-elphdynamics_amd/csrc/cg_wg.hip}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Rpass-analysis=kernel-resource-usage -x hip -c "$F" -o /dev/null 2>&1 |
  python3 -c '
import re, sys
name = None
row = {}
for line in sys.stdin:
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1); row = {}
    for key in ("VGPRs:", "AGPRs:", "VGPR Spill", "SGPR Spill", "Occupancy", "ScratchSize"):
        m2 = re.search(re.escape(key) + r"[^0-9]*([0-9]+)", line)
        if m2 and name: row[key] = m2.group(1)
    if "LDS Size" in line and name:
        print(name, row); name = None
' | while read -r n rest; do echo "$(echo "$n" | c++filt | cut -c1-90) $rest"; done
