#!/bin/bash
# diagnostic library whose k_cg_wg iterates WITHOUT its meeting (-DELPH_WG_NOMEET; the numbers it computes mean nothing, never the
# product): elphdynamics_amd/libelphgpu_nomeet.so — the bound tools/time_wg_nomeet.py measures
set -euo pipefail
cd "$(dirname "$0")/.."
O=elphdynamics_amd/build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DELPH_WG_NOMEET -x hip -c elphdynamics_amd/csrc/cg_wg.hip -o $O/cg_wg.hip.nomeet.o
OBJS=$(ls $O/*.hip.o $O/cg_fast_mc?_npl?.o $O/*.cpp.o $O/build_info_product.o | grep -v "/cg_wg.hip.o")
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $O/cg_wg.hip.nomeet.o -o elphdynamics_amd/libelphgpu_nomeet.so
echo built elphdynamics_amd/libelphgpu_nomeet.so
