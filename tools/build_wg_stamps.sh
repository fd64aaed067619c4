#!/bin/bash
# diagnostic library with per-phase stamps in k_cg_wg (never the product): elphdynamics_amd/libelphgpu_stamps.so
# usage: tools/build_wg_stamps.sh [wave whose stamps are reported, default 0]
set -euo pipefail
cd "$(dirname "$0")/.."
O=elphdynamics_amd/build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DELPH_WG_STAMPS -DELPH_WG_STAMP_WAVE=${1:-0} -x hip -c elphdynamics_amd/csrc/cg_wg.hip -o $O/cg_wg.hip.stamps.o
OBJS=$(ls $O/*.hip.o $O/cg_fast_mc?_npl?.o $O/*.cpp.o $O/build_info_product.o | grep -v "/cg_wg.hip.o")
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $O/cg_wg.hip.stamps.o -o elphdynamics_amd/libelphgpu_stamps.so
echo built elphdynamics_amd/libelphgpu_stamps.so
