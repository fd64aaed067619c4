#!/bin/bash
# diagnostic library that records where and when every workgroup of k_cg_wg starts (never the product): elphdynamics_amd/libelphgpu_arrive.so
set -euo pipefail
cd "$(dirname "$0")/.."
O=elphdynamics_amd/build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DELPH_WG_ARRIVE ${ARRIVE_EXTRA:-} -x hip -c elphdynamics_amd/csrc/cg_wg.hip -o $O/cg_wg.hip.arrive.o
OBJS=$(ls $O/*.hip.o $O/cg_fast_mc?_npl?.o $O/*.cpp.o $O/build_info_product.o | grep -v "/cg_wg.hip.o")
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $O/cg_wg.hip.arrive.o -o elphdynamics_amd/libelphgpu_arrive.so
echo built elphdynamics_amd/libelphgpu_arrive.so
