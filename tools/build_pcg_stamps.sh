#!/bin/bash
# diagnostic library with per-phase stamps in k_pcg_wg (never the product): elphdynamics_amd/libelphgpu_pcgstamps.so
set -euo pipefail
cd "$(dirname "$0")/.."
O=elphdynamics_amd/build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DELPH_PCG_STAMPS -x hip -c elphdynamics_amd/csrc/pcg_wg.hip -o $O/pcg_wg.hip.stamps.o
OBJS=$(ls $O/*.hip.o $O/cg_fast_mc?_npl?.o $O/*.cpp.o $O/build_info_product.o | grep -v "pcg_wg.hip.o")
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $O/pcg_wg.hip.stamps.o -o elphdynamics_amd/libelphgpu_pcgstamps.so
echo built elphdynamics_amd/libelphgpu_pcgstamps.so
