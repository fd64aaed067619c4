#!/bin/bash
# round 4's forms under time-slicing: four processes share the GPU, each solving on a lattice of another register-exchange form
# (GRID L = 12 at 4 slices per wave, GRID L = 10, HGRID 16 x 16 cells, rectangular GRID 8 x 16); every solve must finish in the
# resident kernel (fallbacks 0) with the bits of the first.
mkdir -p gpurun_out/r04
python3 tools/soak_timesliced.py S 40 1200 > gpurun_out/r04/soak_S.log 2>&1 &
python3 tools/soak_timesliced.py q 24 4000 > gpurun_out/r04/soak_q.log 2>&1 &
python3 tools/soak_timesliced.py Y 16 3000 > gpurun_out/r04/soak_Y.log 2>&1 &
python3 tools/soak_timesliced.py R 24 4000 > gpurun_out/r04/soak_R.log 2>&1 &
wait
cat gpurun_out/r04/soak_S.log gpurun_out/r04/soak_q.log gpurun_out/r04/soak_Y.log gpurun_out/r04/soak_R.log
