#!/bin/bash
# tools/time_shard.sh twice: the register-exchange forms on ring-closed slabs (default) and the lane-program form on open slabs
# (ELPH_SHARD_RING=0 ELPH_WG_NO_DPP=1: round 3); prints "cfg ranks us_per_iteration_device" lines
cd "$(dirname "$0")/.."
for mode in ring_grid open_lane_program; do
  if [ $mode = open_lane_program ]; then export ELPH_SHARD_RING=0 ELPH_WG_NO_DPP=1; fi
  bash tools/time_shard.sh 2>/dev/null | python3 -c "
import json, sys
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); r = list(d['spatial'].values())[0]
        print('$mode', r['config'], 'ranks', r['ranks'], '%.2f us per iteration (device)' % r['us_per_iteration_device'], flush=True)
"
done
