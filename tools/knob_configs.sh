#!/bin/bash
# The other configurations under both wave tile shapes (16 x 4: VRT_TILE_LOG2H=2, 32 x 2: =1), kernel time of a launch alone
for k in 2 1 2 1; do
  echo "VRT_TILE_LOG2H=$k"
  VRT_TILE_LOG2H=$k timeout -k 10 300 python -u tools/bench_configs.py 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('  config', d['config'], {k:(round(v*1e3,2) if isinstance(v,float) and 'ms' in k else v) for k,v in d.items() if 'ms' in k and not isinstance(v,dict)}, d.get('frames_in_flight_ms_per_frame') or d.get('two_frames_in_flight'))"
done
