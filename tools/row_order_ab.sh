#!/bin/bash
# Launches one at a time (--overlap 1): the measured dispatch order of workgroup rows against rows top to bottom
for i in 1 2 3; do for k in 0 1; do
VRT_NO_ROW_ORDER=$k timeout -k 10 200 python bench.py --no-cpu-baseline --no-literal --steps 400 --overlap 1 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('VRT_NO_ROW_ORDER=$k', 'step %.2f us' % (d['ms_per_step']*1e3), 'kernel %.2f us' % (d['roofline']['kernel_ms']*1e3))"
done; done
