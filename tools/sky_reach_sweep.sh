#!/bin/bash
# How steep a descending ray is still tried for the sky short cut (VRT_SKY_REACH), bench frame, two rounds
for i in 1 2; do for r in 100 1.8 0.9 0.6 0.45; do
VRT_SKY_REACH=$r timeout -k 10 200 python bench.py --no-cpu-baseline --no-literal --steps 400 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('reach $r', 'step %.2f us' % (d['ms_per_step']*1e3), 'kernel %.2f us' % (d['roofline']['kernel_ms']*1e3), d['parity'])"
done; done
