for i in 1 2; do
for cfg in "2 2" "1 2" "3 2" "2 3" "2 4" "2 1"; do set -- $cfg
VRT_TILE_LOG2H=$1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-literal --steps 400 --overlap $2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('log2h $1 overlap $2', 'step %.2f us' % (d['ms_per_step']*1e3), 'kernel %.2f us' % (d['roofline']['kernel_ms']*1e3))"
done; done
