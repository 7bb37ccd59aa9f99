# A/B of two builds of the library on one box: alternating bench runs (kernel time by HIP events and step time).
# usage: bash tools/ab_libs.sh raytracing_amd/ab_prev.so [rounds] [further bench.py arguments]
set -u
other=$1; rounds=${2:-3}; shift; shift || true
for i in $(seq $rounds); do
  for lib in "$other" raytracing_amd/libvrt.so; do
    VRT_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 400 "$@" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$lib', 'step %.2f us' % (d['ms_per_step']*1e3), 'kernel %.2f us' % (d['roofline']['kernel_ms']*1e3))" || exit 1
  done
done
