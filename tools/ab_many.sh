# A/B of several builds of the library on one box: alternating bench runs; prints kernel and step time per build and round.
# usage: bash tools/ab_many.sh rounds lib1.so lib2.so ... [-- further bench.py arguments]
set -u
rounds=$1; shift
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ $# -gt 0 ] && shift
for i in $(seq $rounds); do
  for lib in "${libs[@]}"; do
    VRT_LIB=$PWD/$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-literal --steps 400 "$@" 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$lib', 'step %.2f us' % (d['ms_per_step']*1e3), 'kernel %.2f us' % (d['roofline']['kernel_ms']*1e3), 'parity', d.get('parity',{}).get('bgra8_sha256_equals_oracle_golden'))" || exit 1
  done
done
