set -u
mkdir -p gpurun_out && rm -rf gpurun_out/prof gpurun_out/kt gpurun_out/kt_lit
export TMPDIR=/tmp
timeout -k 10 400 python bench.py > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err || exit 1
echo bench done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/kt.log 2>&1 || exit 1
echo kt done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_lit -- python3 bench.py --steps 20 --warmup 2 --no-cpu-baseline --mode literal > gpurun_out/kt_lit.log 2>&1 || exit 1
echo kt literal done
bash tools/profile_pmc.sh --steps 8 || exit 1
echo pmc done
timeout -k 10 500 python tools/bench_configs.py 1 2 3 4 5 > gpurun_out/configs.json 2> gpurun_out/configs.err || exit 1
echo configs done
