set -e
cd /root/repo
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 180 --timeout-method thread 2>&1 | tail -5
timeout -k 10 300 python bench.py --no-cpu-baseline 2>gpurun_out/bench_err.log | tee gpurun_out/bench_quick.json
