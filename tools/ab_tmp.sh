set -e
timeout -k 10 600 python -u -m pytest tests/test_engine_gpu.py tests/test_dropin_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3
