timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_sliced_launch.py -x -q --timeout 500 --timeout-method thread 2>&1 | tail -n 3 || exit 1
timeout -k 10 120 python3 tools/prof_run.py --spp 1024 --launches 3 --kernel 5 || exit 1
timeout -k 10 120 python3 tools/prof_run.py --spp 1024 --launches 2 --kernel 5 --mode stream || exit 1
