set -e
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 400 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -1 gpurun_out/bench_full.json
tools/profile_bench.sh r01e > gpurun_out/profile_r01e.log 2>&1
tail -3 gpurun_out/profile_r01e.log
