set -e
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
timeout -k 10 400 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -1 gpurun_out/bench_full.json
TAG=${1:-r01f}
tools/profile_bench.sh $TAG > gpurun_out/profile_$TAG.log 2>&1
tail -3 gpurun_out/profile_$TAG.log
