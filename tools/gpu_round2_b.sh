set -x
mkdir -p gpurun_out/r2b
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r2b/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -15 gpurun_out/r2b/gpu_tests.log
