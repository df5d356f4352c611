mkdir -p gpurun_out/r06j
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/r06j/gpu_suite.txt 2>&1; tail -3 gpurun_out/r06j/gpu_suite.txt | head -2; grep -E "passed|failed" gpurun_out/r06j/gpu_suite.txt | tail -1
