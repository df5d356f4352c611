mkdir -p gpurun_out/r06d
python -m pytest tests -m gpu -q -x > gpurun_out/r06d/gpu_suite.txt 2>&1; tail -3 gpurun_out/r06d/gpu_suite.txt
python bench.py > gpurun_out/r06d/bench_n1.json 2> gpurun_out/r06d/bench_n1.log; tail -c 300 gpurun_out/r06d/bench_n1.json
bash tools/profile_round.sh r06d main-only > gpurun_out/r06d/profile_round.log 2>&1
cat gpurun_out/r06d/pmc_hbm_bytes.csv | head -20
