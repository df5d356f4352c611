mkdir -p gpurun_out/r05
bash tools/profile_round.sh r05 > gpurun_out/r05/profile_round.log 2>&1
PUB_TIMEOUT=120 PUB_TRACE="wl sp" bash tools/profile_published.sh r05 > gpurun_out/r05/published.log 2>&1
timeout 300 python bench.py --workload config6 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r05/config6_1gpu.json 2> gpurun_out/r05/config6_1gpu.log
tail -c 600 gpurun_out/r05/config6_1gpu.json
for w in nci1 collab; do timeout 200 python bench.py --workload $w --steps 10 --warmup 3 > gpurun_out/r05/bench_$w.json 2> gpurun_out/r05/bench_$w.log; done
python tools/dev/h2h_breakdown.py config3 > gpurun_out/r05/h2h_breakdown.txt 2>&1
ls gpurun_out/r05 | head -80
