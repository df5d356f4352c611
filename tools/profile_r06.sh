#!/bin/bash
# Everything profiles/r06_* comes from, in one call on the GPU box (run from the repo root):  bash tools/profile_r06.sh
mkdir -p gpurun_out/r06
timeout 900 bash tools/profile_round.sh r06 > gpurun_out/r06/profile_round.log 2>&1
PUB_TIMEOUT=120 PUB_TRACE="wl sp" timeout 1500 bash tools/profile_published.sh r06 > gpurun_out/r06/published.log 2>&1
timeout 300 python bench.py --workload config6 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r06/config6_1gpu.json 2> gpurun_out/r06/config6_1gpu.log
tail -c 600 gpurun_out/r06/config6_1gpu.json
for w in nci1 collab dd reddit; do timeout 300 python bench.py --workload $w --steps 10 --warmup 3 > gpurun_out/r06/bench_$w.json 2> gpurun_out/r06/bench_$w.log; done
timeout 200 python tools/dev/h2h_breakdown.py config3 > gpurun_out/r06/h2h_breakdown.txt 2>&1
ls gpurun_out/r06 | head -100
