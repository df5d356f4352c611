#!/bin/bash
# Round 5, after the ShortestPath rework (counter rows, breadth-first search, float64 Gram): the full GPU suite, then the SP
# legs of the four published-like sets (JSON + kernel stats) and the config-4 stand-in.  Run from the repo root on the GPU box.
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r05sp
mkdir -p "$out"
timeout 260 python -m pytest tests -m gpu -q --timeout 150 2>&1 | grep -E "passed|failed|^FAILED|^E  |Timeout" > "$out/gpu_suite.txt"
cat "$out/gpu_suite.txt"
cd /tmp && export TMPDIR=/tmp
for s in nci1 dd reddit collab; do
  timeout 100 python $root/tools/published_like.py $s sp 3 > "$out/pub_${s}_sp.json" 2> "$out/pub_${s}_sp.log"
  timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/pt_$s" -- python $root/tools/published_like.py $s sp 3 > /dev/null 2> "$out/pt_$s.log"
  cp "$(ls $out/pt_$s/*/*kernel_stats.csv 2>/dev/null | head -1)" "$out/pub_${s}_sp_kernel_stats.csv" 2>/dev/null
  rm -rf "$out/pt_$s"
  python $root/tools/kstats.py "$out/pub_${s}_sp_kernel_stats.csv" 30 > "$out/pub_${s}_sp_kernel_stats.txt" 2>/dev/null
  tail -c 300 "$out/pub_${s}_sp.json"; echo
done
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/sp_trace" -- python $root/tools/bench_sp.py 4110 5 > "$out/sp_config4_bench.json" 2> "$out/sp_trace.log"
cp "$(ls $out/sp_trace/*/*kernel_stats.csv 2>/dev/null | head -1)" "$out/sp_config4_kernel_stats.csv" 2>/dev/null
rm -rf "$out/sp_trace"
tail -c 400 "$out/sp_config4_bench.json"
