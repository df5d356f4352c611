#!/bin/bash
# Kernel-level profile of the config-5 step (50 000 graphs, n=30): kernel stats, step timeline, HBM counter passes.
#   tools/profile_config5.sh r03   -> gpurun_out/r03/config5_{kernel_stats.csv,step_timeline.txt,pmc_hbm_bytes.csv,levels.txt}
set -u
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --workload config5 --steps 3 --warmup 1 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/c5_trace" -- $B > "$out/config5_under_trace.json" 2> "$out/c5_trace.log"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/c5_pmc_$c" -- $B > /dev/null 2> "$out/c5_pmc_$c.log"
done
cd "$root"
cp "$(ls $out/c5_trace/*/*kernel_stats.csv | head -1)" "$out/config5_kernel_stats.csv" 2>/dev/null
python tools/step_timeline.py "$out/c5_trace" > "$out/config5_step_timeline.txt" 2>&1
python tools/pmc_summary.py "$out/c5_pmc_FETCH_SIZE" "$out/c5_pmc_WRITE_SIZE" > "$out/config5_pmc_hbm_bytes.csv"
rm -rf "$out/c5_trace" "$out"/c5_pmc_FETCH_SIZE "$out"/c5_pmc_WRITE_SIZE
$B --opt wl.debug=1 > "$out/config5_50k.json" 2> "$out/config5_levels.txt"
tail -c 400 "$out/config5_50k.json"; tail -40 "$out/config5_step_timeline.txt"; cat "$out/config5_levels.txt" | tail -20
