#!/bin/bash
# Collect the round's profiles of the config-3 bench on the GPU box (run from the repo root):
#   tools/profile_round.sh r04        -> gpurun_out/r04/{kernel_stats.csv, step_timeline.txt, pmc_*.csv, bench_n1.json}
# Kernel trace and every counter group are separate rocprofv3 runs (--pmc is never combined with other traces).
set -u
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $B > "$out/bench_under_trace.json" 2> "$out/trace.log"
pmc() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$out/pmc_$name" -- $B > /dev/null 2> "$out/pmc_$name.log"; }
pmc FETCH_SIZE FETCH_SIZE
pmc WRITE_SIZE WRITE_SIZE
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
pmc sq SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU
pmc mem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS
cd "$root"
cp "$(ls $out/trace/*/*kernel_stats.csv | head -1)" "$out/kernel_stats.csv" 2>/dev/null
python tools/step_timeline.py "$out/trace" > "$out/step_timeline.txt" 2>&1
python tools/pmc_summary.py "$out/pmc_FETCH_SIZE" "$out/pmc_WRITE_SIZE" > "$out/pmc_hbm_bytes.csv"
python tools/pmc_summary.py "$out/pmc_mfma" > "$out/pmc_mfma_util.csv"
python tools/pmc_summary.py "$out/pmc_sq" "$out/pmc_mem" > "$out/pmc_sq_waves.csv"
# the trace/counter dumps themselves are large: keep the summaries only
rm -rf "$out/trace" "$out"/pmc_FETCH_SIZE "$out"/pmc_WRITE_SIZE "$out"/pmc_mfma "$out"/pmc_sq "$out"/pmc_mem
python bench.py > "$out/bench_n1.json" 2> "$out/bench_n1.log"
tail -c 600 "$out/bench_n1.json"
[ "${2:-}" = "main-only" ] && exit 0
# ---- the other kernels of the path: ShortestPath (BASELINE config 4 stand-in), WL-OA, config 5 (50 k graphs), and
# the multi-GPU code path at world size 1; kernel stats of the SP and WL-OA runs
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/sp_trace" -- python $root/tools/bench_sp.py 4110 5 > "$out/sp_config4_bench.json" 2> "$out/sp_trace.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/wloa_trace" -- python $root/tools/bench_wloa.py config3 > "$out/wloa.json" 2> "$out/wloa_trace.log"
# HBM counters of the same two runs (separate passes, as for the main bench)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$out/sp_pmc_$c" -- python $root/tools/bench_sp.py 4110 5 > /dev/null 2> "$out/sp_pmc_$c.log"
  rocprofv3 --pmc $c --output-format csv -d "$out/wloa_pmc_$c" -- python $root/tools/bench_wloa.py config3 > /dev/null 2> "$out/wloa_pmc_$c.log"
done
cd "$root"
python tools/pmc_summary.py "$out/sp_pmc_FETCH_SIZE" "$out/sp_pmc_WRITE_SIZE" > "$out/sp_config4_pmc_hbm_bytes.csv"
python tools/pmc_summary.py "$out/wloa_pmc_FETCH_SIZE" "$out/wloa_pmc_WRITE_SIZE" > "$out/wloa_pmc_hbm_bytes.csv"
rm -rf "$out"/sp_pmc_FETCH_SIZE "$out"/sp_pmc_WRITE_SIZE "$out"/wloa_pmc_FETCH_SIZE "$out"/wloa_pmc_WRITE_SIZE
cp "$(ls $out/sp_trace/*/*kernel_stats.csv | head -1)" "$out/sp_config4_kernel_stats.csv" 2>/dev/null
cp "$(ls $out/wloa_trace/*/*kernel_stats.csv | head -1)" "$out/wloa_kernel_stats.csv" 2>/dev/null
rm -rf "$out/sp_trace" "$out/wloa_trace"
bash tools/profile_config5.sh "$tag" > "$out/config5_profile.log" 2>&1
python tools/bench_sharded_1rank.py > "$out/sharded_1rank.txt" 2>&1
tail -c 300 "$out/config5_50k.json"; tail -3 "$out/sharded_1rank.txt"
python tools/kstats.py "$out/kernel_stats.csv" 30 > "$out/kernel_stats_top.txt" 2>&1
