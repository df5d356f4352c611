#!/bin/bash
# The published-dataset stand-ins on the GPU box (run from the repo root):  tools/profile_published.sh r05 [sets...]
#   -> gpurun_out/<tag>/pub_<set>_{wl,sp,wl_e2e}.json + pub_<set>_{wl,sp}_kernel_stats.csv (rocprofv3 --kernel-trace --stats)
# Every run is its own process under a timeout: a route that turns out to take minutes must not eat the GPU budget.
set -u
tag=${1:-r05}; shift
sets=${*:-nci1 dd reddit collab}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for s in $sets; do
  for mode in wl sp wl_e2e; do
    timeout ${PUB_TIMEOUT:-150} python $root/tools/published_like.py $s $mode 6 > "$out/pub_${s}_${mode}.json" 2> "$out/pub_${s}_${mode}.log" \
      || echo "{\"error\": \"rc $? (timeout ${PUB_TIMEOUT:-150}s or failure)\"}" >> "$out/pub_${s}_${mode}.json"
    tail -c 700 "$out/pub_${s}_${mode}.json"; echo
  done
  for mode in ${PUB_TRACE:-wl sp}; do
    timeout ${PUB_TIMEOUT:-150} rocprofv3 --kernel-trace --stats --output-format csv -d "$out/pt_${s}_${mode}" -- \
      python $root/tools/published_like.py $s $mode 3 > /dev/null 2> "$out/pt_${s}_${mode}.log"
    cp "$(ls $out/pt_${s}_${mode}/*/*kernel_stats.csv 2>/dev/null | head -1)" "$out/pub_${s}_${mode}_kernel_stats.csv" 2>/dev/null
    rm -rf "$out/pt_${s}_${mode}"
    python $root/tools/kstats.py "$out/pub_${s}_${mode}_kernel_stats.csv" 12 2>/dev/null
  done
done
