#!/bin/bash
# Timing ablations of the Gram kernel on the config-3 operand (tools' build of the library: make -C grakel_amd/csrc abl).
# GK_GRAM_ABL bit mask: 1 no operand loads, 2 no multiply (K-steps skipped), 4 MFMA on fabricated fragments (no LDS reads),
# 8 no stores, 16 no per-K-step barrier, 32 the store waves skip their batches altogether, 64 no parking, 128 K-step barrier only
# every second step.      bash tools/gram_ablate.sh [masks...] > gpurun_out/gram_ablation.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
masks=${@:-"0 8 32 10 1 9 3 11 43 128"}
for a in $masks; do
  printf "abl %2d: " $a
  GK_GRAM_ABL=$a python tools/gram_only.py 10000 --abl 2>&1 | grep -E "gemm ms|wg0" | sed "s/TOP.*//"
done
