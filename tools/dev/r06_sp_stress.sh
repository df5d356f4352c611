# the ShortestPath / host-driven-relabel tests of round 6 in fresh processes, N times (races behind the LDS-only barriers,
# the batched histogram kernel, the converged-partition copies would show as a flaky comparison)
n=${1:-12}; out=gpurun_out/r06_sp_stress.txt; : > $out
for i in $(seq 1 $n); do
  r=$(timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -p no:cacheprovider -k "shortest or sp_ or large_unit or published or converged or dense_graphs or operand or thousands" 2>&1 | grep -E "passed|failed|error" | tail -1)
  echo "iteration $i: $r" >> $out
done
cat $out
