# the whole GPU suite in fresh processes, N times
n=${1:-5}; out=gpurun_out/r06_full_stress.txt; : > $out
for i in $(seq 1 $n); do
  r=$(timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -1)
  echo "iteration $i: $r" >> $out
done
cat $out
