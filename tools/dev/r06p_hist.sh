out=gpurun_out/r06p; mkdir -p $out
root=$(pwd); cd /tmp; export TMPDIR=/tmp
for o in "sp.hist_no_batch=0" "sp.hist_no_batch=1"; do
  GK_TOOL_OPTS="$o" timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/pt -- python $root/tools/bench_sp.py 4110 5 > $root/$out/pt.log 2>&1
  f=$(find $root/$out/pt -name "*kernel_stats.csv" | head -1); echo "$o"; grep -E "sp_hist_kernel|gm_rows_kernel" $f | cut -c1-20,200-300
  rm -rf $root/$out/pt
done
