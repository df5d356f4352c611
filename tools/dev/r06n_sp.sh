# round 6, byte-wide distance matrices: SP route tests, then the published-like SP runs (asserted) with kernel stats
out=gpurun_out/r06n; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shortest or large_unit or sp_ or published" > $out/tests.txt 2>&1; tail -5 $out/tests.txt
for w in reddit dd collab; do
  timeout 300 python tools/published_like.py $w sp 6 > $out/pub_${w}_sp.json 2> $out/pub_${w}_sp.log; cut -c1-600 $out/pub_${w}_sp.json
  GK_TOOL_OPTS="sp.rows_no_merge=4" timeout 300 python tools/published_like.py $w sp 6 > $out/pub_${w}_sp_nomerge.json 2>> $out/pub_${w}_sp.log; cut -c150-420 $out/pub_${w}_sp_nomerge.json
done
root=$(pwd); cd /tmp; export TMPDIR=/tmp
for w in reddit dd; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/pt_$w -- python $root/tools/published_like.py $w sp 6 > $root/$out/pt_$w.log 2>&1
  f=$(find $root/$out/pt_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $root/$out/pub_${w}_sp_kernel_stats.csv && head -14 $f | cut -c1-110
  rm -rf $root/$out/pt_$w
done
