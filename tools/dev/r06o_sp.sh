# round 6: operand type from the exact self similarities -- SP tests, config 4 (untraced + kernel stats), published-like SP
out=gpurun_out/r06o; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shortest or large_unit or sp_ or published or operand or core or graph_kernel" > $out/tests.txt 2>&1; tail -5 $out/tests.txt
timeout 200 python tools/bench_sp.py 4110 8 > $out/sp_config4.json 2> $out/sp_config4.log; cut -c1-900 $out/sp_config4.json
for w in collab reddit dd nci1; do
  timeout 300 python tools/published_like.py $w sp 6 > $out/pub_${w}_sp.json 2> $out/pub_${w}_sp.log; python -c "
import json; d=json.load(open('$out/pub_${w}_sp.json')); print('$w', round(d['ms_per_fit_transform'],3), d['phases_ms'], d.get('operand'), d.get('checked_against_reference',{}).get('equals_full_set_fixture'))"
done
root=$(pwd); cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/pt -- python $root/tools/bench_sp.py 4110 5 > $root/$out/pt.log 2>&1
f=$(find $root/$out/pt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $root/$out/sp_config4_kernel_stats.csv && head -12 $f | cut -c1-120
rm -rf $root/$out/pt
