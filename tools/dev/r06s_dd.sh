out=gpurun_out/r06s; mkdir -p $out
root=$(pwd); cd /tmp; export TMPDIR=/tmp
for w in dd collab; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/pt_$w -- python $root/tools/published_like.py $w sp 4 > $root/$out/pt_$w.log 2>&1
f=$(find $root/$out/pt_$w -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $root/$out/pub_${w}_sp_kernel_stats.csv && head -8 $f | cut -c1-60,150-330
rm -rf $root/$out/pt_$w
done
