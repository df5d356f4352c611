mkdir -p gpurun_out/r06i
root=$(pwd); cd /tmp; export TMPDIR=/tmp
for s in dd reddit; do
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r06i/tr_$s -- python $root/tools/published_like.py $s wl 5 > /dev/null 2> $root/gpurun_out/r06i/tr_${s}.log
cd $root; python tools/kstats.py $(ls gpurun_out/r06i/tr_$s/*/*kernel_stats.csv | head -1) 14 > gpurun_out/r06i/pub_${s}_wl_kernel_stats.txt; rm -rf gpurun_out/r06i/tr_$s; head -14 gpurun_out/r06i/pub_${s}_wl_kernel_stats.txt; cd /tmp
done
