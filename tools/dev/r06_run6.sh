mkdir -p gpurun_out/r06h
./tools/micro/lanexor
python -m pytest tests -m gpu -q -x -k "signature or dense_graphs or hubs or lookup or published_like_sets_against or natural_fallbacks or transform" > gpurun_out/r06h/sig_tests.txt 2>&1; tail -3 gpurun_out/r06h/sig_tests.txt
root=$(pwd); cd /tmp; export TMPDIR=/tmp
for s in collab reddit; do
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r06h/tr_$s -- python $root/tools/published_like.py $s wl 5 > $root/gpurun_out/r06h/pub_${s}_wl.json 2> $root/gpurun_out/r06h/pub_${s}_wl.log
cd $root; python tools/kstats.py $(ls gpurun_out/r06h/tr_$s/*/*kernel_stats.csv | head -1) 12 > gpurun_out/r06h/pub_${s}_wl_kernel_stats.txt; rm -rf gpurun_out/r06h/tr_$s; head -9 gpurun_out/r06h/pub_${s}_wl_kernel_stats.txt; cd /tmp
done
