mkdir -p gpurun_out/r06e
root=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $root/gpurun_out/r06e/tr -- python $root/tools/bench_sp.py 4110 6 > $root/gpurun_out/r06e/sp_config4.json 2> $root/gpurun_out/r06e/sp_trace.log
cd $root
cp $(ls gpurun_out/r06e/tr/*/*kernel_trace.csv | head -1) gpurun_out/r06e/sp_kernel_trace.csv; rm -rf gpurun_out/r06e/tr
python tools/dev/trace_gaps.py gpurun_out/r06e/sp_kernel_trace.csv gram_low_gm -2 | tee gpurun_out/r06e/sp_step_timeline.txt
tail -c 400 gpurun_out/r06e/sp_config4.json
