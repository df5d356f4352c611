mkdir -p gpurun_out/r06a
python -m pytest tests -m gpu -q --deselect "tests/test_gpu_parity.py::test_published_like_sets_shortest_path_at_full_size[reddit]" > gpurun_out/r06a/gpu_suite.txt 2>&1
tail -5 gpurun_out/r06a/gpu_suite.txt
python bench.py > gpurun_out/r06a/bench_n1.json 2> gpurun_out/r06a/bench_n1.log
tail -c 400 gpurun_out/r06a/bench_n1.json
root=$(pwd); cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r06a/trace -- python $root/bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $root/gpurun_out/r06a/bench_under_trace.json 2> $root/gpurun_out/r06a/trace.log
cd $root
cp $(ls gpurun_out/r06a/trace/*/*kernel_trace.csv | head -1) gpurun_out/r06a/kernel_trace.csv
cp $(ls gpurun_out/r06a/trace/*/*kernel_stats.csv | head -1) gpurun_out/r06a/kernel_stats.csv
rm -rf gpurun_out/r06a/trace
for s in dd reddit collab; do timeout 300 python tools/published_like.py $s sp 5 > gpurun_out/r06a/pub_${s}_sp.json 2> gpurun_out/r06a/pub_${s}_sp.log; tail -c 300 gpurun_out/r06a/pub_${s}_sp.json; tail -2 gpurun_out/r06a/pub_${s}_sp.log; done
