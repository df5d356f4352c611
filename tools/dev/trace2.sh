#!/bin/bash
# kernel timelines of a bench step for the stream route and the host-driven route (development)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${1:-tr}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_stream -- $B > $out/bench_stream.json 2> $out/t_stream.log
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_host -- $B --opt wl.no_stream=1 > $out/bench_host.json 2> $out/t_host.log
cd $root
python tools/step_timeline.py $out/t_stream > $out/timeline_stream.txt 2>&1
python tools/step_timeline.py $out/t_host > $out/timeline_host.txt 2>&1
rm -rf $out/t_stream $out/t_host
$B > $out/bench_plain_stream.json 2>/dev/null
$B --opt wl.no_stream=1 > $out/bench_plain_host.json 2>/dev/null
