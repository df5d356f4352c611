#!/bin/bash
# per-kernel average durations of the bench (development): tools/dev/kstat.sh tag [bench args]
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python $root/bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline "$@" > $out/bench.json 2> $out/t.log
cd $root
python tools/kstats.py $out/t 45 > $out/kstats.txt
python tools/step_timeline.py $out/t > $out/timeline.txt 2>&1
rm -rf $out/t
cat $out/kstats.txt
