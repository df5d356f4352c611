#!/bin/bash
# per-kernel average durations of any python script (development): tools/dev/kstat_py.sh tag script.py [args]
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python $root/"$@" > $out/out.txt 2> $out/t.log
cd $root
python tools/kstats.py $out/t 30 > $out/kstats.txt
rm -rf $out/t
cat $out/kstats.txt
