#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/${1:-ks_sp}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- python $root/tools/bench_sp.py 4110 10 > $out/bench.json 2> $out/t.log
cd $root
python tools/kstats.py $out/t 40 > $out/kstats.txt
rm -rf $out/t
cat $out/kstats.txt; tail -c 900 $out/bench.json
