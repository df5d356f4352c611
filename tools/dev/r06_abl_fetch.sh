mkdir -p gpurun_out/r06m
root=$(pwd); cd /tmp; export TMPDIR=/tmp
for a in 0 8 32; do
  GK_GRAM_ABL=$a timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/r06m/f_$a -- python $root/tools/gram_only.py 10000 --abl > $root/gpurun_out/r06m/run_$a.txt 2>&1
  cd $root; python tools/pmc_summary.py gpurun_out/r06m/f_$a 2>/dev/null | grep gram_ws | head -2; grep "gemm ms" gpurun_out/r06m/run_$a.txt | head -1 | cut -c1-100; rm -rf gpurun_out/r06m/f_$a; cd /tmp
done
