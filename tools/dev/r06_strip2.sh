mkdir -p gpurun_out/r06c
V="gram.strip=4 gram.strip=8 gram.strip=16"
( GK_AB_VERTICES=100 python tools/dev/rowblock_ab.py 10000 1250 $V
  GK_AB_VERTICES=100 python tools/dev/rowblock_ab.py 10000 5000 $V
  GK_AB_VERTICES=100 python tools/dev/rowblock_ab.py 20000 20000 $V
  GK_AB_VERTICES=100 python tools/dev/rowblock_ab.py 4000 4000 $V
  python tools/dev/rowblock_ab.py 20000 20000 $V
  python tools/dev/rowblock_ab.py 100000 12500 $V
  python tools/dev/rowblock_ab.py 50000 50000 gram.strip=16,gram.no_sym=1 gram.strip=8,gram.no_sym=1 gram.no_sym=1 gram.strip=12 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06c/strip_sweep2.txt
