mkdir -p gpurun_out/r06c
python tools/gram_only.py 10000 gram.strip=4 gram.strip=8 gram.strip=16 gram.strip=2 gram.strip=8,gram.xcc=1 2>&1 | grep -v "^$" | tee gpurun_out/r06c/strip_config3.txt | cut -c1-150
python tools/dev/rowblock_ab.py 200000 25000 gram.strip=4 gram.strip=8 gram.strip=16 gram.strip=32 2>&1 | tee gpurun_out/r06c/strip_rowblock_200k.txt | tail -6
python tools/dev/rowblock_ab.py 50000 6250 gram.strip=4 gram.strip=8 gram.strip=16 gram.strip=32 2>&1 | tee gpurun_out/r06c/strip_rowblock_50k.txt | tail -6
python tools/dev/rowblock_ab.py 50000 50000 gram.strip=4 gram.strip=8 gram.strip=16 gram.strip=32 2>&1 | tee gpurun_out/r06c/strip_sym_50k.txt | tail -6
