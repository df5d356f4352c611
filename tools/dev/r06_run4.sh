mkdir -p gpurun_out/r06f
python -m pytest tests -m gpu -q -x -k "shortest_path or sp_ or nci1 or core or caller or mutag or small_sets or published" > gpurun_out/r06f/sp_tests.txt 2>&1; tail -3 gpurun_out/r06f/sp_tests.txt
python tools/bench_sp.py 4110 8 > gpurun_out/r06f/sp_config4.json 2>gpurun_out/r06f/sp.log; python -c "
import json; z=json.load(open('gpurun_out/r06f/sp_config4.json')); print({k:z[k] for k in z if 'ms' in k or 'phase' in k})"
