mkdir -p gpurun_out/r06j
timeout 600 python -m pytest tests -m gpu -q -x > gpurun_out/r06j/gpu_suite.txt 2>&1; grep -E "passed|failed" gpurun_out/r06j/gpu_suite.txt | tail -1
timeout 300 python bench.py > gpurun_out/r06j/bench_n1.json 2> gpurun_out/r06j/bench_n1.log; python - <<'PY'
import json
b=json.load(open('gpurun_out/r06j/bench_n1.json'))
print(b['value'], b['ms_per_step'], b['roofline']['frac'], b.get('phases_ms'))
print({k:(v.get('ms_per_fit_transform') or v.get('ms_per_step')) if isinstance(v,dict) else v for k,v in b['extra'].items()})
PY
