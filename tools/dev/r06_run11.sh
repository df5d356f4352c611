mkdir -p gpurun_out/r06l
timeout 400 python -m pytest tests -m gpu -q -x -k "er_sets or config3 or small_sets or mutag or graph_major or published_like_sets_against or config5_full" > gpurun_out/r06l/tests.txt 2>&1; tail -3 gpurun_out/r06l/tests.txt | head -2
timeout 200 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r06l/bench.json 2> gpurun_out/r06l/bench.log; python -c "
import json; b=json.load(open('gpurun_out/r06l/bench.json')); print(b['ms_per_step'], b['phases_ms'])"
for w in config5 collab nci1; do timeout 200 python tools/published_like.py $w wl 5 > gpurun_out/r06l/$w.json 2>/dev/null || timeout 200 python bench.py --workload $w --no-cpu-baseline --no-extras --steps 5 > gpurun_out/r06l/$w.json 2>/dev/null; python -c "
import json; z=json.load(open('gpurun_out/r06l/$w.json')); print('$w', z.get('ms_per_step'), z.get('phases_ms'))"; done
