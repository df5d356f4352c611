mkdir -p gpurun_out/r06k
timeout 200 python -m pytest tests -m gpu -q -x -k "large_unit or shortest_path_route or published_like_sets_shortest" > gpurun_out/r06k/tests.txt 2>&1; tail -3 gpurun_out/r06k/tests.txt | head -2
for s in dd reddit collab; do timeout 200 python tools/published_like.py $s sp 5 > gpurun_out/r06k/pub_${s}_sp.json 2> gpurun_out/r06k/pub_${s}_sp.log; python -c "
import json; z=json.load(open('gpurun_out/r06k/pub_${s}_sp.json')); print('$s', z['ms_per_fit_transform'], z['phases_ms'], z['checked_against_reference'] is not None)"; done
