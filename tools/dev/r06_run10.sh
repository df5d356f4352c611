mkdir -p gpurun_out/r06k
timeout 300 python -m pytest tests -m gpu -q -x -k "f64 or counts_above or exactness or published_like_sets_shortest or dense_gram or float_weights or split" > gpurun_out/r06k/tests2.txt 2>&1; tail -3 gpurun_out/r06k/tests2.txt | head -2
for s in dd reddit; do timeout 200 python tools/published_like.py $s sp 5 > gpurun_out/r06k/pub_${s}_sp.json 2> gpurun_out/r06k/pub_${s}_sp.log; python -c "
import json; z=json.load(open('gpurun_out/r06k/pub_${s}_sp.json')); print('$s', z['ms_per_fit_transform'], z['phases_ms'], z['checked_against_reference'] is not None)"; done
