mkdir -p gpurun_out/r06i
timeout 240 python -m pytest tests -m gpu -q -x -k "thousands_of_vertices or natural_fallbacks or published_like_sets_against or relabel_path or stream_relabel" > gpurun_out/r06i/tests.txt 2>&1; tail -3 gpurun_out/r06i/tests.txt
for s in dd reddit; do timeout 120 python tools/published_like.py $s wl 5 > gpurun_out/r06i/pub_${s}_wl.json 2> gpurun_out/r06i/pub_${s}_wl.log; python -c "
import json; z=json.load(open('gpurun_out/r06i/pub_${s}_wl.json')); print('$s', z['ms_per_step'], z['phases_ms'], z.get('relabel_route'), z['checks']['matches_reference_checksums'])"; done
