out=gpurun_out/r06u; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "converged or wave_signature or published or route or partition or mutag or small_sets or thousands" > $out/tests.txt 2>&1; tail -3 $out/tests.txt
for w in collab reddit dd nci1; do
  timeout 300 python tools/published_like.py $w wl 6 > $out/pub_${w}_wl.json 2> $out/pub_${w}_wl.log; python -c "
import json; d=json.load(open('$out/pub_${w}_wl.json')); print('$w', round(d['ms_per_step'],3), d['phases_ms'], d.get('relabel_route','')[:14], d['checks']['matches_reference_checksums'])"
done
