for o in "sp.pkw_waves=4" "sp.pkw_waves=8" "sp.pkw_waves=0"; do
  echo "== $o"
  GK_TOOL_OPTS="$o" python tools/bench_sp.py 4110 8 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_fit_transform'],4), d['phases_ms'])"
done
for w in collab; do for o in "sp.pkw_waves=4" "sp.pkw_waves=8"; do GK_TOOL_OPTS="$o" python tools/published_like.py $w sp 6 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$w $o', round(d['ms_per_fit_transform'],3), d['phases_ms'], d['checked_against_reference']['equals_full_set_fixture'])"; done; done
