for o in "gram.dd=0" "gram.dd=1" "gram.strip=2" "gram.strip=4" "gram.strip=8" "gram.strip=1"; do
  echo "== $o"
  GK_TOOL_OPTS="$o" python tools/bench_sp.py 4110 8 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['ms_per_fit_transform'],4), d['phases_ms'], d['gram_kernel_ms'])"
done
