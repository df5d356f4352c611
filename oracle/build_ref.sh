#!/usr/bin/env bash
# Build a scratch copy of the real reference (ysig/GraKeL v0.1.11) OUTSIDE the repo so that
# tests/golden/make_golden.py and tests/test_oracle.py can import it in THIS container.
# /root/reference is read-only and its Cython extensions are not built, hence the copy.
# Nothing from the reference is ever copied into /root/repo.  (SURVEY.md 8c recipe.)
set -euo pipefail
SRC=${GK_REFERENCE:-/root/reference}
DST=${GK_REF_BUILD:-/tmp/grakel_oracle}
if [ ! -d "$SRC/grakel" ]; then echo "no reference at $SRC" >&2; exit 3; fi
if python3 -c "import sys; sys.path.insert(0,'$DST'); import grakel" 2>/dev/null; then
  echo "reference already built at $DST"; exit 0
fi
rm -rf "$DST"; mkdir -p "$DST"; cp -r "$SRC/." "$DST/"; chmod -R u+w "$DST"
( cd "$DST" && python3 setup.py build_ext --inplace > "$DST/build.log" 2>&1 )
python3 -c "import sys; sys.path.insert(0,'$DST'); import grakel; print('built grakel', grakel.__version__, 'at $DST')"
