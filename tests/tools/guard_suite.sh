#!/bin/bash
# The whole GPU suite with red zones around every device block (option debug.guard, api.hip): a write outside a block
# fails the test that made it.  usage (GPU box, repo root): tests/tools/guard_suite.sh [out_file]
out=${1:-gpurun_out/guard_suite.txt}
GK_TEST_GUARD=1 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider 2>&1 | grep -E "passed|failed|red zone|Error" | tail -15 > "$out"
cat "$out"
