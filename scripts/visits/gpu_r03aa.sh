#!/bin/bash
# r03 visit AA: reference shuffles drawn on the device (bnf_row_keys): full GPU suite, time per epoch of the draw
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03aa}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest.txt" 2>&1; echo "rc=$?"; grep -E "passed|failed|^FAILED|^E  " "$OUT/pytest.txt" | tail -12 | cut -c1-250
echo "== shuffle draw cost"; timeout 600 python scripts/shuffle_cost.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee "$OUT/shuffle_cost.txt"
