#!/bin/bash
# r05aa: gemm_tn8 on 128 x 256 tiles where the layer's output is that wide (C5's layer 0): fp8 tests, C5/8 fp8 per-kernel, BNF_TN8_NARROW=1 against default
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r05aa}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest"; timeout 900 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_configs.py -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^E   +(Assertion|assert)" | cut -c1-220 | head -20 | tee "$OUT/pytest.txt"
for rep in 1 2; do for nar in 1 0; do
  if [ $nar = 1 ]; then export BNF_TN8_NARROW=1; else unset BNF_TN8_NARROW; fi
  echo "narrow=$nar $(BNF_BENCH_DTYPE=fp8 timeout 300 python scripts/bench_configs.py C5 2>/dev/null | tail -1 | grep -o '"member_steps_per_s": [0-9.]*') $(BNF_BENCH_DTYPE=fp8 timeout 200 python scripts/profile_config.py "C5/8 wind-like MAP (bf16)" 2>/dev/null | grep -E "wgrad_l0" | tr -s ' ')"
done; done 2>&1 | tee "$OUT/c5.txt"
