#!/bin/bash
# r05n: dgrad0 tile prefetch on the 128-feature forms: panel tests, C5 (bf16 + fp8) and C2 A/B against the previous build
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r05n}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest"; timeout 900 python -m pytest tests/test_gpu_panel.py tests/test_gpu_configs.py tests/test_gpu_fp8.py -x -q -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2 3; do for dt in bf16 fp8; do for v in prev new; do if [ $v = prev ]; then export BNF_LIB=$ROOT/ab/libbnf_prev.so; else unset BNF_LIB; fi; echo -n "C5 $dt $v "; BNF_BENCH_DTYPE=$dt timeout 600 python scripts/bench_configs.py C5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['member_steps_per_s'])"; done; done; done | tee "$OUT/ab_c5.txt"; unset BNF_LIB
echo "== C2"; VARIANTS="prev:ab/libbnf_prev.so new:" REPS=2 bash scripts/gpu_abn.sh 2>&1 | tee "$OUT/ab_c2.txt"
