#!/bin/bash
# quick visit: panel tests, same-box A/B at C2 (+ C3 / C5 when CFG=1), phase clocks of ab/libbnf_ablate.so
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-quick}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest panel"; timeout 900 python -m pytest tests/test_gpu_panel.py tests/test_gpu_parity.py ${MORE_TESTS:-} -x -q -p no:cacheprovider 2>&1 | tail -4 | tee "$OUT/pytest_panel.txt"
echo "== A/B C2"; VARIANTS="${VARIANTS:-prev:ab/libbnf_prev.so new:}" REPS=${REPS:-3} bash scripts/gpu_abn.sh 2>&1 | tee "$OUT/ab_c2.txt"
if [ "${CFG:-0}" = 1 ]; then
echo "== A/B configs"; for c in ${CONFIGS:-C3 C5}; do for v in prev new; do if [ $v = prev ]; then export BNF_LIB=$ROOT/ab/libbnf_prev.so; else unset BNF_LIB; fi; echo -n "$c $v "; timeout 600 python scripts/bench_configs.py $c 2>/dev/null | tail -1 | cut -c1-260; done; done | tee "$OUT/ab_configs.txt"; unset BNF_LIB
fi
if [ "${CLK:-1}" = 1 ]; then LIBS="${LIBS:-ablate}" THREADS="${THREADS:-0 448}" bash scripts/visits/gpu_clk.sh ${1:-quick}; fi
