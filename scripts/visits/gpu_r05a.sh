#!/bin/bash
# r05a: first visit of the L1T panel kernel (last layer on transposed tiles, activation evaluated once): GPU suite,
# per-leaf errors of both builds, same-box A/B at C2 and at C3 / C5 against the round-4 kernels (ab/libbnf_prev.so)
set -u; ulimit -c 0
TAG=${1:-r05a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest panel"; timeout 900 python -m pytest tests/test_gpu_panel.py tests/test_gpu_parity.py -x -q -p no:cacheprovider 2>&1 | tail -15 | tee "$OUT/pytest_panel.txt"
echo "== leaf diag (new)"; timeout 300 python scripts/l1t_leaf_diag.py 2>&1 | grep -v "^/opt" | tee "$OUT/leaf_new.txt"
echo "== leaf diag (prev)"; BNF_LIB=$ROOT/ab/libbnf_prev.so timeout 300 python scripts/l1t_leaf_diag.py 2>&1 | grep -v "^/opt" | tee "$OUT/leaf_prev.txt"
echo "== A/B C2"; VARIANTS="prev:ab/libbnf_prev.so new:" REPS=3 bash scripts/gpu_abn.sh 2>&1 | tee "$OUT/ab_c2.txt"
echo "== A/B configs"; for c in C3 C5; do for v in prev new; do if [ $v = prev ]; then export BNF_LIB=$ROOT/ab/libbnf_prev.so; else unset BNF_LIB; fi; echo -n "$c $v "; timeout 600 python scripts/bench_configs.py $c 2>/dev/null | tail -1 | cut -c1-260; done; done | tee "$OUT/ab_configs.txt"; unset BNF_LIB
echo "== pytest -m gpu (all)"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -8 | tee "$OUT/pytest_gpu.txt"
