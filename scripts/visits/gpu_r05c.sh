#!/bin/bash
# r05c: L1T with the dZ_L copy under the next contraction: panel tests, same-box A/B at C2 / C3 / C5, phase clocks
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r05c}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest panel"; timeout 900 python -m pytest tests/test_gpu_panel.py tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -p no:cacheprovider 2>&1 | tail -5 | tee "$OUT/pytest_panel.txt"
echo "== A/B C2"; VARIANTS="prev:ab/libbnf_prev.so new:" REPS=3 bash scripts/gpu_abn.sh 2>&1 | tee "$OUT/ab_c2.txt"
echo "== A/B configs"; for c in C3 C5; do for v in prev new; do if [ $v = prev ]; then export BNF_LIB=$ROOT/ab/libbnf_prev.so; else unset BNF_LIB; fi; echo -n "$c $v "; timeout 600 python scripts/bench_configs.py $c 2>/dev/null | tail -1 | cut -c1-260; done; done | tee "$OUT/ab_configs.txt"; unset BNF_LIB
for lib in ablate; do
  for thr in 0 448; do
    echo "=== $lib thread $thr"
    BNF_ABLATE=$(( thr * 256 )) BNF_LIB=$ROOT/ab/libbnf_$lib.so BNF_PHASE_PROF=panel_fwd_bwd timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep "phase clocks" | sed 's/.*total/total/' | tail -1
  done
done | tee "$OUT/phase_clocks.txt"
