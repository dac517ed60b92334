#!/bin/bash
# r03 visit D: GPU tests, same-box A/B (baseline kernels / HEAD / static priority), phase clocks
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03d}; mkdir -p "$OUT"; cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -8 "$OUT/pytest.txt"
echo "== A/B C2"; VARIANTS="prev:ab/libbnf_prev.so new: prio:ab/libbnf_prio.so" STEPS=30 REPS=3 bash scripts/gpu_abn.sh 2>&1 | tee "$OUT/ab.txt"
echo "== phase clocks"; THREADS="0 448" bash scripts/gpu_phase_clocks.sh 2>&1 | tee "$OUT/phase_clocks.txt"
