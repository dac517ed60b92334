#!/bin/bash
# r05j: C5/8 -- per-kernel table in bf16 and fp8, phase clocks of its panel form (<4, 2, true, ., 1, 128>), waves 0 / 7
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r05j}; mkdir -p "$OUT"; cd "$ROOT"
for dt in bf16 fp8; do echo "== C5/8 $dt"; BNF_BENCH_DTYPE=$dt timeout 300 python scripts/profile_config.py "C5/8 wind-like MAP (bf16)" 2>/dev/null; done | tee "$OUT/c5_profiles.txt"
for thr in 0 448; do echo "=== C5 thread $thr"; BNF_ABLATE=$(( thr * 256 )) BNF_LIB=$ROOT/ab/libbnf_ablate.so BNF_PHASE_PROF=panel_fwd_bwd timeout 300 python scripts/bench_configs.py C5 2>&1 | grep "phase clocks" | sed 's/.*total/total/' | tail -1; done | tee "$OUT/c5_phase_clocks.txt"
