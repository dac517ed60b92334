#!/bin/bash
# r04m: k_featurize_cols (lane = row, wave = column block, 64-row workgroups) against k_featurize
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04m; mkdir -p "$OUT"; cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -12 | tee "$OUT/pytest_gpu.txt"
one() {  # name [env...]
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-all 2> "$OUT/bench_$name.err" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name', 'ms/step', round(d['ms_per_step'],4), 'panel_us', round(d['roofline']['avg_launch_us'],1))"
  grep "\[bench\]" "$OUT/bench_$name.err" | head -6 | awk '{printf "   %s %s us", $2, $4} END {print ""}'
}
for rep in 1 2 3; do
  one rowwalk BNF_FEAT_COLS=0
  one cols X=1
done 2>&1 | tee "$OUT/ab.txt"
for c in C3 C4 C5; do for f in 0 1; do echo -n "$c cols=$f "; BNF_FEAT_COLS=$f timeout 600 python scripts/bench_configs.py $c 2>/dev/null | tail -1 | cut -c1-190; done; done | tee "$OUT/configs.txt"
for c in "C5/8 wind-like MAP (bf16)"; do for f in 0 1; do echo "== $c cols=$f"; BNF_FEAT_COLS=$f timeout 200 python scripts/profile_config.py "$c" 2>/dev/null | grep featurize; done; done | tee -a "$OUT/configs.txt"
