#!/bin/bash
# r04f: LDS-DMA of the weight-gradient stream kernels as raw buffer loads; panel copies by raw buffer stores; one LDS base per tile
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04f; mkdir -p "$OUT"; cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
echo "== pytest -m gpu (in-tree = cur)"
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -8 | tee "$OUT/pytest_gpu.txt"
one() {  # name lib [env...]
  local name=$1 lib=$2; shift 2
  env "$@" BNF_LIB=$lib timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-all 2> "$OUT/bench_$name.err" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name', 'ms/step', round(d['ms_per_step'],4), 'panel_us', round(d['roofline']['avg_launch_us'],1), 'frac', round(d['roofline']['frac'],4), 'loss', round(d['final_loss_mean'],3))"
  grep "\[bench\]" "$OUT/bench_$name.err" | head -4 | awk '{printf "   %s %s us", $2, $4} END {print ""}'
}
for rep in 1 2 3; do
  one prev $ROOT/ab/libbnf_prev.so X=1
  one r04e $ROOT/ab/libbnf_r04e.so X=1
  one nobufdma $ROOT/ab/libbnf_nobufdma.so X=1
  one cur $ROOT/bayesnf_amd/libbnf_hip.so X=1
done 2>&1 | tee "$OUT/ab.txt"
echo "t=$(( $(date +%s) - T0 ))s"
echo "== phase clocks (ablate build of the current code)"
THREADS="0 448" ABL_MASKS="0" bash scripts/gpu_phase_clocks.sh 2>&1 | tee "$OUT/phase_clocks.txt"
echo "== configs"; for c in C3 C4 C5; do timeout 600 python scripts/bench_configs.py $c 2>/dev/null | tail -1; done > "$OUT/configs_bench.jsonl"; cut -c1-300 "$OUT/configs_bench.jsonl"
echo "t=$(( $(date +%s) - T0 ))s"
