#!/bin/bash
# r04d: weight fragments by buffer loads (no VALU address arithmetic), priority of the contraction phases, ring depth 8
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04d; mkdir -p "$OUT"; cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
echo "== pytest panel + parity + sweep (in-tree = cur)"
timeout 900 python -m pytest tests/test_gpu_panel.py tests/test_gpu_parity.py tests/test_gpu_sweep.py tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -8 | tee "$OUT/pytest_panel.txt"
one() {  # name lib [env...]
  local name=$1 lib=$2; shift 2
  env "$@" BNF_LIB=$lib timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-all 2> "$OUT/bench_$name.err" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name', 'ms/step', round(d['ms_per_step'],4), 'panel_us', round(d['roofline']['avg_launch_us'],1), 'frac', round(d['roofline']['frac'],4), 'loss', round(d['final_loss_mean'],3))"
}
for rep in 1 2 3; do
  one prev $ROOT/ab/libbnf_prev.so X=1
  one nosaddr $ROOT/ab/libbnf_nosaddr.so X=1
  one cur $ROOT/bayesnf_amd/libbnf_hip.so X=1
  one cprio1 $ROOT/ab/libbnf_cprio1.so X=1
  one cprio2 $ROOT/ab/libbnf_cprio2.so X=1
  one pd8 $ROOT/ab/libbnf_pd8.so X=1
done 2>&1 | tee "$OUT/ab.txt"
echo "t=$(( $(date +%s) - T0 ))s"
echo "== phase clocks (ablate build of the current code)"
THREADS="0 448" ABL_MASKS="0" bash scripts/gpu_phase_clocks.sh 2>&1 | tee "$OUT/phase_clocks.txt"
echo "t=$(( $(date +%s) - T0 ))s"
