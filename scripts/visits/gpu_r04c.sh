#!/bin/bash
# r04c: row phase in every wave + layer 0 folded / swapped (F0 forms): whole -m gpu suite on the in-tree build, the panel file
# again with the fold switched off, alternated same-box benches, phase clocks.
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04c; mkdir -p "$OUT"; cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
echo "== pytest panel + parity + sweep (in-tree)"
timeout 900 python -m pytest tests/test_gpu_panel.py tests/test_gpu_parity.py tests/test_gpu_sweep.py -m gpu -q -p no:cacheprovider 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -15 | tee "$OUT/pytest_panel.txt"
echo "t=$(( $(date +%s) - T0 ))s"
echo "== pytest panel file, BNF_PANEL_FOLD0=0"
BNF_PANEL_FOLD0=0 timeout 600 python -m pytest tests/test_gpu_panel.py -m gpu -q -p no:cacheprovider 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -4 | tee "$OUT/pytest_nofold.txt"
one() {  # name lib [env...]
  local name=$1 lib=$2; shift 2
  env "$@" BNF_LIB=$lib timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-all 2> "$OUT/bench_$name.err" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name', 'ms/step', round(d['ms_per_step'],4), 'panel_us', round(d['roofline']['avg_launch_us'],1), 'frac', round(d['roofline']['frac'],4), 'loss', round(d['final_loss_mean'],3))"
}
for rep in 1 2 3; do
  one prev $ROOT/ab/libbnf_prev.so X=1
  one step1 $ROOT/ab/libbnf_new.so X=1
  one cur_nofold $ROOT/bayesnf_amd/libbnf_hip.so BNF_PANEL_FOLD0=0
  one cur $ROOT/bayesnf_amd/libbnf_hip.so X=1
  one norows $ROOT/ab/libbnf_norows.so X=1
done 2>&1 | tee "$OUT/ab.txt"
echo "t=$(( $(date +%s) - T0 ))s"
for n in prev cur; do grep "\[bench\]" "$OUT/bench_$n.err" | head -8 | tee "$OUT/bench_${n}_hip_events.txt"; done
echo "== phase clocks (ablate build of the current code)"
THREADS="0 448" ABL_MASKS="0" bash scripts/gpu_phase_clocks.sh 2>&1 | tee "$OUT/phase_clocks.txt"
echo "t=$(( $(date +%s) - T0 ))s"
echo "== whole -m gpu suite"
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -15 | tee "$OUT/pytest_gpu.txt"
echo "t=$(( $(date +%s) - T0 ))s"
