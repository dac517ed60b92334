#!/bin/bash
# r04a: step-1 panel micro-optimisations (forward median form, DPP sums, zero-C contraction start, layer-0 prefetch),
# attribution builds with one knob off each, the packed-clamp option and the in-kernel dK0 experiment -- parity tests per
# build that changes arithmetic, then alternated same-box benches.
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04a; mkdir -p "$OUT"; cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
T0=$(date +%s)
echo "== pytest (in-tree = new): panel + parity + sweep"
timeout 900 python -m pytest tests/test_gpu_panel.py tests/test_gpu_parity.py tests/test_gpu_sweep.py -m gpu -q -p no:cacheprovider -x 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -6 | tee "$OUT/pytest_new.txt"
echo "t=$(( $(date +%s) - T0 ))s"
echo "== pytest panel file, pkclamp build"
BNF_LIB=$ROOT/ab/libbnf_pkclamp.so timeout 600 python -m pytest tests/test_gpu_panel.py -m gpu -q -p no:cacheprovider -x 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -4 | tee "$OUT/pytest_pkclamp.txt"
echo "== pytest panel file, dk0 build, BNF_PANEL_DK0=1"
BNF_PANEL_DK0=1 BNF_LIB=$ROOT/ab/libbnf_dk0.so timeout 600 python -m pytest tests/test_gpu_panel.py -m gpu -q -p no:cacheprovider 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -8 | tee "$OUT/pytest_dk0.txt"
echo "t=$(( $(date +%s) - T0 ))s"
one() {  # name lib [env...]
  local name=$1 lib=$2; shift 2
  env "$@" BNF_LIB=$lib timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-all 2> "$OUT/bench_$name.err" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name', 'ms/step', round(d['ms_per_step'],4), 'panel_us', round(d['roofline']['avg_launch_us'],1), 'frac', round(d['roofline']['frac'],4), 'loss', round(d['final_loss_mean'],3))"
}
for rep in 1 2; do
  one prev $ROOT/ab/libbnf_prev.so X=1
  one new $ROOT/bayesnf_amd/libbnf_hip.so X=1
  one pkclamp $ROOT/ab/libbnf_pkclamp.so X=1
  one nofwds $ROOT/ab/libbnf_nofwds.so X=1
  one nodpp $ROOT/ab/libbnf_nodpp.so X=1
  one nozpeel $ROOT/ab/libbnf_nozpeel.so X=1
  one nopre0 $ROOT/ab/libbnf_nopre0.so X=1
  one dk0off $ROOT/ab/libbnf_dk0.so X=1
  one dk0on $ROOT/ab/libbnf_dk0.so BNF_PANEL_DK0=1
done 2>&1 | tee "$OUT/ab.txt"
echo "t=$(( $(date +%s) - T0 ))s"
grep "\[bench\]" "$OUT/bench_new.err" | head -8 | tee "$OUT/bench_new_hip_events.txt"
grep "\[bench\]" "$OUT/bench_dk0on.err" | head -8 | tee "$OUT/bench_dk0on_hip_events.txt"
grep "\[bench\]" "$OUT/bench_prev.err" | head -8 | tee "$OUT/bench_prev_hip_events.txt"
echo "== phase clocks (ablate build of the new code): waves 0 and 7, masks 0 / 16 (no A0 MFMAs) / 8 (no panel copies to HBM) / 4 (no LDS stores)"
THREADS="0 448" ABL_MASKS="0 16 8 4" bash scripts/gpu_phase_clocks.sh 2>&1 | tee "$OUT/phase_clocks.txt"
echo "t=$(( $(date +%s) - T0 ))s"
