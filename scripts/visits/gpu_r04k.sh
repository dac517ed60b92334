#!/bin/bash
# r04k: gemm_tn_ring with the second wave of every SIMD a quarter period out of phase
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04k; mkdir -p "$OUT"; cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== parity (dephase build)"
BNF_LIB=$ROOT/ab/libbnf_dephase.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_panel.py tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -5
one() {  # name lib
  local name=$1 lib=$2
  BNF_LIB=$lib timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-all 2> "$OUT/bench_$name.err" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name', 'ms/step', round(d['ms_per_step'],4), 'panel_us', round(d['roofline']['avg_launch_us'],1))"
  grep "\[bench\]" "$OUT/bench_$name.err" | head -4 | awk '{printf "   %s %s us", $2, $4} END {print ""}'
}
for rep in 1 2 3; do
  one cur $ROOT/bayesnf_amd/libbnf_hip.so
  one dephase $ROOT/ab/libbnf_dephase.so
done 2>&1 | tee "$OUT/ab.txt"
for c in C3 C4 C5; do for v in cur dephase; do lib=$ROOT/ab/libbnf_$v.so; echo -n "$c $v "; BNF_LIB=$lib timeout 600 python scripts/bench_configs.py $c 2>/dev/null | tail -1 | cut -c1-190; done; done | tee "$OUT/configs.txt"
