#!/bin/bash
# r04n: dgrad0 / layer-0 weight fragments by raw buffer loads
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04n; mkdir -p "$OUT"; cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" | tail -4
one() {  # name lib
  local name=$1 lib=$2
  BNF_LIB=$lib timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name', 'ms/step', round(d['ms_per_step'],4), 'panel_us', round(d['roofline']['avg_launch_us'],1))"
}
for rep in 1 2 3; do
  one before $ROOT/ab/libbnf_before.so
  one cur $ROOT/bayesnf_amd/libbnf_hip.so
done 2>&1 | tee "$OUT/ab.txt"
