#!/bin/bash
# r04s: 64-row panels, two workgroups per CU (W = 512, non-H0L form) against the 128-row non-H0L form -- does a second,
# independent workgroup on the CU hide one's MFMA phases under the other's epilogues?
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04s; mkdir -p "$OUT"; cd "$ROOT"
export BNF_LIB=$ROOT/ab/libbnf_bm64.so BNF_PANEL_NO_H0L=1
BNF_PANEL_BM64=1 timeout 900 python -m pytest tests/test_gpu_panel.py -m gpu -q -p no:cacheprovider -k "step_vs_oracle or minibatch" 2>&1 | grep -E "passed|failed" | tail -2
one() {
  timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-all 2> "$OUT/err_$1.txt" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', 'ms/step', round(d['ms_per_step'],4), 'panel_us', round(d['roofline']['avg_launch_us'],1))"
}
for rep in 1 2; do
  one bm128
  BNF_PANEL_BM64=1 one bm64
done 2>&1 | tee "$OUT/ab.txt"
