#!/bin/bash
# panel pipeline A/B: parity tests, then bench + phase clocks for BNF_PANEL_RT = 2 and 4
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r02c}; mkdir -p "$OUT"; cd "$ROOT"
for rt in ${RTS:-4}; do
  echo "== pytest panel RT=$rt"; timeout 600 python -m pytest tests/test_gpu_panel.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4
  echo "== bench panel RT=$rt"
  BNF_PIPELINE=3 timeout 300 python bench.py --steps 20 --warmup 3 --profile-all --no-cpu-baseline > "$OUT/bench_panel_rt$rt.json" 2> "$OUT/bench_panel_rt$rt.err"
  python -c "import json;d=json.load(open('$OUT/bench_panel_rt$rt.json'));print('ms/step',round(d['ms_per_step'],3),'value',round(d['value']))"
  grep "\[bench\]" "$OUT/bench_panel_rt$rt.err" | head -4
  BNF_LIB=$ROOT/ab/libbnf_ablate.so BNF_PIPELINE=3 BNF_PHASE_PROF=panel_fwd_bwd timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep "phase clocks"
done
