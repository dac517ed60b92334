#!/bin/bash
# A/B of the step plumbing around the panel kernel: side stream for the feature kernels
# (BNF_SIDE), one fused weight-packing launch (BNF_PACK_SPLIT=1 = the four old launches)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r02s}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest panel + parity"; timeout 900 python -m pytest tests/test_gpu_panel.py tests/test_gpu_parity.py tests/test_gpu_sweep.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4
run() {  # name, env...
  local name=$1; shift
  for rep in 1 2; do
    env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
    python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print('$name ms/step',round(d['ms_per_step'],4),'value',round(d['value']),'panel us',round(d['roofline']['avg_launch_us'],1))"
  done
}
run base BNF_SIDE=0 BNF_PACK_SPLIT=1
run pack BNF_SIDE=0
run side BNF_SIDE=1 BNF_PACK_SPLIT=1
run both BNF_SIDE=1
env BNF_SIDE=0 timeout 300 python bench.py --steps 10 --warmup 3 --profile-all --no-cpu-baseline 2>&1 >/dev/null | grep "\[bench\]"
