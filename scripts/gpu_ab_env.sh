#!/bin/bash
# same-box A/B of bench.py under different environment switches:
#   gpu_ab_env.sh <outdir> "<name>:<ENV=..> <ENV=..>" ...      (first runs the panel / parity tests)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-ab}; shift; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest panel + parity + sweep"
timeout 900 python -m pytest tests/test_gpu_panel.py tests/test_gpu_parity.py tests/test_gpu_sweep.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -6
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  for rep in 1 2; do
    env $envs timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
    python -c "import json;d=json.load(open('$OUT/bench_$name.json'));print('$name ms/step',round(d['ms_per_step'],4),'value',round(d['value']),'panel us',round(d['roofline']['avg_launch_us'],1),'frac',round(d['roofline']['frac'],4))"
  done
done
