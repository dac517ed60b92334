#!/bin/bash
# r04g: order of the weight-gradient kernels after the panel kernel (layer 0 first, as in rounds 1-3, vs last)
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04g; mkdir -p "$OUT"; cd "$ROOT"
one() {  # name [env...]
  local name=$1; shift
  env "$@" timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-all 2> "$OUT/bench_$name.err" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name', 'ms/step', round(d['ms_per_step'],4), 'panel_us', round(d['roofline']['avg_launch_us'],1), 'frac', round(d['roofline']['frac'],4))"
  grep "\[bench\]" "$OUT/bench_$name.err" | head -4 | awk '{printf "   %s %s us", $2, $4} END {print ""}'
}
for rep in 1 2 3; do
  one fwd BNF_WGRAD_ORDER=fwd
  one rev X=1
done 2>&1 | tee "$OUT/ab.txt"
for c in C3 C5; do for o in fwd rev; do echo -n "$c $o "; BNF_WGRAD_ORDER=$o timeout 600 python scripts/bench_configs.py $c 2>/dev/null | tail -1 | cut -c1-200; done; done | tee "$OUT/configs.txt"
timeout 600 python -m pytest tests/test_gpu_panel.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
