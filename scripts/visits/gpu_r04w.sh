#!/bin/bash
# r04w: several panels per workgroup (-DBNF_PANEL_LOOP=1 build, BNF_PANEL_PPW at run time)
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04w; mkdir -p "$OUT"; cd "$ROOT"
BNF_PANEL_PPW=2 BNF_LIB=$ROOT/ab/libbnf_loop.so timeout 900 python -m pytest tests/test_gpu_panel.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -2
one() {  # label lib ppw
  BNF_PANEL_PPW=$3 BNF_LIB=$ROOT/ab/libbnf_$2.so timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', 'ms/step', round(d['ms_per_step'],4), 'panel_us', round(d['roofline']['avg_launch_us'],1))"
}
for rep in 1 2; do
  one cur cur 1
  for p in 1 2 4 5 10 20; do one loop_ppw$p loop $p; done
done 2>&1 | tee "$OUT/ab.txt"
