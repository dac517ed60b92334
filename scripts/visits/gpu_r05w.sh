#!/bin/bash
# r05w: split-K of the layer-0 weight-gradient stream (gemm_tn_skinny / skinny8): BNF_SKINNY_SPLITK = 1 .. 8 at C2, bf16 and fp8
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r05w}; mkdir -p "$OUT"; cd "$ROOT"
for dt in fp8 bf16; do for sk in 0 1 2 3 4 6 8; do
  if [ $sk = 0 ]; then unset BNF_SKINNY_SPLITK; else export BNF_SKINNY_SPLITK=$sk; fi
  python bench.py --dtype $dt --steps 20 --warmup 3 --no-cpu-baseline --profile-all 2> "$OUT/b.err" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dt splitk=$sk', round(d['ms_per_step'],4))" | tr '\n' ' '; grep -E "gemm_wgrad_l0|adam_map" "$OUT/b.err" | tr '\n' ' '; echo
done; done 2>&1 | tee "$OUT/skinny_splitk.txt"
