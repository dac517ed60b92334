#!/bin/bash
# perf experiments: bench kernel table under BNF_ABLATE masks
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; OUT=$ROOT/gpurun_out/${1:-abl}; mkdir -p "$OUT"
for m in ${ABL_MASKS:-0 1 2 3 4 7 16 32 39}; do
  echo "=== BNF_ABLATE=$m"
  BNF_ABLATE=$m timeout 300 python bench.py --steps 5 --warmup 3 --profile-all --no-cpu-baseline 2>&1 >/dev/null | grep "\[bench\]" | tee "$OUT/abl_$m.txt"
done
