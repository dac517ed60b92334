#!/bin/bash
# same-box alternation of ENVIRONMENT switches of the in-tree build: gpu_envab.sh <tag> "<pytest -k or empty>" "name:ENV=V ENV2=V" ...
# (C2 bench, 30 steps, three alternations; then optionally the C5/8 share: CFGS="C5 C3")
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; KEXPR=$2; shift 2; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
[ -n "$KEXPR" ] && timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -k "$KEXPR" 2>&1 | grep -E "passed|failed|error|FAILED" | tail -8 | tee "$OUT/pytest.txt"
for rep in 1 2 3; do
  for spec in "$@"; do
    v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
    env $envs timeout 200 python bench.py --steps ${STEPS:-30} --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$v', 'ms/step', round(d['ms_per_step'],4), 'panel_us', round(d['roofline']['avg_launch_us'],1), 'frac', round(d['roofline']['frac'],4))"
  done
done 2>&1 | tee "$OUT/ab.txt"
for cfg in ${CFGS:-}; do
  for rep in 1 2; do
    for spec in "$@"; do
      v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
      echo "== $cfg $v"; env $envs timeout 300 python scripts/bench_configs.py $cfg 2>/dev/null | tail -1 | cut -c1-260
    done
  done
done 2>&1 | tee "$OUT/cfg_ab.txt"
