#!/bin/bash
# C2 bench (30 steps) for a list of engine builds, alternated three times: gpu_c2ab.sh <tag> "<pytest -k>" name[:ENV=V] ...
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; KEXPR=$2; shift 2; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
first=${1%%:*}
[ -n "$KEXPR" ] && BNF_LIB=$ROOT/ab/libbnf_$first.so timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "$KEXPR" 2>&1 | grep -E "passed|failed|error" | tail -3 | tee "$OUT/pytest.txt"
for rep in 1 2 3; do
  for spec in "$@"; do
    v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
    env $envs BNF_LIB=$ROOT/ab/libbnf_$v.so timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$spec', 'ms/step', round(d['ms_per_step'],4), 'panel_us', round(d['roofline']['avg_launch_us'],1), 'frac', round(d['roofline']['frac'],4))"
  done
done 2>&1 | tee "$OUT/ab.txt"
