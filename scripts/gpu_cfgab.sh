#!/bin/bash
# per-kernel table of one of scripts/bench_configs.py's configurations for a list of engine builds:
#   gpu_cfgab.sh <tag> <C3|C4|C5> "<pytest -k expression>" name[:ENV=V] ...     (ab/libbnf_<name>.so)
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; CFG=$2; KEXPR=$3; shift 3; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
declare -A FULL=([C3]="C3/8 air_quality-like VI" [C4]="C4/8 synthetic minibatch MLE" [C5]="C5/8 wind-like MAP (bf16)")
first=${1%%:*}
[ -n "$KEXPR" ] && BNF_LIB=$ROOT/ab/libbnf_$first.so timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "$KEXPR" 2>&1 | grep -E "passed|failed|error" | tail -3 | tee "$OUT/pytest.txt"
for rep in 1 2; do
  for spec in "$@"; do
    v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
    echo "== $spec"
    env $envs BNF_LIB=$ROOT/ab/libbnf_$v.so timeout 300 python scripts/bench_configs.py $CFG 2>/dev/null | tail -1 | cut -c100-200
    env $envs BNF_LIB=$ROOT/ab/libbnf_$v.so timeout 200 python scripts/profile_config.py "${FULL[$CFG]}" 2>/dev/null
  done
done 2>&1 | tee "$OUT/ab.txt"
