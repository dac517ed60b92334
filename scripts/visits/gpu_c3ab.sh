#!/bin/bash
# C3/8 (VI) per-kernel table for a list of engine builds: gpu_c3ab.sh <tag> name[:ENV=V] ...   (ab/libbnf_<name>.so)
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
first=${1%%:*}
BNF_LIB=$ROOT/ab/libbnf_$first.so timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "vi or VI" 2>&1 | grep -E "passed|failed|error" | tail -3 | tee "$OUT/pytest_vi.txt"
for rep in 1 2; do
  for spec in "$@"; do
    v=${spec%%:*}; envs=""; [ "$spec" != "$v" ] && envs=${spec#*:}
    echo "== $spec"
    env $envs BNF_LIB=$ROOT/ab/libbnf_$v.so timeout 300 python scripts/bench_configs.py C3 2>/dev/null | tail -1 | cut -c100-200
    env $envs BNF_LIB=$ROOT/ab/libbnf_$v.so timeout 200 python scripts/profile_config.py "C3/8 air_quality-like VI" 2>/dev/null | grep -v "featurize\|wgrad_l0"
  done
done 2>&1 | tee "$OUT/c3_ab.txt"
