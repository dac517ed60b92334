#!/bin/bash
# One counter pass over an arbitrary command: bash scripts/gpu_counters_cmd.sh <tag> "<counters>" <cmd...>
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( cd "$ROOT" && timeout 600 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d "$OUT/pass1" -o pmc -- "$@" > "$OUT/cmd.out" 2> "$OUT/cmd.err" )
echo "rc=$?"
find "$OUT/pass1" -name "*kernel_trace*" -delete
cd "$ROOT"; python scripts/counter_summary.py "$OUT" | tee "$OUT/counters.md"
