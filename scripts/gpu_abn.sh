#!/bin/bash
# Same-box A/B/... of several builds of the engine, alternated so that box-to-box and thermal drift cancel.
# VARIANTS="name:path name2:path2 new:" (empty path = the in-tree build); STEPS, REPS.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
for rep in $(seq ${REPS:-3}); do
  for v in ${VARIANTS:-prev:ab/libbnf_prev.so new:}; do
    name=${v%%:*}; path=${v#*:}
    if [ -n "$path" ]; then export BNF_LIB=$ROOT/$path; else unset BNF_LIB; fi
    python bench.py --steps ${STEPS:-30} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],4), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1))"
  done
done
unset BNF_LIB
