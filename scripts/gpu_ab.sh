#!/bin/bash
# Same-box A/B of two builds of the engine: ab/libbnf_prev.so (BNF_LIB) vs the in-tree build,
# alternated so that box-to-box and thermal drift cancel.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
for rep in 1 2 3; do
  for v in prev new; do
    if [ $v = prev ]; then export BNF_LIB=$ROOT/ab/libbnf_prev.so; else unset BNF_LIB; fi
    python bench.py --steps ${STEPS:-30} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1))"
  done
done
