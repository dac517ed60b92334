#!/bin/bash
# r04p: VI sampler -- one Philox call per (sample, quad of parameters) and sample + pack in one kernel (k_vi_sample_pack)
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04p; mkdir -p "$OUT"; cd "$ROOT"
CUR=${1:-$ROOT/ab/libbnf_cur.so}
BNF_LIB=$CUR timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -k "vi or VI" 2>&1 | tail -8 | tee "$OUT/pytest_vi.txt"
for rep in 1 2; do
  for v in r04z cur; do
    lib=$ROOT/ab/libbnf_$v.so; [ $v = cur ] && lib=$CUR
    echo "== $v"; BNF_LIB=$lib timeout 300 python scripts/bench_configs.py C3 2>/dev/null | tail -1 | cut -c1-200
    BNF_LIB=$lib timeout 200 python scripts/profile_config.py "C3/8 air_quality-like VI" 2>/dev/null
  done
done 2>&1 | tee "$OUT/c3_ab.txt"
echo "== unfused with the new stream"; BNF_VI_SAMPLE_PACK=0 BNF_LIB=$CUR timeout 200 python scripts/profile_config.py "C3/8 air_quality-like VI" 2>/dev/null | tee "$OUT/c3_unfused.txt"
