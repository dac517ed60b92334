#!/bin/bash
# r05l: the fp32 engine with its contractions on split-bf16 MFMAs (BNF_F32_SPLIT = 2 / 3 pieces): fp32 parity tests + goldens
# under each build, bench --dtype fp32
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r05l}; mkdir -p "$OUT"; cd "$ROOT"
for v in cur f32s3 f32s6; do
  if [ $v = cur ]; then unset BNF_LIB; else export BNF_LIB=$ROOT/ab/libbnf_$v.so; fi
  echo "== $v: fp32 tests"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_estimator.py tests/test_gpu_anywidth.py tests/test_gpu_sweep.py -q -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|^E  " | cut -c1-250 | head -8
  echo "== $v: bench fp32"; python bench.py --dtype fp32 --steps 10 --warmup 2 --no-cpu-baseline --profile-all 2> "$OUT/bench_$v.err" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), round(d['value']), d['final_loss_mean'])"; grep "\[bench\]" "$OUT/bench_$v.err" | head -8
  echo "== $v: smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
done 2>&1 | tee "$OUT/f32_split.txt"
