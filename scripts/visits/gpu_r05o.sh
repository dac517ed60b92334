#!/bin/bash
# r05o: f32 hidden-layer contractions on 128 x 256 tiles (BNF_F32_WIDE=0 / 1 on one binary): fp32 tests, bench fp32 / fp32_exact
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r05o}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_estimator.py tests/test_gpu_anywidth.py tests/test_gpu_sweep.py tests/test_gpu_configs.py -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^E   +(Assertion|assert)" | cut -c1-200 | head
for rep in 1 2; do for wide in 0 1; do for dt in fp32 fp32_exact; do BNF_F32_WIDE=$wide python bench.py --dtype $dt --steps 10 --warmup 2 --no-cpu-baseline --profile-all 2> "$OUT/bench_${dt}_$wide.err" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wide=$wide $dt', round(d['ms_per_step'],3), round(d['value']))"; grep -E "gemm_fwd |gemm_dgrad " "$OUT/bench_${dt}_$wide.err" | tr '\n' ' '; echo; done; done; done 2>&1 | tee "$OUT/f32_wide.txt"
