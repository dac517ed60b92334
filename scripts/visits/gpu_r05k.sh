#!/bin/bash
# r05k: fragment-major copies for the weight-gradient kernels (bf16): tests, then the same binary with BNF_FM=0 / 1 alternated
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r05k}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest"; timeout 1200 python -m pytest tests/test_gpu_panel.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_anywidth.py tests/test_gpu_fp8.py -x -q -p no:cacheprovider 2>&1 | tail -6 | tee "$OUT/pytest.txt"
echo "== C2 BNF_FM=0 / 1"; for rep in 1 2 3; do for fm in 0 1; do BNF_FM=$fm python bench.py --steps 30 --warmup 3 --no-cpu-baseline --profile-all 2> "$OUT/bench_fm$fm.err" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fm=$fm', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1))"; grep -E "gemm_wgrad " "$OUT/bench_fm$fm.err" | head -2 | tr '\n' ' '; echo; done; done 2>&1 | tee "$OUT/ab_c2.txt"
grep "\[bench\]" "$OUT/bench_fm1.err" | head -8
echo "== configs"; for c in C3 C4 C5; do for fm in 0 1; do echo -n "$c fm=$fm "; BNF_FM=$fm timeout 600 python scripts/bench_configs.py $c 2>/dev/null | tail -1 | cut -c1-230; done; done | tee "$OUT/ab_configs.txt"
