#!/bin/bash
# r03 visit AF: ring weight-gradient kernel as four waves of 128 x 128 (BNF_RING_WAVES=4): parity tests, C2 / C3 / C4 A/B
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03af}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest (4 waves)"; BNF_RING_WAVES=4 timeout 600 python -m pytest tests/test_gpu_panel.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider > "$OUT/pytest_w4.txt" 2>&1; echo "rc=$?"; grep -E "passed|failed|^FAILED|^E  " "$OUT/pytest_w4.txt" | tail -12 | cut -c1-250
b() { timeout 180 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --profile-all 2>"$OUT/err.txt" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1))"; grep "gemm_wgrad " "$OUT/err.txt" | head -1; }
for rep in 1 2 3; do
  b w8
  BNF_RING_WAVES=4 b w4
done 2>&1 | tee "$OUT/ab_c2.txt"
cfg() { timeout 400 python scripts/bench_configs.py $2 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
  d=json.loads(l); print('$1', d['config'][:5], round(d['member_steps_per_s'],1), round(d['algorithmic_tflops'],1), round(d['final_loss_mean'],1))"; }
for c in C3 C4; do for rep in 1 2; do cfg w8 $c; BNF_RING_WAVES=4 cfg w4 $c; done; done 2>&1 | tee "$OUT/ab_cfg.txt"
