#!/bin/bash
# two-group panel kernel: parity + bench for a few stagger values
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
echo "== pytest panel (BNF_PANEL2=${P2:-1})"; BNF_PANEL2=${P2:-1} timeout 600 python -m pytest tests/test_gpu_panel.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -4
for sg in ${STAGGERS:-0}; do
  echo "== bench BNF_PANEL2=${P2:-1} stagger=$sg"
  BNF_PANEL2=${P2:-1} BNF_PANEL_STAGGER=$sg timeout 300 python bench.py --steps 20 --warmup 3 --profile-all --no-cpu-baseline 2>/tmp/err.txt | python -c "import sys,json;d=json.loads(sys.stdin.read());print('ms/step',round(d['ms_per_step'],3),'value',round(d['value']),'panel us',round(d['roofline']['avg_launch_us'],1))"
  grep "\[bench\] panel" /tmp/err.txt
done
BNF_PANEL2=${P2:-1} BNF_LIB=$ROOT/ab/libbnf_ablate.so BNF_PHASE_PROF=panel_fwd_bwd timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep "phase clocks"
