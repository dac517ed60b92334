#!/bin/bash
# r04 evidence visit: -m gpu suite, smoke, bench (+cpu_baseline), rocprof kernel stats, PMC HBM traffic, SQ counters
# (-> profiles/sq_counters.json, pmc_traffic.json), per-config benches + per-kernel tables of C3 / C4 / C5, phase clocks
set -u; ulimit -c 0
TAG=${1:-r04z}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" > "$OUT/pytest_gpu.txt"; tail -3 "$OUT/pytest_gpu.txt"
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1; echo "smoke rc=$?"; tail -2 "$OUT/smoke.txt"
echo "== bench"; timeout 900 python bench.py --steps 30 --warmup 5 --profile-all ${BENCH_ARGS:-} > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
cat "$OUT/bench.json"; grep "\[bench\]" "$OUT/bench.err" | tee "$OUT/bench_hip_events.txt"
echo "== bench x3 (20 steps, the driver's command)"; for i in 1 2 3; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['value']), round(d['roofline']['frac'],4), round(d['roofline']['avg_launch_us'],1))"; done | tee "$OUT/bench_repeat.txt"
echo "== bench without the device preheat (20 steps)"; python bench.py --steps 20 --warmup 3 --no-cpu-baseline --preheat-ms 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['value']), round(d['roofline']['frac'],4), round(d['roofline']['avg_launch_us'],1), 'preheat 0')" | tee -a "$OUT/bench_repeat.txt"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/rocprof_bench.json" 2> "$OUT/rocprof.err"; echo "rocprof rc=$?"
cd "$ROOT"
for db in $(find "$OUT/prof" -name "*.db" | head -1); do python scripts/rocprof_summary.py "$db" > "$OUT/kernel_stats.md"; done
head -14 "$OUT/kernel_stats.md" | cut -c1-170
find "$OUT/prof" -name "*kernel_trace*" -size +10M -delete; find "$OUT/prof" -name "*.db" -size +20M -delete
bash scripts/gpu_pmc.sh ${TAG}_pmc 2>&1 | tail -12
PASS1="SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" PASS2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR" PASS3="GRBM_GUI_ACTIVE" PASS4="GRBM_GUI_ACTIVE" PASS5="GRBM_GUI_ACTIVE" bash scripts/gpu_counters.sh ${TAG}_ctr 2>&1 | tail -22 | cut -c1-200
python scripts/counter_summary.py "$OUT/../${TAG}_ctr" "$OUT/sq_counters.json" > /dev/null
echo "== configs"; for c in C3 C4 C5; do timeout 600 python scripts/bench_configs.py $c 2>/dev/null | tail -1; done > "$OUT/configs_bench.jsonl"; cut -c1-330 "$OUT/configs_bench.jsonl"
for c in "C3/8 air_quality-like VI" "C4/8 synthetic minibatch MLE" "C5/8 wind-like MAP (bf16)"; do echo "== $c"; timeout 200 python scripts/profile_config.py "$c" 2>/dev/null; done > "$OUT/config_profiles.txt"
echo "== phase clocks"; THREADS="0 448" bash scripts/gpu_phase_clocks.sh 2>&1 | tee "$OUT/phase_clocks.txt"
echo "== bench fp32"; timeout 600 python bench.py --dtype fp32 --steps 10 --warmup 2 --no-cpu-baseline --profile-all > "$OUT/bench_fp32.json" 2> "$OUT/bench_fp32.err"; cut -c1-400 "$OUT/bench_fp32.json"; grep "\[bench\]" "$OUT/bench_fp32.err" > "$OUT/bench_fp32_hip_events.txt"
echo "== C1 step time"; timeout 300 python scripts/c1_step_time.py 2>/dev/null | tee "$OUT/c1_step_time.txt"
echo "== C1 without the LDS feature panel"; BNF_PANEL_NO_H0L=1 timeout 300 python scripts/c1_step_time.py 2>/dev/null | grep bf16 | tee -a "$OUT/c1_step_time.txt"
echo "== shuffle draw cost"; timeout 600 python scripts/shuffle_cost.py 2>/dev/null | tee "$OUT/shuffle_cost.txt"
echo "== PMC traffic of C3 / C5"; bash scripts/gpu_pmc_cfg.sh ${TAG}_pmc C3 C5 2>&1 | cut -c1-200 | tail -28
echo "== 200-step runs"; for i in 1 2; do python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['value']), round(d['roofline']['frac'],4), round(d['roofline']['avg_launch_us'],1))"; done | tee "$OUT/bench_200_steps.txt"
echo "== same-box A/B against the round-3 kernels"; for rep in 1 2 3; do for v in prev cur; do if [ $v = prev ]; then export BNF_LIB=$ROOT/ab/libbnf_prev.so; else unset BNF_LIB; fi; python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],4))"; done; done | tee "$OUT/ab_vs_round3.txt"; unset BNF_LIB
