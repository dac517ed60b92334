#!/bin/bash
# evidence visit (parametrized by tag; was scripts/visits/gpu_r05z.sh): -m gpu suite, smoke, bench (+cpu_baseline), rocprof kernel stats, PMC HBM traffic, SQ counters
# (-> profiles/sq_counters.json, pmc_traffic.json), the other configurations in bf16 AND fp8 with per-kernel tables, phase
# clocks, fp32 / fp8 bench lines, PMC traffic of C5/8 in both dtypes, same-box A/B against the round-4 kernels
set -u; ulimit -c 0
TAG=${1:-r06z}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" > "$OUT/pytest_gpu.txt"; tail -3 "$OUT/pytest_gpu.txt"
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1; echo "smoke rc=$?"; tail -3 "$OUT/smoke.txt"
echo "== bench"; timeout 900 python bench.py --steps 30 --warmup 5 --profile-all > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
cat "$OUT/bench.json"; grep "\[bench\]" "$OUT/bench.err" | tee "$OUT/bench_hip_events.txt"
echo "== bench x3 (20 steps, the driver's command)"; for i in 1 2 3; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['value']), round(d['roofline']['frac'],4), round(d['roofline']['avg_launch_us'],1))"; done | tee "$OUT/bench_repeat.txt"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/rocprof_bench.json" 2> "$OUT/rocprof.err"; echo "rocprof rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof8" -o bench -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --dtype fp8 > "$OUT/rocprof_bench_fp8.json" 2> "$OUT/rocprof8.err"; echo "rocprof fp8 rc=$?"
cd "$ROOT"
for db in $(find "$OUT/prof" -name "*.db" | head -1); do python scripts/rocprof_summary.py "$db" > "$OUT/kernel_stats.md"; done
for db in $(find "$OUT/prof8" -name "*.db" | head -1); do python scripts/rocprof_summary.py "$db" > "$OUT/kernel_stats_fp8.md"; done
head -14 "$OUT/kernel_stats.md" | cut -c1-170; head -12 "$OUT/kernel_stats_fp8.md" | cut -c1-170
find "$OUT/prof" "$OUT/prof8" -name "*kernel_trace*" -size +10M -delete; find "$OUT/prof" "$OUT/prof8" -name "*.db" -size +20M -delete
bash scripts/gpu_pmc.sh ${TAG}_pmc 2>&1 | tail -12
PASS1="SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" PASS2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR" PASS3="GRBM_GUI_ACTIVE" PASS4="GRBM_GUI_ACTIVE" PASS5="GRBM_GUI_ACTIVE" bash scripts/gpu_counters.sh ${TAG}_ctr 2>&1 | tail -22 | cut -c1-200
python scripts/counter_summary.py "$OUT/../${TAG}_ctr" "$OUT/sq_counters.json" > /dev/null
echo "== configs (bf16 / fp8)"; for c in C3 C4 C5; do for dt in bf16 fp8; do BNF_BENCH_DTYPE=$dt timeout 600 python scripts/bench_configs.py $c 2>/dev/null | tail -1 | sed "s/^{/{\"dtype\": \"$dt\", /"; done; done > "$OUT/configs_bench.jsonl"; cut -c1-330 "$OUT/configs_bench.jsonl"
for dt in bf16 fp8; do for c in "C3/8 air_quality-like VI" "C4/8 synthetic minibatch MLE" "C5/8 wind-like MAP (bf16)"; do echo "== $c [$dt]"; BNF_BENCH_DTYPE=$dt timeout 200 python scripts/profile_config.py "$c" 2>/dev/null; done; done > "$OUT/config_profiles.txt"
echo "== phase clocks"; LIBS=ablate THREADS="0 448" bash scripts/gpu_clk.sh ${TAG}_clk 2>&1 | tee "$OUT/phase_clocks.txt"
echo "== bench fp32_split"; timeout 600 python bench.py --dtype fp32_split --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-400 | tee "$OUT/bench_fp32_split.json"
echo "== bench fp32"; timeout 600 python bench.py --dtype fp32 --steps 10 --warmup 2 --no-cpu-baseline --profile-all > "$OUT/bench_fp32.json" 2> "$OUT/bench_fp32.err"; cut -c1-400 "$OUT/bench_fp32.json"; grep "\[bench\]" "$OUT/bench_fp32.err" > "$OUT/bench_fp32_hip_events.txt"
echo "== bench fp8"; timeout 600 python bench.py --dtype fp8 --steps 30 --warmup 5 --no-cpu-baseline --profile-all > "$OUT/bench_fp8.json" 2> "$OUT/bench_fp8.err"; cut -c1-600 "$OUT/bench_fp8.json"; grep "\[bench\]" "$OUT/bench_fp8.err" | tee "$OUT/bench_fp8_hip_events.txt"
echo "== C1 step time"; timeout 300 python scripts/c1_step_time.py 2>/dev/null | tee "$OUT/c1_step_time.txt"
echo "== PMC traffic of C5 (bf16, fp8)"; bash scripts/gpu_pmc_cfg.sh ${TAG}_pmc C5 2>&1 | cut -c1-200 | tail -14; BNF_BENCH_DTYPE=fp8 bash scripts/gpu_pmc_cfg.sh ${TAG}_pmc8 C5 2>&1 | cut -c1-200 | tail -14
echo "== PMC traffic of C2 fp8"; cd /tmp; for c in FETCH_SIZE WRITE_SIZE; do timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$ROOT/gpurun_out/${TAG}_pmc8_C2/$c" -o pmc -- python "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --dtype fp8 > /dev/null 2>&1; done; cd "$ROOT"; python scripts/pmc_summary.py "gpurun_out/${TAG}_pmc8_C2" | tee "gpurun_out/${TAG}_pmc8_C2/traffic.md" | cut -c1-200 | head -10; find "gpurun_out/${TAG}_pmc8_C2" -name "*.csv" -size +5M -delete
echo "== 200-step runs"; for dt in bf16 fp8; do python bench.py --steps 200 --warmup 10 --no-cpu-baseline --dtype $dt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dt', round(d['ms_per_step'],4), round(d['value']), round(d['roofline']['frac'],4), round(d['roofline']['avg_launch_us'],1))"; done | tee "$OUT/bench_200_steps.txt"
echo "== same-box A/B against the round-4 kernels"; for rep in 1 2 3; do for v in prev cur; do if [ $v = prev ]; then export BNF_LIB=$ROOT/ab/libbnf_prev.so; else unset BNF_LIB; fi; python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],4))"; done; done | tee "$OUT/ab_vs_round4.txt"; unset BNF_LIB
du -sh gpurun_out
