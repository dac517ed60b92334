#!/bin/bash
# r02 visit N: bench (default = panel pipeline) + rocprof kernel stats + PMC traffic + MFMA/VALU counters
set -u
TAG=${1:-r02n}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1; echo "smoke rc=$?"; tail -2 "$OUT/smoke.txt"
echo "== bench"; timeout 900 python bench.py --steps 30 --warmup 5 --profile-all ${BENCH_ARGS:-} > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
cat "$OUT/bench.json"; grep "\[bench\]" "$OUT/bench.err"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/rocprof_bench.json" 2> "$OUT/rocprof.err"; echo "rocprof rc=$?"
cd "$ROOT"
for db in $(find "$OUT/prof" -name "*.db" | head -1); do python scripts/rocprof_summary.py "$db" > "$OUT/kernel_stats.md"; done
head -20 "$OUT/kernel_stats.md" | cut -c1-180
find "$OUT/prof" -name "*kernel_trace*" -size +10M -delete; find "$OUT/prof" -name "*.db" -size +20M -delete
bash scripts/gpu_pmc.sh ${TAG}_pmc 2>&1 | tail -14
PASS1="SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" PASS2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR" PASS3="GRBM_GUI_ACTIVE" PASS4="GRBM_GUI_ACTIVE" PASS5="GRBM_GUI_ACTIVE" bash scripts/gpu_counters.sh ${TAG}_ctr 2>&1 | tail -22 | cut -c1-260
