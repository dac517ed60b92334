#!/bin/bash
# r02 visit A: ceilings probe + C3/C4/C5 full workloads (tests, throughput, rocprof kernel stats, C5 counters)
set -u
TAG=${1:-r02a}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== probe"; [ -x scripts/probes/panel_probe ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o scripts/probes/panel_probe scripts/probes/panel_probe.hip; timeout 300 scripts/probes/panel_probe > "$OUT/panel_probe.txt" 2>&1; echo "probe rc=$?"; cat "$OUT/panel_probe.txt"
echo "== pytest configs"; timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider -x > "$OUT/pytest_configs.txt" 2>&1; echo "pytest rc=$?"
tail -15 "$OUT/pytest_configs.txt"
cd /tmp && export TMPDIR=/tmp
for c in C3 C4 C5; do
  echo "== bench_configs $c (rocprofv3 kernel trace)"
  timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$c" -o cfg -- python "$ROOT/scripts/bench_configs.py" $c > "$OUT/bench_$c.json" 2> "$OUT/bench_$c.err"; echo "rc=$?"
  cat "$OUT/bench_$c.json"; tail -3 "$OUT/bench_$c.err"
  (cd "$ROOT" && for db in $(find "$OUT/prof_$c" -name "*.db" | head -1); do python scripts/rocprof_summary.py "$db" > "$OUT/kernel_stats_$c.md"; done)
  find "$OUT/prof_$c" -name "*kernel_trace*" -size +10M -delete
  find "$OUT/prof_$c" -name "*.db" -size +20M -delete
done
echo "== C5 counters (MFMA / VALU busy)"
timeout 600 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pass1" -o pmc -- \
   python "$ROOT/scripts/bench_configs.py" C5 > "$OUT/pmc_C5.json" 2> "$OUT/pmc_C5.err"; echo "pmc rc=$?"
find "$OUT/pass1" -name "*kernel_trace*" -delete
cd "$ROOT"
python scripts/counter_summary.py "$OUT" > "$OUT/counters_C5.md"; head -12 "$OUT/counters_C5.md" | cut -c1-400
du -sh "$OUT"
