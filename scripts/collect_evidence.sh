#!/bin/bash
# copies what visits/gpu_r04z.sh <tag> merged into gpurun_out/ to profiles/<tag>_* and stamps the two counter files
set -e
TAG=${1:-r04z}; G=gpurun_out; P=profiles
for f in bench.json bench_200_steps.txt bench_fp32.json bench_fp32_hip_events.txt bench_hip_events.txt bench_repeat.txt c1_step_time.txt config_profiles.txt configs_bench.jsonl phase_clocks.txt pytest_gpu.txt shuffle_cost.txt smoke.txt ab_vs_round3.txt; do cp $G/$TAG/$f $P/${TAG}_$f; done
cp $G/$TAG/kernel_stats.md $P/${TAG}_bench_kernel_stats.md
cp $G/${TAG}_pmc/traffic.md $P/${TAG}_pmc_hbm_traffic.md; cp $G/${TAG}_pmc_C3/traffic.md $P/${TAG}_pmc_hbm_traffic_C3.md; cp $G/${TAG}_pmc_C5/traffic.md $P/${TAG}_pmc_hbm_traffic_C5.md
cp $G/${TAG}_ctr/counters.md $P/${TAG}_mfma_valu_counters.md
python scripts/stamp_profiles.py $G/${TAG}_pmc/traffic.json $G/$TAG/sq_counters.json
