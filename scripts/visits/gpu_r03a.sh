#!/bin/bash
# r03 visit A: GPU tests, same-box A/B of the lean-activation panel kernel, cold-start flake hunt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03a}; mkdir -p "$OUT"; cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > "$OUT/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -15 "$OUT/pytest.txt"
echo "== A/B"; STEPS=30 bash scripts/gpu_ab.sh 2>&1 | tee "$OUT/ab.txt"
echo "== cold hunt"
python scripts/flake_hunt_cold.py > "$OUT/cold_0.txt" 2>&1   # fills the oracle cache
N=${COLD_N:-25}
for w in 1 2 3 4 5 6 7 8; do
  ( for i in $(seq $N); do HUNT_ORDER=$([ $((i % 2)) = 0 ] && echo shuffle || echo test) python scripts/flake_hunt_cold.py; done > "$OUT/cold_$w.txt" 2>&1 ) &
done
wait
cat "$OUT"/cold_*.txt | grep -c "^cold pid"; cat "$OUT"/cold_*.txt | grep -E "DEVIATION|Error|error" | head -20
cat "$OUT"/cold_*.txt | awk '/^cold pid/{e+=$4; b+=$6} END{print "cold total evaluations", e, "deviating", b}'
