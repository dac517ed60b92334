#!/bin/bash
# (re)creates tests/golden/ from the reference's own test data -- runs only where /root/reference exists
set -e
SRC=${1:-/root/reference/tests/test_data}; DST=$(dirname "$0")/../tests/golden; mkdir -p "$DST"
for f in chickenpox.8.train.csv chickenpox.8.test.csv bnf-map.chickenpox.8.mini.pred.csv bnf-mle.chickenpox.8.mini.pred.csv bnf-vi.chickenpox.8.mini.pred.csv; do
  cp "$SRC/$f" "$DST/$f"
done
