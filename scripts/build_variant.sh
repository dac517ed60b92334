#!/bin/bash
# A/B builds of the engine: scripts/build_variant.sh <name> [-DKNOB=V ...]  ->  ab/libbnf_<name>.so (+ /tmp/v/<name>.s,
# register / scratch table of the panel kernels).  ab/ travels to the GPU box; BNF_LIB selects a build at run time.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); NAME=$1; shift
mkdir -p "$ROOT/ab" /tmp/v/$NAME
cd /tmp/v/$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -Wno-unused-value "$@" \
  -save-temps=obj "$ROOT/bayesnf_amd/csrc/bnf_api.hip" -o /tmp/v/$NAME/libbnf.so 2>&1 | grep -E "error" || true
cp /tmp/v/$NAME/libbnf.so "$ROOT/ab/libbnf_$NAME.so"
cp /tmp/v/$NAME/bnf_api-hip-amdgcn-amd-amdhsa-gfx950.s /tmp/v/$NAME.s
python "$ROOT/scripts/isa_report.py" /tmp/v/$NAME.s '<8, 4, true, false, 1, 64>' | grep -E "k_panel|valu" | grep -E "8, 4, true|valu"
