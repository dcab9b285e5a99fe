#!/bin/bash
# compile one persistent-kernel translation unit with -save-temps and print resource usage; the asm stays in /tmp/psasm
# usage: tools/ps_asm.sh [q4b32|q3h]
set -e
FMT=${1:-q4b32}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/psasm && cd /tmp/psasm
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=8 \
  -I "$ROOT/include" -I "$ROOT/inferflow_amd/csrc" -c "$ROOT/inferflow_amd/csrc/experimental/ifa_dpersist_$FMT.hip" -o /tmp/psasm/ps_$FMT.o -save-temps=obj 2>&1 | grep -E "error|warning: v" -A3 || true
S=/tmp/psasm/ifa_dpersist_$FMT-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "^_ZN3ifa13k_dec_persist.*:|; codeLenInByte|; ScratchSize|; NumVgprs:|; NumSgprs|sgpr_spill|vgpr_spill" "$S" | sed 's/ *;.*@.*//' | paste - - - - - - - | awk '{print}' | cut -c1-220 || true
