#!/bin/bash
# measurement: variants of k_gemm_mid (ring depth, ablations), one library per variant, per-kernel rocprofv3 averages of a 128-token prefill
# (run on the build box first: builds lib_variants/mid_<name>.so; then on the GPU box with RUN=1)
set -e
R=$(cd $(dirname $0)/.. && pwd)
CS=$R/inferflow_amd/csrc
VARIANTS=${VARIANTS:-"ns3:-DIFA_MID_NS=3 ns4:-DIFA_MID_NS=4 ns5:-DIFA_MID_NS=5 nox:-DIFA_MID_ABL=1 nomfma:-DIFA_MID_ABL=2 nodq:-DIFA_MID_ABL=3 nobar:-DIFA_MID_ABL=4"}
if [ -z "$RUN" ]; then
  mkdir -p $R/lib_variants
  for v in $VARIANTS; do
    name=${v%%:*}; flag=$(echo ${v#*:} | tr ',' ' ')
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=8 $flag -I $R/include -I $CS -c $CS/ifa_gemm_mid.hip -o /tmp/mid_$name.o
    objs=$(ls $R/inferflow_amd/lib/obj/*.o | grep -v ifa_gemm_mid.hip.o)
    hipcc --offload-arch=gfx950 -shared -fPIC -o $R/lib_variants/mid_$name.so $objs /tmp/mid_$name.o -L/opt/rocm/lib -lrccl
    echo built $name
  done
  exit 0
fi
cd /tmp && export TMPDIR=/tmp
for v in $VARIANTS; do
  name=${v%%:*}
  rm -rf /tmp/pp
  IFA_LIB=$R/lib_variants/mid_$name.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o st -- python $R/tools/prefill_steps.py llama2_7b ${T:-128} 5 > /tmp/pp.log 2>&1 || true
  f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
  echo "== $name: $(grep prefill /tmp/pp.log | tail -1)"; grep k_gemm_mid $f | cut -d, -f1-4,6,7
done
