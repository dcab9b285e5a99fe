#!/bin/bash
# W1 / W3 as two 512-thread workgroups per CU (lib_variants/glu512r3*, option rpw_ffn = 2) against the default 1024-thread launch
for v in glu512r3 glu512r3np2; do
  echo "== $v"
  IFA_LIB=$PWD/lib_variants/$v/libinferflow_amd.so timeout 300 python tools/ab_option.py rpw_ffn --values 1,2 --steps 20 --prompt 21 --kernels 2>&1 | grep -v amdgpu
done
echo "== default library"
timeout 300 python tools/ab_option.py rpw_ffn --values 0,2 --steps 20 --prompt 21 --kernels 2>&1 | grep -v amdgpu
