#!/bin/bash
# A / B of the preloaded weight pointers (round 5): the default library against lib_variants/gemvonly (k_dec_gemv preloads only, before k_dec_qkv_attn got them), alternating
for r in 1 2; do
  for v in default gemvonly; do
    if [ $v = default ]; then unset IFA_LIB; else export IFA_LIB=$PWD/lib_variants/$v/libinferflow_amd.so; fi
    echo "== $v (round $r)"
    timeout 300 python tools/ab_option.py graph --values 1 --steps 20 --prompt 21 --rounds 3 --kernels 2>&1 | grep -v amdgpu
  done
done
unset IFA_LIB
echo "== phase stamps, default library"
timeout 200 python tools/trace_kernels.py 2>&1 | grep -v amdgpu
echo "== phase stamps, gemvonly"
IFA_LIB=$PWD/lib_variants/gemvonly/libinferflow_amd.so timeout 200 python tools/trace_kernels.py 2>&1 | grep -v amdgpu
