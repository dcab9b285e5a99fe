#!/bin/bash
# Decode rate by context (Llama-2-7B Q4, F16 and Q8 KV cache): tools/ctx_prof.py at 256 .. 16384 keys (GPU box, repo root)
for kv in "" q8; do for c in 256 600 1024 2048 4096 8192 16384; do python tools/ctx_prof.py $c $kv 2>&1 | grep "^ctx" | sed "s/  ids.*//"; done; done
