#!/usr/bin/env bash
mkdir -p gpurun_out/r3o
for a in "32 64 64 64 64 bn" "32 64 64 64 64 pre" "32 64 64 64 64" "16 128 128 32 32 bn" "16 128 128 32 32 pre" "16 128 128 32 32"; do echo "== wgrad $a"; timeout 300 python tools/debug_sp_wgrad.py $a 4 2>&1 | grep -E "^run|bad elements"; done 2>&1 | tee gpurun_out/r3o/ab.log
