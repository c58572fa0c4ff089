#!/usr/bin/env bash
mkdir -p gpurun_out/r3p
for a in "32 64 64 64 64 bn" "16 128 128 32 32 bn" "48 32 32 128 128 bn" "8 128 128 16 16 bn"; do echo "== wgrad $a"; timeout 300 python tools/debug_sp_wgrad.py $a 4 2>&1 | grep -E "^run|bad elements"; done 2>&1 | tee gpurun_out/r3p/ab.log
