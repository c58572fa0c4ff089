#!/usr/bin/env bash
mkdir -p gpurun_out/r3l
for a in "32 64 64 64 64 bn" "16 128 128 32 32 bn" "48 32 32 128 128 bn" "8 128 128 16 16 bn" "16 64 64 32 64"; do echo "== $a"; timeout 300 python tools/debug_sp_wgrad.py $a 4; done 2>&1 | tee gpurun_out/r3l/wgrad.log
timeout 300 python -m pytest tests/test_ops_convsp.py -m gpu -q -k big --tb=short 2>&1 | tail -40 | tee gpurun_out/r3l/ops.log
