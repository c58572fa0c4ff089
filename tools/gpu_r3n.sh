#!/usr/bin/env bash
mkdir -p gpurun_out/r3n
for lib in "" tools/exp/libwslhip_longrel.so; do
  echo "##### lib=${lib:-product}"
  for a in "32 64 64 64 64 bn" "16 128 128 32 32 bn" "48 32 32 128 128 bn" "8 128 128 16 16 bn" "32 32 64 64 64"; do echo "== wgrad $a"; WSL_LIB=$lib timeout 300 python tools/debug_sp_wgrad.py $a 4 2>&1 | grep -E "^run|bad elements|co:|ci:|ky:"; done
done 2>&1 | tee gpurun_out/r3n/ab.log
timeout 300 python -m pytest tests/test_ops_convsp.py -m gpu -q --tb=line 2>&1 | tail -8 | tee gpurun_out/r3n/ops.log
