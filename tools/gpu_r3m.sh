#!/usr/bin/env bash
# in-place MFMA in wgrad too; A = product (release s_nop 7+3), B = light release (s_nop 0)
mkdir -p gpurun_out/r3m
for lib in "" tools/exp/libwslhip_norel.so; do
  echo "##### lib=${lib:-product}"
  for a in "32 64 64 64 64 bn" "16 128 128 32 32 bn" "48 32 32 128 128 bn" "8 128 128 16 16 bn" "32 32 64 64 64"; do echo "== wgrad $a"; WSL_LIB=$lib timeout 300 python tools/debug_sp_wgrad.py $a 4 2>&1 | grep -E "^run|bad elements"; done
  for a in "8 128 128 16 32 bn" "32 64 64 64 64 bn" "8 128 128 64 32 bn" "48 32 32 128 128 bn" "16 256 256 16 16 bn"; do
    bad=0; for i in 1 2 3 4 5 6; do r=$(WSL_LIB=$lib python tools/debug_sp_case.py $a 2>&1 | grep -E "max err" | sed 's/.*bad elements \([0-9]*\) of.*/\1/'); [ "$r" != "0" ] && bad=$((bad+1)); done; echo "conv [$a]: $bad of 6 runs with bad elements"
  done
done 2>&1 | tee gpurun_out/r3m/ab.log
timeout 300 python -m pytest tests/test_ops_convsp.py -m gpu -q --tb=line 2>&1 | tail -8 | tee gpurun_out/r3m/ops.log
timeout 300 python tools/sweep_layers_sp.py --dec --only-sp 2>&1 | tee gpurun_out/r3m/sweep.log
