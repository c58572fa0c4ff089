#!/usr/bin/env bash
# which of the defensive measures the split kernels need now that the LDS coefficient tables are gone:
#   product (asm in-place MFMA + fences + full waits) | builtin (compiler MFMA chains, full waits) | builtin_nowait | asm_nowait
mkdir -p gpurun_out/r3r
for v in product builtin builtin_nowait asm_nowait; do
  lib=""; [ $v != product ] && lib=tools/exp/libwslhip_$v.so
  echo "##### $v"
  for a in "32 64 64 64 64 bn" "16 128 128 32 32 bn" "48 32 32 128 128 bn" "8 128 128 16 16 bn"; do echo -n "wgrad [$a]: "; WSL_LIB=$lib timeout 300 python tools/debug_sp_wgrad.py $a 6 2>&1 | grep -E "bad elements"; done
  for a in "8 128 128 16 32 bn" "32 64 64 64 64 bn" "8 128 128 64 32 bn" "48 32 32 128 128 bn" "16 256 256 16 16 bn"; do
    bad=0; for i in 1 2 3 4 5 6; do r=$(WSL_LIB=$lib python tools/debug_sp_case.py $a 2>&1 | grep -E "max err" | sed 's/.*bad elements \([0-9]*\) of.*/\1/'); [ "$r" != "0" ] && bad=$((bad+1)); done; echo "conv [$a]: $bad of 6 runs with bad elements"
  done
  WSL_LIB=$lib timeout 300 python tools/sweep_layers_sp.py --dec --only-sp 2>&1 | grep "^| sum"
done 2>&1 | tee gpurun_out/r3r/ab.log
