#!/usr/bin/env bash
# Winograd vs direct MFMA kernel, layer by layer (the 3x3 shapes of the unet_cct step at batch 64).  Via gpurun:
#   bash tools/sweep_wino.sh > gpurun_out/<tag>/wino_sweep.log
for shape in "64 16 16 256 256" "64 32 16 256 256" "64 32 32 128 128" "64 64 32 128 128" "64 64 64 64 64" "64 128 64 64 64" \
             "64 128 128 32 32" "64 256 128 32 32" "64 256 256 16 16" "64 16 32 128 128" "64 32 64 64 64" "64 64 128 32 32" "64 128 256 16 16"; do
  for raw in "" 1; do
    MB_RAW=$raw python tools/microbench_conv.py $shape | tail -1
    MB_RAW=$raw MB_WINO=1 python tools/microbench_conv.py $shape | tail -1
  done
done
