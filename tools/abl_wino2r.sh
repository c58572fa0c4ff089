#!/usr/bin/env bash
# phase ablations of conv_wino2r_kernel (experiments build, plain source = a data-gradient launch): WSL_WINO2R_ABLATE bits
# 1 no MFMAs, 2 no DMA after the first chunk, 4 no epilogue, 8 no operand reads from LDS
for a in ${ABLS:-0 4 2 6 7 14}; do echo "== WSL_WINO2R_ABLATE=$a"; for c in "64 16 16 256 256" "64 32 32 128 128" "64 128 128 32 32"; do WSL_WINO2R_ABLATE=$a MB_WINO=1 MB_RAW=1 python tools/microbench_conv.py $c 2>&1 | grep "us"; done; done
