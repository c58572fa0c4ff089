#!/usr/bin/env bash
# phase ablations of wgrad_wino_kernel (experiments build): 1 = no matrix phase, 2 = loads + staging only for the first tile
for a in ${ABLS:-0 1 2 3 6 10 14}; do echo "== WSL_WGWINO_ABLATE=$a"; for c in "64 16 16 256 256" "64 32 16 256 256" "64 32 32 128 128" "64 128 128 32 32"; do WSL_WGWINO_ABLATE=$a python tools/microbench_wgrad.py $c 2>&1 | grep wgrad; done; done
