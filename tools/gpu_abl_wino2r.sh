#!/usr/bin/env bash
# Phase ablations of the raw-source Winograd conv (experiments build; wrong results by design, timing only) per layer shape, one box:
#   bash tools/gpu_abl_wino2r.sh <out dir> ["0 16 4 20 2 6 32 64 96 128"]
# arms: 1 no MFMAs, 2 no DMA after the first chunk, 4 no epilogue, 6 channel loop only, 8 no patch reads, 16 half the input transforms per MFMA
# (the ceiling of a 64-channel output block), 32 epilogue without its stores, 64 without its statistics, 96 output transform only, 128 stores
# in the address pattern of a lane-transposed epilogue.  Record: profiles/r6_wino2r_ablations.md
set -u
O="$1"; ARMS="${2:-0 16 4 20 2 6 32 64 96 128}"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
for a in $ARMS; do echo "== WSL_WINO2R_ABLATE=$a"; for c in "64 16 16 256 256" "64 32 32 128 128" "64 64 64 64 64" "64 128 128 32 32" "64 256 256 16 16"; do WSL_WINO2R_ABLATE=$a MB_WINO=1 MB_RAW=1 python tools/microbench_conv.py $c 2>&1 | grep "us"; done; done > "$O/abl_wino2r.log" 2>&1
cat "$O/abl_wino2r.log"
