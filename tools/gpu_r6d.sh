#!/usr/bin/env bash
# round 6, call d: is the epilogue's store cost (60 us of 202 on 16 -> 16 @ 256^2) the fragmented address pattern?  Arm 128 stores the same
# bytes in the pattern a lane-transposed epilogue would have (whole 64-byte lines per store instruction).  + the strict unet test on the GPU.
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
for a in 0 128 32; do echo "== WSL_WINO2R_ABLATE=$a"; for c in "64 16 16 256 256" "64 32 32 128 128" "64 64 64 64 64" "64 128 128 32 32"; do WSL_WINO2R_ABLATE=$a MB_WINO=1 MB_RAW=1 python tools/microbench_conv.py $c 2>&1 | grep "us"; done; done > "$O/abl_store_pattern.log" 2>&1
cat "$O/abl_store_pattern.log"
(timeout 900 python -m pytest tests/test_error_budget.py -m gpu -q --tb=short -s -k "unet_compositions_strict" 2>&1 | tail -8) > "$O/pytest_unet_strict.log"; cat "$O/pytest_unet_strict.log" | cut -c1-400
