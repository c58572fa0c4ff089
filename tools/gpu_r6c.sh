#!/usr/bin/env bash
# round 6, call c: per-layer sweep with the REAL data-gradient launches (BatchNorm-backward epilogue) as a fourth column, and the epilogue
# ablations of the raw-source Winograd conv: 32 = no global stores, 64 = no statistics, 96 = output transform only, 4 = no epilogue
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
(timeout 600 python tools/sweep_layers.py 2>&1 | tail -14) > "$O/sweep_layers.md"; cat "$O/sweep_layers.md"
for a in 0 32 64 96 4; do echo "== WSL_WINO2R_ABLATE=$a"; for c in "64 16 16 256 256" "64 32 32 128 128" "64 64 64 64 64"; do WSL_WINO2R_ABLATE=$a MB_WINO=1 MB_RAW=1 python tools/microbench_conv.py $c 2>&1 | grep "us"; done; done > "$O/abl_epilogue.log" 2>&1
cat "$O/abl_epilogue.log"
