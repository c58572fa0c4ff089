#!/usr/bin/env bash
# WIDE Winograd tiles (4 x 128 for the 16-channel layers, 4 x 64 for the 32-channel blocks) against 8 x 64 / 8 x 32, per layer
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
for k in "0 0" "1 0" "0 64" "1 64"; do set -- $k; echo "== WSL_WINO16_WIDE=$1 WSL_WINO32_WIDE=$2"; WSL_WINO16_WIDE=$1 WSL_WINO32_WIDE=$2 timeout 600 python tools/sweep_layers.py 2>&1 | tail -12; done > "$O/sweep_wide.md" 2>&1
cat "$O/sweep_wide.md"
