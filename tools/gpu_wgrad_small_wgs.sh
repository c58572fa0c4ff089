#!/usr/bin/env bash
# persistent-workgroup count of the narrow-operand weight gradients (first conv 1 -> 16, classifier 16 -> 4 at 256^2):
# 768 (3 per CU, the plan) against 1024 / 1536 / 2048 -- they hold 68 / 101 registers, the HBM-side kernels at 0.33-0.36 of peak
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
for w in 768 1024 1536 2048 3072; do echo "== WSL_WGRAD_WGS=$w"; for c in "64 1 16 256 256" "64 16 4 256 256"; do WSL_WGRAD_WGS=$w python tools/microbench_wgrad.py $c 2>&1 | grep "us"; done; done > "$O/wgrad_small_wgs.log" 2>&1
cat "$O/wgrad_small_wgs.log"
