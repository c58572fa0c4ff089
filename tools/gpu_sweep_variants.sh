#!/usr/bin/env bash
# per-layer sweep of tagged experiments builds (wsl4mis_amd/csrc/build.sh expvar: tools/exp/libwslhip_exp_<tag>.so) against the plain
# experiments build on one box:   TAGS="v1 v2 v4" bash tools/gpu_sweep_variants.sh <out dir>
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
for t in "" ${TAGS:-v1 v2 v4} ""; do echo "== tag '${t}'"; WSL_EXP_LIB=$t timeout 600 python tools/sweep_layers.py 2>&1 | tail -11; done > "$O/sweep_variants.md" 2>&1
cat "$O/sweep_variants.md"
