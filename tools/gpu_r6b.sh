#!/usr/bin/env bash
# round 6, call b: separable bilinear backward (tests + step A/B against the round's base build) and the phase ablations of the raw-source
# Winograd conv incl. the new arm 16 (half the input transforms per MFMA = the upper bound of a 64-channel output block, NT = 4)
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
(timeout 600 python -m pytest tests/test_ops_bn_pool_up.py tests/test_net.py -m gpu -q --tb=short 2>&1 | tail -6) > "$O/pytest.log"; tail -3 "$O/pytest.log"
REPS=3 PREC=f32 VARIANTS="product r6base" bash tools/gpu_step_ab.sh "$O/ab" 2>&1 | tail -8
for a in 0 16 4 20 2 6; do echo "== WSL_WINO2R_ABLATE=$a"; for c in "64 16 16 256 256" "64 32 32 128 128" "64 64 64 64 64" "64 128 128 32 32" "64 256 256 16 16"; do WSL_WINO2R_ABLATE=$a MB_WINO=1 MB_RAW=1 python tools/microbench_conv.py $c 2>&1 | grep "us"; done; done > "$O/abl_wino2r.log" 2>&1
cat "$O/abl_wino2r.log"
