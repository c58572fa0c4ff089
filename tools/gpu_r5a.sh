#!/usr/bin/env bash
# round 5, bundle a: the conflict-free LDS layouts of the Winograd kernels against round 4's build on ONE box
#   1. numerics of the new build on the hardware (the conv op tests: every Winograd instantiation, edge tiles, two sources)
#   2. per-layer sweep, old experiments build then new
#   3. the f32 step, product library against round 4's (tools/exp/libwslhip_r4.so), alternating
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
timeout 600 python -m pytest tests/test_ops_conv.py tests/test_net.py -x -q -m gpu 2>&1 | tail -5 | tee "$O/pytest_conv.log"
WSL_EXP_LIB=old timeout 300 python tools/sweep_layers.py 2>&1 | tee "$O/sweep_old.log"
timeout 300 python tools/sweep_layers.py 2>&1 | tee "$O/sweep_new.log"
PREC=f32 VARIANTS="product r4" REPS="${REPS:-2}" bash tools/gpu_step_ab.sh "$O"
