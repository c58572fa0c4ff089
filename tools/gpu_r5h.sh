#!/usr/bin/env bash
# round 5, bundle h: f32 step of the product build against tools/exp/libwslhip_prev.so (alternating, one box) + GPU tests of the conv ops
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
(timeout 900 python -m pytest tests/test_ops_conv.py tests/test_ops_convsp.py tests/test_net.py -m gpu -q -x 2>&1 | tail -2) > "$O/pytest_conv_gpu.log"
cat "$O/pytest_conv_gpu.log"
VARIANTS="product ${PREV:-prev}" PREC=f32 REPS=${REPS:-3} bash tools/gpu_step_ab.sh "$O" 2>&1 | tail -6
