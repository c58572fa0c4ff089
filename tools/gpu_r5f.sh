#!/usr/bin/env bash
# round 5, bundle f: the interleaved tile walk in all three persistent weight-gradient kernels: GPU tests of the conv ops, then the step of
# both precisions against the build before the change, alternating on one box
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
(timeout 900 python -m pytest tests/test_ops_conv.py tests/test_ops_convsp.py tests/test_net.py -m gpu -q -x 2>&1 | tail -5) > "$O/pytest_conv_gpu.log"
cat "$O/pytest_conv_gpu.log"
VARIANTS="product prev" PREC=split_f16x3 REPS=2 bash tools/gpu_step_ab.sh "$O" 2>&1 | tail -4
VARIANTS="product prev" PREC=f32 REPS=2 bash tools/gpu_step_ab.sh "$O" 2>&1 | tail -4
