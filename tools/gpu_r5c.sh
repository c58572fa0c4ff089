#!/usr/bin/env bash
# round 5, bundle c: per-layer sweeps of several experiments builds on ONE box (LIBS="old new 3w": tools/exp/libwslhip_exp_<tag>.so, "new" =
# libwslhip_exp.so), then the f32 step product against round 4's build
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
timeout 600 python -m pytest tests/test_ops_conv.py tests/test_net.py -x -q -m gpu 2>&1 | tail -1 | tee "$O/pytest_conv.log"
for rep in 1 2; do
for t in ${LIBS:-old new}; do
  if [ "$t" = new ]; then unset WSL_EXP_LIB; else export WSL_EXP_LIB=$t; fi
  timeout 300 python tools/sweep_layers.py 2>/dev/null | tee "$O/sweep_${t}_$rep.log" | tail -11
done
done
unset WSL_EXP_LIB
[ "${STEP:-1}" = 1 ] && PREC=f32 VARIANTS="product r4" REPS="${REPS:-2}" bash tools/gpu_step_ab.sh "$O"
