#!/usr/bin/env bash
# reworked conv_sp against the previous kernels and its own A / B switches, every layer shape, best of 3 rounds of 30 launches
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
export SWEEP_BEST=3 SWEEP_REPS=30
timeout 600 python -m pytest tests/test_ops_convsp.py tests/test_concurrency.py -x -q -m gpu 2>&1 | tail -2 | tee -a "$O/sweep_ab.log"
for v in ${VARIANTS:-product prev}; do
  lib=""; [ "$v" != product ] && lib="tools/exp/libwslhip_$v.so"
  echo "== $v" | tee -a "$O/sweep_ab.log"
  WSL_LIB=$lib timeout 300 python tools/sweep_layers_sp.py --dec --only-sp 2>&1 | grep "@\|sum" | cut -d'|' -f2,4,6,8 | tee -a "$O/sweep_ab.log"
done
