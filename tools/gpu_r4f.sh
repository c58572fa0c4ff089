#!/usr/bin/env bash
# where do the mid layers of conv_sp spend their time after the rework?  ablations (experiments build; WRONG results by design), best of 3
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
export SWEEP_BEST=3 SWEEP_REPS=30
for a in 0 1 3 11 27 31 30 28 24 16 4 20; do
  echo "== ablate $a" | tee -a "$O/ablate.log"
  WSL_SP_ABLATE=$a timeout 200 python tools/sweep_layers_sp.py --mid --only-sp --exp 2>&1 | grep "@" | cut -d'|' -f2,4,6 | tee -a "$O/ablate.log"
done
echo "== old lib" | tee -a "$O/ablate.log"
WSL_LIB=tools/exp/libwslhip_old.so timeout 200 python tools/sweep_layers_sp.py --mid --only-sp 2>&1 | grep "@" | cut -d'|' -f2,4,6 | tee -a "$O/ablate.log"
echo "== product" | tee -a "$O/ablate.log"
timeout 200 python tools/sweep_layers_sp.py --mid --only-sp 2>&1 | grep "@" | cut -d'|' -f2,4,6 | tee -a "$O/ablate.log"
