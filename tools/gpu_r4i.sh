#!/usr/bin/env bash
# two workgroups per CU started out of phase (the second half of the grid sleeps first): do staging and matrix work overlap then?
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
export SWEEP_BEST=3 SWEEP_REPS=30
for sg in 0 1 2 3; do
  echo "== stagger $sg" | tee -a "$O/stagger.log"
  WSL_SP_STAGGER=$sg timeout 200 python tools/sweep_layers_sp.py --mid --only-sp --exp 2>&1 | grep "@" | cut -d'|' -f2,4,6 | tee -a "$O/stagger.log"
done
