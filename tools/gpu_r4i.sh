#!/usr/bin/env bash
# two workgroups per CU started out of phase: do staging and matrix work overlap then?  mode 0: the second half of the grid sleeps first;
# mode 1: every second workgroup to arrive on a CU (ticket per CU from HW_ID / XCC_ID) -- units of ~1 us
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
export SWEEP_BEST=3 SWEEP_REPS=30
for mode in 1 0; do for sg in 0 1 2 3 4 6; do
  echo "== mode $mode stagger $sg" | tee -a "$O/stagger.log"
  WSL_SP_STAGGER_MODE=$mode WSL_SP_STAGGER=$sg timeout 200 python tools/sweep_layers_sp.py --mid --only-sp --exp 2>&1 | grep "@" | cut -d'|' -f2,4,6 | tee -a "$O/stagger.log"
done; done
