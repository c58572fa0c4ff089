#!/usr/bin/env bash
# the last combination: workgroups of a CU guaranteed out of phase (ticket per CU) AND the wave in its MFMA loop favoured by the issue
# arbiter (s_setprio 2) -- if staging and matrix work of two waves of a SIMD can overlap at all, this is where it shows
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
export SWEEP_BEST=3 SWEEP_REPS=30 WSL_LIB=tools/exp/libwslhip_expprio.so WSL_SP_STAGGER_MODE=1
for sg in 0 1 2 3; do
  echo "== prio 2, stagger $sg" | tee -a "$O/prio_stagger.log"
  WSL_SP_STAGGER=$sg timeout 200 python tools/sweep_layers_sp.py --mid --only-sp 2>&1 | grep "@" | cut -d'|' -f2,4,6 | tee -a "$O/prio_stagger.log"
done
