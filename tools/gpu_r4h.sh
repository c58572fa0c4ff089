#!/usr/bin/env bash
# does the second workgroup per CU buy anything?  conv_sp mid layers at 1 / 2 workgroups per CU, whole kernel and MFMA-only ablation
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
export SWEEP_BEST=3 SWEEP_REPS=30
for pc in 1 2; do for a in 0 30 1; do
  echo "== per_cu $pc ablate $a" | tee -a "$O/percu.log"
  WSL_SP_PERCU=$pc WSL_SP_ABLATE=$a timeout 200 python tools/sweep_layers_sp.py --mid --only-sp --exp 2>&1 | grep "@" | cut -d'|' -f2,4,6 | tee -a "$O/percu.log"
done; done
