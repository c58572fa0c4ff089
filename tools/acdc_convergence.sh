#!/usr/bin/env bash
# Real-data convergence record: the reference's canonical fold-1 schedule (code/train_wss.sh:6-45:
# --max_iterations 60000 --batch_size 12, unet, SGD 0.01 poly) on the ACDC scribbles for pCE, pCE + TV and pCE + GatedCRF,
# three trainer processes side by side on ONE MI355X.  Needs data/ACDC (remove `data/` from .gpurunignore for this call).
#   bash tools/acdc_convergence.sh gpurun_out/<tag> [max_iterations]
set -u
O="$1"; IT="${2:-60000}"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
pids=()
for loss in pce pce_tv pce_gatedcrf; do
  (timeout 3000 python examples/train_acdc_scribble.py --root_path data/ACDC --fold fold1 --sup_type scribble --model unet --loss $loss \
     --labeled_type all --max_iterations "$IT" --batch_size 12 --val_every 1000 --log_every 100 --no_hd95 --quiet \
     --curve_json "$O/curve_$loss.json" > "$O/train_$loss.log" 2>&1) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
for loss in pce pce_tv pce_gatedcrf; do tail -2 "$O/train_$loss.log"; done
