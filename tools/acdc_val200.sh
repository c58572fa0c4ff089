#!/usr/bin/env bash
# the reference's full 60000-iteration fold-1 schedule (code/train_wss.sh:6-45) for pCE + TV and pCE at the REFERENCE's
# validation cadence -- every 200 iterations (train_weakly_supervised_pCE_TV_2D.py:144), 300 chances at a maximum of a noisy curve,
# where rounds 2-3 validated every 1000 -- two trainer processes side by side on one MI355X.  Needs data/ACDC.
#   bash tools/acdc_val200.sh gpurun_out/<tag>
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
pids=()
for loss in pce_tv pce; do
  (timeout 1700 python examples/train_acdc_scribble.py --root_path data/ACDC --fold fold1 --sup_type scribble --model unet --loss $loss \
     --labeled_type all --max_iterations 60000 --batch_size 12 --val_every 200 --log_every 1000 --no_hd95 --quiet --seed 2022 \
     --curve_json "$O/r4_acdc_fold1_val200_curve_$loss.json" > "$O/train_$loss.log" 2>&1) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
for loss in pce_tv pce; do tail -3 "$O/train_$loss.log"; done
