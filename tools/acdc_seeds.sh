#!/usr/bin/env bash
# Seed spread of the two compositions that landed below the README's figures: pCE and pCE + TV, two more seeds each,
# same schedule as tools/acdc_convergence.sh, four trainer processes side by side on ONE MI355X.  Needs data/ACDC.
#   bash tools/acdc_seeds.sh gpurun_out/<tag> [max_iterations]
set -u
O="$1"; IT="${2:-60000}"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
pids=()
for loss in pce pce_tv; do for seed in 11 12; do
  (timeout 2400 python examples/train_acdc_scribble.py --root_path data/ACDC --fold fold1 --sup_type scribble --model unet --loss $loss \
     --labeled_type all --max_iterations "$IT" --batch_size 12 --val_every 1000 --log_every 100 --no_hd95 --quiet --seed $seed \
     --curve_json "$O/curve_${loss}_seed$seed.json" > "$O/train_${loss}_seed$seed.log" 2>&1) &
  pids+=($!)
done; done
for p in "${pids[@]}"; do wait "$p"; done
for f in "$O"/train_*.log; do echo "$f"; tail -2 "$f"; done
