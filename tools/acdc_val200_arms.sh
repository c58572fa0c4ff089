#!/usr/bin/env bash
# Follow-up arms of tools/acdc_val200.sh (profiles/r4_acdc_val200.md): the reference's pCE script defaults to base lr 0.03
# (train_weakly_supervised_pCE_2D.py:48) and the README's plot was made on a 30000-iteration schedule (its lr panel reaches 0 at 30 k).
#   A: pce, lr 0.03, 60000 iterations   B: pce_tv, lr 0.01, 30000   C: pce, lr 0.03, 30000        (three trainers side by side)
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
run() {  # tag loss lr iters
  (timeout 1700 python examples/train_acdc_scribble.py --root_path data/ACDC --fold fold1 --sup_type scribble --model unet --loss $2 \
     --labeled_type all --max_iterations $4 --base_lr $3 --batch_size 12 --val_every 200 --log_every 1000 --no_hd95 --quiet --seed 2022 \
     --curve_json "$O/r4_acdc_fold1_val200_curve_$1.json" > "$O/train_$1.log" 2>&1) &
}
run pce_lr03_60k pce 0.03 60000
run pce_tv_30k pce_tv 0.01 30000
run pce_lr03_30k pce 0.03 30000
wait
for t in pce_lr03_60k pce_tv_30k pce_lr03_30k; do echo "== $t"; tail -2 "$O/train_$t.log"; done
