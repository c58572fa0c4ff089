#!/usr/bin/env bash
# HIP arm of profiles/r3_acdc_short_schedule.md: the schedule of tests/acdc_oracle_arm/oracle_acdc_short.sh (1500 iterations of the 60000-iteration
# poly schedule, batch 12, validation every 100, seeds 2022 and 11, pCE and pCE + TV), four trainer processes side by side on ONE
# MI355X.  Needs data/ACDC.      bash tools/acdc_short_hip.sh gpurun_out/<tag> [matched]
set -u
# "matched": the oracle arm's initial state (tests/acdc_oracle_arm/make_acdc_init.py) and its dropout-mask stream (--oracle_stream) -- the two arms
# then run the same trajectory up to fp32 round-off
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
M="${2:-}"; tag=hip; [ "$M" = matched ] && tag=hip_matched
pids=()
for loss in pce pce_tv; do for seed in 2022 11; do
  (timeout 1200 python examples/train_acdc_scribble.py --root_path data/ACDC --fold fold1 --sup_type scribble --model unet --loss $loss \
     --labeled_type all --max_iterations 60000 --stop_iterations 1500 --batch_size 12 --val_every 100 --log_every 20 --no_hd95 --quiet \
     --seed $seed $([ "$M" = matched ] && echo "--oracle_stream --resume tools/exp/acdc_init_unet_seed$seed.pth") \
     --curve_json "$O/r3_acdc_short_${tag}_${loss}_seed$seed.json" > "$O/train_${loss}_seed$seed.log" 2>&1) &
  pids+=($!)
done; done
for p in "${pids[@]}"; do wait "$p"; done
for f in "$O"/train_*.log; do echo "$f"; tail -2 "$f"; done
