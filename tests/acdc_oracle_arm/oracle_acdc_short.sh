#!/usr/bin/env bash
# Oracle arm of profiles/r3_acdc_short_schedule.md: four CPU runs (two at a time, 3 threads each), build container only.
set -u
cd "$(dirname "$0")/../.."
O=gpurun_out/r3_oracle_short; mkdir -p "$O"
for seed in 2022 11; do
  pids=()
  for loss in pce_tv pce; do
    nice -n 19 python tests/acdc_oracle_arm/oracle_acdc_short.py --loss $loss --seed $seed --stop_iterations "${1:-1500}" --val_every 100 --threads 3 \
      --curve_json profiles/r3_acdc_short_oracle_${loss}_seed$seed.json > "$O/${loss}_seed$seed.log" 2>&1 &
    pids+=($!)
  done
  for p in "${pids[@]}"; do wait "$p"; done
done
