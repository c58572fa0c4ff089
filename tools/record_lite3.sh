#!/usr/bin/env bash
# the remaining bench compositions on the final tree
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
(timeout 120 python bench.py --crf-radius 2 --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_crf_r2.json"
(timeout 120 python bench.py --serial-decoders --no-cpu-baseline --no-split-record 2>/dev/null | tail -1) > "$O/bench_serial.json"
for l in pce_tv pce_ms pce_entropy; do (timeout 120 python bench.py --loss $l --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_unet_$l.json"; done
for f in crf_r2 serial unet_pce_tv unet_pce_ms unet_pce_entropy; do python - "$O/bench_$f.json" "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); s = d.get("split_f16x3") or {}
    print(sys.argv[2], d["value"], d["ms_per_step"], "split:", s.get("value"), s.get("ms_per_step"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
