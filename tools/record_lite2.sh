#!/usr/bin/env bash
# the other bench compositions on the final tree (each with its nested split record)
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
(timeout 120 python bench.py --loss pce --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_pce.json"
(timeout 120 python bench.py --loss ours_proposed --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_ours.json"
(timeout 120 python bench.py --loss mean_teacher --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_mt.json"
(timeout 120 python bench.py --net unet --loss pce --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_unet_pce.json"
for f in pce ours mt unet_pce; do python - "$O/bench_$f.json" "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); s = d.get("split_f16x3") or {}
    print(sys.argv[2], d["value"], d["ms_per_step"], "split:", s.get("value"), s.get("ms_per_step"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
