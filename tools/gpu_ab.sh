#!/usr/bin/env bash
# step A / B of the product build against the round's base build on one box (REPS rounds, alternating) + per-family kernel times of both
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
REPS=${REPS:-3} PREC=${PREC:-f32} VARIANTS="${VARIANTS:-product r6base}" bash tools/gpu_step_ab.sh "$O/ab" 2>&1 | tail -8
python - "$O" ${VARIANTS:-product r6base} <<'PY'
import json, sys
for v in sys.argv[2:]:
    d = json.loads(open(f"{sys.argv[1]}/ab/bench_{v}_2.json").read()); r = d["roofline"]; h = r["hbm_roofline"]
    print(v, {k.split(" ")[0]: (x["ms_per_step"], x["avg_launch_us"]) for k, x in r["kernels"].items()})
    print(v, {k: x["ms_per_step"] for k, x in h["kernels"].items()}, h["all_hbm_kernels_ms_per_step"], r["all_mfma_kernels_ms_per_step"])
    print(v, {k: (x["ms"], x["avg_us"]) for k, x in d["kernels"].items() if "wino2" in k})
PY
