#!/usr/bin/env bash
# round 5, bundle j: XCD-aware tile order of the GatedCRF kernel: GPU loss tests, then its time in the step of both builds
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
(timeout 600 python -m pytest tests/test_ops_loss.py -m gpu -q -x 2>&1 | tail -2)
VARIANTS="product prev2" PREC=f32 REPS=2 bash tools/gpu_step_ab.sh "$O" > /dev/null 2>&1
python - "$O" <<'PY'
import json, sys
for v in ("product", "prev2"):
    for r in (1, 2):
        d = json.loads(open(f"{sys.argv[1]}/bench_{v}_{r}.json").read()); h = d["roofline"]["hbm_roofline"]["kernels"]
        print(v, r, d["value"], d["ms_per_step"], "gatedcrf ms", h["gatedcrf_fwd_kernel"]["ms_per_step"], "head ms", h["loss_head(reduce+finalize+bwd+mix)"]["ms_per_step"])
PY
