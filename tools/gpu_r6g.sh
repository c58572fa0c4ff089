#!/usr/bin/env bash
# round 6, call g: the d form (data-gradient epilogue writes the gradient in front of the BatchNorm output, the apply pass reads no keep
# mask): GPU tests of the ops / nets + step A / B against the round's base build on one box
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
(timeout 900 python -m pytest tests/test_ops_bn_pool_up.py tests/test_net.py tests/test_ops_conv.py tests/test_fullsize.py -m gpu -q --tb=short 2>&1 | tail -6) > "$O/pytest.log"; tail -3 "$O/pytest.log"
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) > "$O/smoke.log"; cat "$O/smoke.log"
REPS=3 PREC=f32 VARIANTS="product r6base" bash tools/gpu_step_ab.sh "$O/ab" 2>&1 | tail -8
python - "$O" <<'PY'
import json, sys
for v in ("product", "r6base"):
    d = json.loads(open(f"{sys.argv[1]}/ab/bench_{v}_2.json").read()); h = d["roofline"]["hbm_roofline"]
    print(v, {k: x["ms_per_step"] for k, x in h["kernels"].items()}, h["all_hbm_kernels_ms_per_step"], d["roofline"]["all_mfma_kernels_ms_per_step"])
PY
