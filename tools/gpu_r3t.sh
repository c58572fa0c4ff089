#!/usr/bin/env bash
set -u
O=gpurun_out/r3t; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
(timeout 600 python -m pytest tests/test_ops_bn_pool_up.py tests/test_net.py tests/test_ops_convsp.py -m gpu -q --tb=short 2>&1 | tail -4) > "$O/pytest.log"; cat "$O/pytest.log"
(timeout 300 python bench.py --conv-precision split_f16x3 --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_split.json"
(timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_f32.json"
python - <<'PY'
import json
for f in ("bench_split", "bench_f32"):
    d = json.loads(open(f"gpurun_out/r3t/{f}.json").read())
    print(f, d["value"], d["ms_per_step"], {k: round(v["ms"], 2) for k, v in d["kernels"].items() if v["ms"] > 2})
PY
