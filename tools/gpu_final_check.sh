#!/usr/bin/env bash
# last call of a round: the GPU suite as the driver runs it (no fp64 budget leg), smoke, the default bench line -- on the tree as committed
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
(timeout 1200 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -5) > "$O/pytest_gpu.log"
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) > "$O/smoke.log"
(timeout 600 python bench.py 2>"$O/bench_stderr.log" | tail -1) > "$O/bench_default.json"
cat "$O/pytest_gpu.log" "$O/smoke.log"; python - "$O/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read()); r = d["roofline"]
print(d["value"], d["ms_per_step"], "split", d["split_f16x3"]["value"], "cpu", d["cpu_baseline"]["value"], "traffic", r["traffic"], r["traffic_detail"]["source"][:40])
PY
