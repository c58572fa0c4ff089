#!/usr/bin/env bash
# round 6, first call: GPU tests (all but the fp64 error-budget module) on the hook-free product build, the default bench line in its new
# format (3 timed regions, library_sha, nested modules_path / split record / traffic), the --path modules line alone, a rocprofv3 kernel
# summary of the serialised step and the per-layer sweep -- this round's baseline on ONE box
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
(timeout 1300 python -m pytest tests -m gpu -q --tb=short --deselect tests/test_error_budget.py 2>&1 | tail -30) > "$O/pytest.log"
tail -5 "$O/pytest.log"
(timeout 900 python bench.py --steps 20 --warmup 5 2>"$O/bench_stderr.log" | tail -1) > "$O/bench_default.json"
(timeout 300 python bench.py --path modules --steps 20 --warmup 5 2>"$O/modules_stderr.log" | tail -1) > "$O/bench_modules.json"
python - "$O/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print("default", d["value"], d["ms_per_step"], d.get("repeats", {}).get("values"), "sha", (d["config"].get("library_sha") or "")[:12],
      "split", (d.get("split_f16x3") or {}).get("value"), "modules", (d.get("modules_path") or {}).get("value"),
      (d.get("modules_path") or {}).get("aten_kernels_per_step"), "traffic", (d["roofline"] or {}).get("traffic"),
      ((d["roofline"] or {}).get("traffic_detail") or {}).get("algorithmic_bytes_per_launch"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
tail -c 1500 "$O/bench_modules.json"; echo
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-split-record --repeats 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_f32_serial" -- $B --steps 10 --warmup 3 --serial-decoders > "$R/$O/bench_serial_under_rocprof.log" 2>/dev/null
cd "$R"; rm -f "$O"/prof_*/*/*kernel_trace.csv
(timeout 600 python tools/sweep_layers.py 2>&1 | tail -14) > "$O/sweep_layers.md"
cat "$O/sweep_layers.md"
