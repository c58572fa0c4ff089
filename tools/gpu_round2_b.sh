#!/usr/bin/env bash
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
(timeout 600 python -m pytest tests/test_ops_loss.py tests/test_python_api.py -m gpu -q --tb=short -x 2>&1 | tail -5) > "$O/pytest_loss.log"
(timeout 300 python bench.py --no-cpu-baseline 2>"$O/bench_stderr.log" | tail -1) > "$O/bench_default.json"
(timeout 300 python bench.py --force-dp --no-cpu-baseline 2>"$O/bench_forcedp_stderr.log" | tail -1) > "$O/bench_forcedp.json"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/trace" -- python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --serial-decoders --no-prof > /dev/null 2>&1
cd "$R"; python tools/per_launch_table.py "$O/trace" 5 "$O/per_launch.md"; rm -rf "$O/trace"
tail -2 "$O/pytest_loss.log"
python - "$O" <<'PY'
import json, sys
for f in ("default", "forcedp"):
    try:
        d = json.loads(open(f"{sys.argv[1]}/bench_{f}.json").read()); r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], r["hbm_roofline"]["kernels"].get("gatedcrf_fwd_kernel"), d.get("dp"))
    except Exception as e:
        print(f, "FAILED", e)
PY
