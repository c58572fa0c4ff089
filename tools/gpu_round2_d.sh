#!/usr/bin/env bash
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
(timeout 300 python bench.py --no-cpu-baseline --no-prof 2>/dev/null | tail -1) > "$O/bench_noprof.json"
(timeout 300 python bench.py --force-dp --no-cpu-baseline --no-prof 2>/dev/null | tail -1) > "$O/bench_forcedp_noprof.json"
(timeout 300 python -X faulthandler bench.py --force-dp --no-cpu-baseline > "$O/bench_forcedp.json" 2>"$O/bench_forcedp_stderr.log"; echo "rc=$?" >> "$O/bench_forcedp_stderr.log")
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/trace" -- python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-prof --force-dp > /dev/null 2>&1
cd "$R"; python tools/per_launch_table.py "$O/trace" 5 "$O/per_launch_dp_all.md" --all; rm -rf "$O/trace"
python - "$O" <<'PY'
import json, sys
for f in ("noprof", "forcedp_noprof", "forcedp"):
    try:
        d = json.loads(open(f"{sys.argv[1]}/bench_{f}.json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("dp", {}).get("per_rank_ms_per_step_main_stream_blocked_on_allreduce"))
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -12 "$O/bench_forcedp_stderr.log"
