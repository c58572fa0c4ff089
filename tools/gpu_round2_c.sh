#!/usr/bin/env bash
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
(timeout 900 python -m pytest tests/test_ops_bn_pool_up.py tests/test_ops_conv.py tests/test_net.py tests/test_fullsize.py tests/test_python_api.py -m gpu -q --tb=short -x 2>&1 | tail -15) > "$O/pytest.log"
(timeout 300 python bench.py --no-cpu-baseline 2>"$O/bench_stderr.log" | tail -1) > "$O/bench_default.json"
(timeout 200 python -X faulthandler bench.py --force-dp --no-cpu-baseline --no-prof --steps 5 --warmup 2 > "$O/bench_forcedp.json" 2>"$O/bench_forcedp_stderr.log"; echo "rc=$?" >> "$O/bench_forcedp_stderr.log")
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/trace" -- python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --serial-decoders --no-prof > /dev/null 2>&1
cd "$R"; python tools/per_launch_table.py "$O/trace" 5 "$O/per_launch.md"; rm -rf "$O/trace"
tail -3 "$O/pytest.log"; tail -5 "$O/bench_forcedp_stderr.log"
python - "$O" <<'PY'
import json, sys
d = json.loads(open(f"{sys.argv[1]}/bench_default.json").read()); r = d["roofline"]
print("default", d["value"], d["ms_per_step"], {k: v["ms_per_step"] for k, v in r["hbm_roofline"]["kernels"].items()}, {k[:20]: v["ms_per_step"] for k, v in r["kernels"].items()})
PY
