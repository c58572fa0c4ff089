#!/usr/bin/env bash
# What changes when a kernel switch is flipped that must not alter the arithmetic: (1) gradient digests of one full-size step in both
# precisions, product build against the build before the flip (tools/exp/libwslhip_prev.so) -- must be equal; (2) the default bench line;
# (3) rocprofv3 --kernel-trace --stats of both precisions and the events-vs-rocprof cross-check.  About 3 GPU-minutes.
#   bash tools/record_lite.sh gpurun_out/<tag>
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
(python tools/grad_digest.py 2>&1 | sed 's/^/product /'; python tools/grad_digest.py --lib tools/exp/libwslhip_prev.so 2>&1 | sed 's/^/prev    /') | tee "$O/grad_digest.log"
(timeout 300 python bench.py 2>"$O/bench_stderr.log" | tail -1) > "$O/bench_default.json"
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-split-record"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_f32_serial" -- $B --steps 10 --warmup 3 --serial-decoders > "$R/$O/bench_serial_under_rocprof.log" 2>/dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_split_serial" -- $B --steps 10 --warmup 3 --serial-decoders --conv-precision split_f16x3 > "$R/$O/bench_split_serial_under_rocprof.log" 2>/dev/null
cd "$R"; rm -f "$O"/prof_*/*/*kernel_trace.csv
python - "$O/bench_default.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read()); s = d.get("split_f16x3") or {}
print("default", d["value"], d["ms_per_step"], "split:", s.get("value"), s.get("ms_per_step"))
PY
python tools/check_rocprof_vs_bench.py "$O"/prof_f32_serial/*/*kernel_stats.csv "$O/bench_serial_under_rocprof.log"
python tools/check_rocprof_vs_bench.py "$O"/prof_split_serial/*/*kernel_stats.csv "$O/bench_split_serial_under_rocprof.log"
