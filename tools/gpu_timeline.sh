#!/usr/bin/env bash
# kernel TIMELINE of the default two-stream step (rocprofv3 --kernel-trace, timestamps kept): how much of a step has no
# kernel running (dependency gaps), how much has two; analysed by tools/timeline_gaps.py
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-split-record --no-pmc-refresh --no-prof --repeats 1"
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/trace2" -- $B --steps 6 --warmup 4 > "$R/$O/bench_trace2.log" 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/trace1" -- $B --steps 6 --warmup 4 --serial-decoders > "$R/$O/bench_trace1.log" 2>/dev/null
cd "$R"
for t in trace2 trace1; do python tools/timeline_gaps.py "$O"/$t/*/*kernel_trace.csv > "$O/timeline_$t.md"; cat "$O/timeline_$t.md"; done
rm -f "$O"/trace*/*/*kernel_trace.csv
