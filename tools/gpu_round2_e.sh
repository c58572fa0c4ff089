#!/usr/bin/env bash
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd /tmp; export TMPDIR=/tmp
for m in nodp dp; do
  f=""; [ $m = dp ] && f="--force-dp"
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/trace_$m" -- python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-prof --serial-decoders $f > /dev/null 2>&1
  python "$R/tools/per_launch_table.py" "$R/$O/trace_$m" 5 "$R/$O/per_launch_$m.md"; rm -rf "$R/$O/trace_$m"
done
