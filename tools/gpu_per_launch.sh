#!/usr/bin/env bash
# per-launch table of one serialised step (rocprofv3 kernel trace): bash tools/gpu_per_launch.sh gpurun_out/<tag> [bench args]
set -u
O="$1"; shift; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/trace" -- python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-prof --serial-decoders "$@" > /dev/null 2>&1
python "$R/tools/per_launch_table.py" "$R/$O/trace" 5 "$R/$O/per_launch_serial.md"; rm -rf "$R/$O/trace"
tail -3 "$R/$O/per_launch_serial.md"
