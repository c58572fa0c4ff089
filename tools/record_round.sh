#!/usr/bin/env bash
# End-of-round record on the GPU box: tests, smoke, the bench compositions, rocprofv3 summaries.  Usage (via gpurun):
#   bash tools/record_round.sh gpurun_out/<tag>
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
(timeout 600 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -4) > "$O/pytest_gpu.log"
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) > "$O/smoke.log"
(timeout 400 python bench.py 2>"$O/bench_stderr.log" | tail -1) > "$O/bench_default.json"
(timeout 300 python bench.py --loss ours_proposed --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_ours.json"
(timeout 300 python bench.py --crf-radius 2 --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_crf_r2.json"
(timeout 300 python bench.py --loss mean_teacher 2>/dev/null | tail -1) > "$O/bench_mt.json"
(timeout 300 python bench.py --serial-decoders --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_serial.json"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_serial" -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --serial-decoders > "$R/$O/bench_serial_under_rocprof.log" 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_default" -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
# HBM traffic of the MFMA kernels: two counter-only passes (never combined with a trace domain)
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/$O/pmc_fetch" -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --serial-decoders --no-prof > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/$O/pmc_write" -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --serial-decoders --no-prof > /dev/null 2>&1
cd "$R"; rm -f "$O"/prof_*/*/*kernel_trace.csv
python tools/pmc_traffic.py "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_traffic.json" > /dev/null 2>&1; rm -rf "$O/pmc_fetch" "$O/pmc_write"
tail -1 "$O/pytest_gpu.log"; cat "$O/smoke.log"
for f in default ours crf_r2 mt serial; do python - "$O/bench_$f.json" "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read()); r = d["roofline"]
print(sys.argv[2], d["value"], d["ms_per_step"], r["achieved"], r["frac"], r["avg_launch_us"], (d.get("cpu_baseline") or {}).get("value"))
PY
done
python tools/check_rocprof_vs_bench.py "$O"/prof_serial/*/*kernel_stats.csv "$O/bench_serial_under_rocprof.log"
