#!/usr/bin/env bash
# what differs between boxes: the default (two-stream) and the serialised line, the kernel timeline of the two-stream step, GPU clocks / power
# sampled by rocm-smi during the two-stream run, host facts
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
B="python bench.py --no-cpu-baseline --no-split-record --no-pmc-refresh --no-prof --steps 40 --warmup 10"
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower --showtemp --json 2>/dev/null | head -c 1500; echo; sleep 0.5; done ) > "$O/smi.log" 2>&1 &
SMI=$!
$B 2>/dev/null | tail -1 > "$O/bench_two.json"
$B --serial-decoders 2>/dev/null | tail -1 > "$O/bench_serial.json"
kill $SMI 2>/dev/null
python - "$O" <<'PY'
import json, sys, re
o = sys.argv[1]
a, b = (json.loads(open(f"{o}/bench_{n}.json").read()) for n in ("two", "serial"))
print("two-stream", a["value"], a["repeats"]["values"], "serial", b["value"], "gain", round(a["value"] / b["value"], 4))
sc, pw = [], []
for ln in open(f"{o}/smi.log"):
    m = re.search(r'"sclk clock speed:"\s*:\s*"\((\d+)Mhz\)"', ln)
    if m: sc.append(int(m.group(1)))
    m = re.search(r'Power \(W\)"\s*:\s*"([\d.]+)"', ln)
    if m: pw.append(float(m.group(1)))
print("sclk samples", sc[:30], "power", pw[:30])
PY
nproc; lscpu | grep -E "Model name|MHz" | head -3; uptime
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/$O/trace2" -- python $R/bench.py --no-cpu-baseline --no-split-record --no-pmc-refresh --no-prof --repeats 1 --steps 6 --warmup 4 > /dev/null 2>&1
cd "$R"; python tools/timeline_gaps.py "$O"/trace2/*/*kernel_trace.csv | head -3; rm -f "$O"/trace2/*/*kernel_trace.csv
