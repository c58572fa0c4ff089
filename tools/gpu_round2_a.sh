#!/usr/bin/env bash
# Round-2 first GPU pass: new parity tests + bench lines with the wider event families.  Usage (via gpurun):
#   bash tools/gpu_round2_a.sh gpurun_out/<tag>
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
rm -f gpurun_out/labelmap_rates.jsonl
(timeout 1500 python -m pytest tests -m gpu -q --tb=short -s --durations=12 2>&1 | tail -150) > "$O/pytest_gpu.log"
cp gpurun_out/labelmap_rates.jsonl gpurun_out/fullsize_error_budget.json "$O"/ 2>/dev/null
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) > "$O/smoke.log"
(timeout 500 python bench.py 2>"$O/bench_stderr.log" | tail -1) > "$O/bench_default.json"
(timeout 400 python bench.py --loss pce 2>/dev/null | tail -1) > "$O/bench_pce.json"
(timeout 300 python bench.py --force-dp --no-cpu-baseline 2>"$O/bench_forcedp_stderr.log" | tail -1) > "$O/bench_forcedp.json"
(timeout 300 python bench.py --net unet --loss pce --cpu-threads 0 2>/dev/null | tail -1) > "$O/bench_unet_pce.json"
(timeout 300 python bench.py --loss mean_teacher 2>/dev/null | tail -1) > "$O/bench_mt.json"
nproc > "$O/host.txt"; grep MemTotal /proc/meminfo >> "$O/host.txt"
tail -3 "$O/pytest_gpu.log"; cat "$O/smoke.log"
for f in default pce forcedp unet_pce mt; do python - "$O/bench_$f.json" "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); r = d["roofline"]
    print(sys.argv[2], d["value"], d["ms_per_step"], r["achieved"], r["frac"], r["issued_frac"], r["whole_step_issued_frac"], (d.get("cpu_baseline") or {}).get("value"), d.get("dp"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
