#!/usr/bin/env bash
# round-3 record on the final tree: ACDC short schedule (matched HIP arm) alongside the GPU test-suite, then smoke, both bench lines with
# roofline.traffic collected in-run (--pmc-refresh), rocprofv3 summaries of both steps, SQ counters of the split step.
set -u
O=gpurun_out/r3u; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
rm -f gpurun_out/labelmap_rates.jsonl gpurun_out/fullsize_error_budget*.json gpurun_out/fullsize_replayed_decisions.json gpurun_out/fullres_regulariser_compositions.json
(bash tools/acdc_short_hip.sh "$O/acdc" matched > "$O/acdc.log" 2>&1) &
ACDC=$!
(timeout 1700 python -m pytest tests -m gpu -q --tb=short -s 2>&1 | grep -v "^iteration" | tail -70) > "$O/pytest_gpu.log"
cp gpurun_out/labelmap_rates.jsonl gpurun_out/fullsize_error_budget*.json gpurun_out/fullsize_replayed_decisions.json gpurun_out/fullres_regulariser_compositions.json "$O"/ 2>/dev/null
wait $ACDC
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > "$O/smoke.log"
(timeout 700 python bench.py --pmc-refresh 2>"$O/bench_stderr.log" | tail -1) > "$O/bench_default.json"
(timeout 500 python bench.py --conv-precision split_f16x3 --no-cpu-baseline --pmc-refresh 2>>"$O/bench_stderr.log" | tail -1) > "$O/bench_split.json"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_serial" -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --serial-decoders > "$R/$O/bench_serial_under_rocprof.log" 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_split" -- python "$R/bench.py" --conv-precision split_f16x3 --steps 10 --warmup 3 --no-cpu-baseline --serial-decoders > "$R/$O/bench_split_under_rocprof.log" 2>/dev/null
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d "$R/$O/pmc_sq1" -- python "$R/bench.py" --conv-precision split_f16x3 --steps 2 --warmup 1 --no-cpu-baseline --serial-decoders --no-prof > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$R/$O/pmc_sq2" -- python "$R/bench.py" --conv-precision split_f16x3 --steps 2 --warmup 1 --no-cpu-baseline --serial-decoders --no-prof > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d "$R/$O/pmc_sq3" -- python "$R/bench.py" --conv-precision split_f16x3 --steps 2 --warmup 1 --no-cpu-baseline --serial-decoders --no-prof > /dev/null 2>&1
cd "$R"; rm -f "$O"/prof_*/*/*kernel_trace.csv
python tools/pmc_mfma.py "$O/pmc_sq1" "$O/pmc_sq2" "$O/pmc_sq3" > "$O/pmc_sq_split.md" 2>/dev/null; rm -rf "$O/pmc_sq1" "$O/pmc_sq2" "$O/pmc_sq3"
tail -4 "$O/pytest_gpu.log"; cat "$O/smoke.log"; tail -9 "$O/acdc.log"
for f in default split; do python - "$O/bench_$f.json" "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); r = d["roofline"] or {}
    print(sys.argv[2], d["value"], d["ms_per_step"], r.get("kernel"), r.get("achieved"), r.get("frac"), r.get("traffic"), (r.get("traffic_detail") or {}).get("source", "")[:60], (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
python tools/check_rocprof_vs_bench.py "$O"/prof_serial/*/*kernel_stats.csv "$O/bench_serial_under_rocprof.log"
python tools/check_rocprof_vs_bench.py "$O"/prof_split/*/*kernel_stats.csv "$O/bench_split_under_rocprof.log"
