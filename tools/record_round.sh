#!/usr/bin/env bash
# End-of-round record on the GPU box: tests (incl. the fp64 error budget), smoke, the bench compositions of BASELINE.json's
# configs, rocprofv3 summaries, PMC traffic + SQ passes.  Usage (via gpurun):   bash tools/record_round.sh gpurun_out/<tag>
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
rm -f gpurun_out/labelmap_rates.jsonl gpurun_out/fullsize_error_budget.json
(timeout 1500 python -m pytest tests -m gpu -q --tb=short -s 2>&1 | grep -v "^iteration" | tail -60) > "$O/pytest_gpu.log"
cp gpurun_out/labelmap_rates.jsonl gpurun_out/fullsize_error_budget.json "$O"/ 2>/dev/null
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) > "$O/smoke.log"
(timeout 500 python bench.py 2>"$O/bench_stderr.log" | tail -1) > "$O/bench_default.json"
(timeout 400 python bench.py --loss pce 2>/dev/null | tail -1) > "$O/bench_pce.json"
(timeout 300 python bench.py --loss ours_proposed 2>/dev/null | tail -1) > "$O/bench_ours.json"
(timeout 300 python bench.py --crf-radius 2 --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_crf_r2.json"
(timeout 400 python bench.py --loss mean_teacher 2>/dev/null | tail -1) > "$O/bench_mt.json"
(timeout 300 python bench.py --net unet --loss pce 2>/dev/null | tail -1) > "$O/bench_unet_pce.json"
(timeout 300 python bench.py --force-dp --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_forcedp.json"
(timeout 300 python bench.py --serial-decoders --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_serial.json"
for l in pce_tv pce_ms pce_entropy ce_dice ustm; do (timeout 300 python bench.py --loss $l --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_unet_$l.json"; done
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_serial" -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --serial-decoders > "$R/$O/bench_serial_under_rocprof.log" 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_default" -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
# counters: separate passes, never combined with a trace domain
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/$O/pmc_fetch" -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --serial-decoders --no-prof > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/$O/pmc_write" -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --serial-decoders --no-prof > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d "$R/$O/pmc_sq1" -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --serial-decoders --no-prof > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$R/$O/pmc_sq2" -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --serial-decoders --no-prof > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d "$R/$O/pmc_sq3" -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --serial-decoders --no-prof > /dev/null 2>&1
cd "$R"; rm -f "$O"/prof_*/*/*kernel_trace.csv
python tools/pmc_traffic.py "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_traffic.json" > /dev/null 2>&1; rm -rf "$O/pmc_fetch" "$O/pmc_write"
python tools/pmc_mfma.py "$O/pmc_sq1" "$O/pmc_sq2" "$O/pmc_sq3" > "$O/pmc_sq.md" 2>/dev/null; rm -rf "$O/pmc_sq1" "$O/pmc_sq2" "$O/pmc_sq3"
tail -3 "$O/pytest_gpu.log"; cat "$O/smoke.log"
for f in default pce ours crf_r2 mt unet_pce forcedp serial; do python - "$O/bench_$f.json" "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); r = d["roofline"] or {}
    print(sys.argv[2], d["value"], d["ms_per_step"], r.get("achieved"), r.get("frac"), r.get("issued_frac"), r.get("whole_step_issued_frac"), (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
python tools/check_rocprof_vs_bench.py "$O"/prof_serial/*/*kernel_stats.csv "$O/bench_serial_under_rocprof.log"
