#!/usr/bin/env bash
# End-of-round record on the GPU box: tests (incl. the fp64 error budget and the reproducibility / concurrency guards), smoke, the bench
# compositions of BASELINE.json's configs (every default line carries the nested split-precision record), rocprofv3 summaries of both
# precisions, PMC traffic of both (merged into one file) + SQ passes of both steps.
#   bash tools/record_round.sh gpurun_out/<tag>          (via gpurun; about 15 GPU-minutes)
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
rm -f gpurun_out/labelmap_rates.jsonl gpurun_out/fullsize_error_budget.json
(WSL_FP64_BUDGET=1 timeout 1500 python -m pytest tests -m gpu -q --tb=short -s 2>&1 | grep -v "^iteration" | tail -60) > "$O/pytest_gpu.log"
cp gpurun_out/labelmap_rates.jsonl gpurun_out/fullsize_error_budget*.json gpurun_out/fullsize_replayed_decisions*.json "$O"/ 2>/dev/null
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) > "$O/smoke.log"
(timeout 900 python bench.py 2>"$O/bench_stderr.log" | tail -1) > "$O/bench_default.json"
(timeout 400 python bench.py --loss pce --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_pce.json"
(timeout 300 python bench.py --loss ours_proposed --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_ours.json"
(timeout 300 python bench.py --crf-radius 2 --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_crf_r2.json"
(timeout 400 python bench.py --loss mean_teacher --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_mt.json"
(timeout 300 python bench.py --net unet --loss pce --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_unet_pce.json"
(timeout 300 python bench.py --serial-decoders --no-cpu-baseline --no-split-record 2>/dev/null | tail -1) > "$O/bench_serial.json"
(timeout 300 python bench.py --path modules --steps 20 --warmup 5 2>/dev/null | tail -1) > "$O/bench_modules.json"
(timeout 300 python bench.py --force-dp --no-cpu-baseline --no-split-record 2>/dev/null | tail -1) > "$O/bench_forcedp.json"
for l in pce_tv pce_ms pce_entropy; do (timeout 300 python bench.py --loss $l --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_unet_$l.json"; done
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-split-record --no-pmc-refresh --repeats 1"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_f32_serial" -- $B --steps 10 --warmup 3 --serial-decoders > "$R/$O/bench_serial_under_rocprof.log" 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_split_serial" -- $B --steps 10 --warmup 3 --serial-decoders --conv-precision split_f16x3 > "$R/$O/bench_split_serial_under_rocprof.log" 2>/dev/null
# counters: separate passes, never combined with a trace domain
for prec in f32 split_f16x3; do for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --output-format csv -d "$R/$O/pmc_${prec}_$c" -- $B --steps 2 --warmup 1 --serial-decoders --no-prof --conv-precision $prec > /dev/null 2>&1
done; done
for prec in f32 split_f16x3; do
  S="$B --steps 2 --warmup 1 --serial-decoders --no-prof --conv-precision $prec"
  t=${prec%%_*}
  timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d "$R/$O/pmc_sq1_$t" -- $S > /dev/null 2>&1
  timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$R/$O/pmc_sq2_$t" -- $S > /dev/null 2>&1
  timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d "$R/$O/pmc_sq3_$t" -- $S > /dev/null 2>&1
done
cd "$R"; rm -f "$O"/prof_*/*/*kernel_trace.csv
python tools/pmc_traffic.py "$O/pmc_f32_FETCH_SIZE" "$O/pmc_f32_WRITE_SIZE" "$O/pmc_traffic_f32.json" > /dev/null 2>&1
python tools/pmc_traffic.py "$O/pmc_split_f16x3_FETCH_SIZE" "$O/pmc_split_f16x3_WRITE_SIZE" "$O/pmc_traffic_split.json" > /dev/null 2>&1
python - "$O" <<'PY'
import json, sys
o = sys.argv[1]
a, b = json.load(open(o + "/pmc_traffic_f32.json")), json.load(open(o + "/pmc_traffic_split.json"))
a["source"] += "; the split-precision kernels (conv_sp_kernel, wgrad_sp_kernel) from the same passes over `--conv-precision split_f16x3`"
for k, v in b["kernels"].items():
    if "_sp_" in k:
        a["kernels"][k] = v
json.dump(a, open(o + "/pmc_traffic.json", "w"), indent=1)
PY
rm -rf "$O"/pmc_f32_* "$O"/pmc_split_f16x3_*
for t in f32 split; do python tools/pmc_mfma.py "$O/pmc_sq1_$t" "$O/pmc_sq2_$t" "$O/pmc_sq3_$t" > "$O/pmc_sq_$t.md" 2>/dev/null; rm -rf "$O/pmc_sq1_$t" "$O/pmc_sq2_$t" "$O/pmc_sq3_$t"; done
tail -3 "$O/pytest_gpu.log"; cat "$O/smoke.log"
for f in default pce ours crf_r2 mt unet_pce serial unet_pce_tv unet_pce_ms unet_pce_entropy; do python - "$O/bench_$f.json" "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read()); r = d["roofline"] or {}; s = d.get("split_f16x3") or {}
    print(sys.argv[2], d["value"], d["ms_per_step"], r.get("achieved"), r.get("frac"), "split:", s.get("value"), s.get("ms_per_step"), (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
python tools/check_rocprof_vs_bench.py "$O"/prof_f32_serial/*/*kernel_stats.csv "$O/bench_serial_under_rocprof.log"
python tools/check_rocprof_vs_bench.py "$O"/prof_split_serial/*/*kernel_stats.csv "$O/bench_split_serial_under_rocprof.log"
