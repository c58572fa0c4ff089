#!/usr/bin/env bash
# mid-round record on the final split-path tree: tests, smoke, both bench lines, rocprof + SQ counters of the split step, ACDC short schedule
set -u
O=gpurun_out/r3s; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
(timeout 1200 python -m pytest tests -m gpu -q --tb=short --deselect tests/test_error_budget.py 2>&1 | tail -30) > "$O/pytest_gpu.log"
(timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2) > "$O/smoke.log"
(timeout 500 python bench.py 2>"$O/bench_stderr.log" | tail -1) > "$O/bench_default.json"
(timeout 300 python bench.py --conv-precision split_f16x3 --no-cpu-baseline 2>/dev/null | tail -1) > "$O/bench_split.json"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$O/prof_split" -- python "$R/bench.py" --conv-precision split_f16x3 --steps 10 --warmup 3 --no-cpu-baseline --serial-decoders > "$R/$O/bench_split_under_rocprof.log" 2>/dev/null
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d "$R/$O/pmc_sq1" -- python "$R/bench.py" --conv-precision split_f16x3 --steps 2 --warmup 1 --no-cpu-baseline --serial-decoders --no-prof > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$R/$O/pmc_sq2" -- python "$R/bench.py" --conv-precision split_f16x3 --steps 2 --warmup 1 --no-cpu-baseline --serial-decoders --no-prof > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d "$R/$O/pmc_sq3" -- python "$R/bench.py" --conv-precision split_f16x3 --steps 2 --warmup 1 --no-cpu-baseline --serial-decoders --no-prof > /dev/null 2>&1
cd "$R"; rm -f "$O"/prof_*/*/*kernel_trace.csv
python tools/pmc_mfma.py "$O/pmc_sq1" "$O/pmc_sq2" "$O/pmc_sq3" > "$O/pmc_sq_split.md" 2>/dev/null; rm -rf "$O/pmc_sq1" "$O/pmc_sq2" "$O/pmc_sq3"
tail -3 "$O/pytest_gpu.log"; cat "$O/smoke.log"; cat "$O/bench_default.json" "$O/bench_split.json" | cut -c1-700
bash tools/acdc_short_hip.sh "$O/acdc"
