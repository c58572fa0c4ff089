#!/usr/bin/env bash
# round 4, conv_sp rework (weights' DMA ahead of the MFMA loop, LDS-only barriers, coefficient table, one epilogue per instantiation):
# correctness first, then the layer sweep of every build side by side on one box, then the step.
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
timeout 600 python -m pytest tests/test_ops_convsp.py tests/test_concurrency.py -x -q -m gpu > "$O/pytest_convsp.log" 2>&1; echo "pytest rc=$?" | tee -a "$O/pytest_convsp.log"
tail -3 "$O/pytest_convsp.log"
for v in product old db0 minw2; do
  lib=""; [ "$v" != product ] && lib="tools/exp/libwslhip_$v.so"
  echo "== $v" | tee -a "$O/sweep.log"
  WSL_LIB=$lib timeout 300 python tools/sweep_layers_sp.py --dec --only-sp 2>&1 | tee -a "$O/sweep.log" | tail -15
done
timeout 300 python bench.py --conv-precision split_f16x3 --steps 30 --warmup 10 --no-split-record --no-cpu-baseline > "$O/bench_split.json" 2> "$O/bench_split.err"; tail -c 600 "$O/bench_split.json"
