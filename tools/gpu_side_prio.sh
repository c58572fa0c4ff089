#!/usr/bin/env bash
# the decoder side stream at normal / highest / lowest stream priority (experiments build, WSL_SIDE_PRIO), alternating on one box
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
for rep in 1 2 3; do for p in 0 1 -1; do
  v=$(WSL_SIDE_PRIO=$p python tools/bench_exp.py --no-cpu-baseline --no-split-record --no-pmc-refresh --no-prof --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['repeats']['values'])")
  echo "rep $rep WSL_SIDE_PRIO=$p $v"
done; done | tee "$O/side_prio.log"
