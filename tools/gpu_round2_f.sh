#!/usr/bin/env bash
# packed-add Winograd transforms + XCD-aware channel-block placement: parity, probe, per-layer and whole-step A/B
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
timeout 900 python -m pytest tests/test_ops_conv.py tests/test_net.py -m gpu -x -q > "$O/pytest_conv_net.log" 2>&1; echo "pytest rc=$?" >> "$O/pytest_conv_net.log"
tail -3 "$O/pytest_conv_net.log"
timeout 300 python tools/mfma_ceiling.py > "$O/mfma_ceiling.log" 2>&1; tail -14 "$O/mfma_ceiling.log"
WSL_EXP_LIB=old timeout 300 python tools/sweep_layers.py > "$O/sweep_old.md" 2>&1
WSL_WGRAD_XCD=0 WSL_WINO_XCD_Y=0 timeout 300 python tools/sweep_layers.py > "$O/sweep_pk.md" 2>&1
timeout 300 python tools/sweep_layers.py > "$O/sweep_pk_xcd.md" 2>&1
cat "$O"/sweep_*.md
for v in old pk pk_xcd; do
  case $v in old) e="WSL_EXP_LIB=old";; pk) e="WSL_WGRAD_XCD=0 WSL_WINO_XCD_Y=0";; *) e="";; esac
  env $e timeout 300 python tools/bench_exp.py --steps 20 --warmup 5 --no-cpu-baseline > "$O/bench_$v.json" 2> "$O/bench_$v.err"
  python - "$O/bench_$v.json" $v <<'P'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); print(sys.argv[2], d['value'], d['ms_per_step'])
except Exception as ex: print(sys.argv[2], 'failed', ex)
P
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$O/bench_product.json" 2> "$O/bench_product.err"; tail -c 600 "$O/bench_product.json"
