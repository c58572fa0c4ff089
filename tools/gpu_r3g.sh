#!/usr/bin/env bash
O=gpurun_out/r3g; mkdir -p $O
timeout 300 python -m pytest tests/test_ops_convsp.py tests/test_net.py -q -m gpu 2>&1 | tail -40 > $O/pytest.log; grep -E "passed|failed|Error|assert " $O/pytest.log | head -20
timeout 250 python tools/sweep_layers_sp.py --dec > $O/sweep_sp.log 2>&1; cat $O/sweep_sp.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --conv-precision split_f16x3 > $O/bench_split.json 2> $O/bench_split.err; tail -3 $O/bench_split.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r3g/bench_split.json").read().strip().splitlines()[-1]); print("split", d["value"], d["ms_per_step"], d["last_losses"]); print(d["roofline"]["measured"])
    for k,v in d['kernels'].items(): print(k, v['calls'], v['ms'], v['avg_us'])
except Exception as e: print("split failed", e)
PY
