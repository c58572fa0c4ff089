#!/usr/bin/env bash
# round 3, call c: split-path tests + bench lines (f32 headline and split record) on one box
O=gpurun_out/r3c; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_convsp.py tests/test_net.py -q -m gpu -x 2>&1 | tail -15 > $O/pytest_sp.log; cat $O/pytest_sp.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err; python - <<'PY'
import json
for f in ("bench_f32",):
    try:
        d = json.loads(open(f"gpurun_out/r3c/{f}.json").read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["last_losses"])
    except Exception as e: print(f, "failed", e)
PY
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --conv-precision split_f16x3 > $O/bench_split.json 2> $O/bench_split.err; tail -3 $O/bench_split.err; python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r3c/bench_split.json").read().strip().splitlines()[-1]); print("split", d["value"], d["ms_per_step"], d["last_losses"]); print(json.dumps(d["roofline"]["kernels"], indent=1))
except Exception as e: print("split failed", e)
PY
