#!/usr/bin/env bash
# 16-channel full-resolution layers: 8 x 64 x 16 (two accumulator sets, 2 workgroups/CU) vs 8 x 32 x 16 (one set, 3 per CU)
for t in 64 32; do echo "== WSL_WINO16_TILE=$t"; for c in "64 16 16 256 256" "64 32 16 256 256"; do
  WSL_WINO16_TILE=$t MB_WINO=1 python tools/microbench_conv.py $c 2>&1 | grep us
  WSL_WINO16_TILE=$t MB_WINO=1 MB_RAW=1 python tools/microbench_conv.py $c 2>&1 | grep us
done; done
for t in 64 32 64 32; do WSL_WINO16_TILE=$t python tools/bench_exp.py --steps 20 --warmup 5 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('tile', $t, d['value'], d['ms_per_step'])"; done
