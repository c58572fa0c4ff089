#!/usr/bin/env bash
mkdir -p gpurun_out/r3w
timeout 25 python tools/ab_split_fullsize.py 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3w/ab_product.log
(timeout 40 python bench.py --conv-precision split_f16x3 --no-cpu-baseline 2>/dev/null | tail -1) > gpurun_out/r3w/bench_split.json; cut -c1-400 gpurun_out/r3w/bench_split.json
