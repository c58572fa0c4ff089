#!/usr/bin/env bash
# forcezero build of wsl_convsp: repeated runs of the configurations that failed, op tests, speed
mkdir -p gpurun_out/r3k
for a in "8 128 128 16 32 bn" "32 64 64 64 64 bn" "8 128 128 64 32 bn" "48 32 32 128 128 bn" "16 256 256 16 16 bn" "16 128 128 64 32"; do
  bad=0; for i in 1 2 3 4 5 6 7 8; do r=$(python tools/debug_sp_case.py $a 2>&1 | grep -E "max err" | sed 's/.*bad elements \([0-9]*\) of.*/\1/'); [ "$r" != "0" ] && bad=$((bad+1)); done; echo "product  [$a]: $bad of 8 runs with bad elements"
done 2>&1 | tee gpurun_out/r3k/loops.log
for i in 1 2 3; do timeout 600 python -m pytest tests/test_ops_convsp.py -m gpu -x -q 2>&1 | tail -3; done | tee gpurun_out/r3k/ops.log
timeout 300 python tools/sweep_layers_sp.py --dec 2>&1 | tee gpurun_out/r3k/sweep.log
timeout 900 python -m pytest tests/test_error_budget.py tests/test_net.py -m gpu -x -q -s 2>&1 | tail -40 | tee gpurun_out/r3k/budget.log
