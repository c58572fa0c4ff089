#!/usr/bin/env bash
mkdir -p gpurun_out/r3q
{
for a in "32 64 64 64 64 bn" "16 128 128 32 32 bn" "48 32 32 128 128 bn" "8 128 128 16 16 bn" "8 256 256 16 16 bn" "16 16 16 256 256 bn"; do echo "== wgrad $a"; timeout 300 python tools/debug_sp_wgrad.py $a 6 2>&1 | grep -E "bad elements"; done
for a in "8 128 128 16 32 bn" "32 64 64 64 64 bn" "8 128 128 64 32 bn" "48 32 32 128 128 bn" "16 256 256 16 16 bn" "16 128 128 32 32 bn" "64 16 16 256 256 bn"; do
  bad=0; for i in 1 2 3 4 5 6; do r=$(python tools/debug_sp_case.py $a 2>&1 | grep -E "max err" | sed 's/.*bad elements \([0-9]*\) of.*/\1/'); [ "$r" != "0" ] && bad=$((bad+1)); done; echo "conv [$a]: $bad of 6 runs with bad elements"
done
} 2>&1 | tee gpurun_out/r3q/loops.log
for i in 1 2; do timeout 300 python -m pytest tests/test_ops_convsp.py tests/test_net.py -m gpu -q --tb=line 2>&1 | tail -5; done | tee gpurun_out/r3q/ops.log
timeout 300 python tools/sweep_layers_sp.py --dec 2>&1 | tee gpurun_out/r3q/sweep.log
timeout 900 python -m pytest tests/test_error_budget.py -m gpu -x -q -s 2>&1 | tail -25 | tee gpurun_out/r3q/budget.log
