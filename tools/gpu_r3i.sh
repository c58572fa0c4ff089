#!/usr/bin/env bash
O=gpurun_out/r3i; mkdir -p $O
for a in "8 128 128 32 32 bn" "32 64 64 64 64 bn" "8 128 128 16 32 bn" "8 128 128 64 32 bn"; do python tools/debug_sp_case.py $a 2>&1 | grep -E "max err" | cut -c1-150; done
timeout 900 python -m pytest tests/test_ops_convsp.py tests/test_net.py -q -m gpu 2>&1 | tail -12 > $O/pytest.log; grep -E "passed|failed" $O/pytest.log
timeout 250 python tools/sweep_layers_sp.py --dec > $O/sweep_sp.log 2>&1; cat $O/sweep_sp.log
