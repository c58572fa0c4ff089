#!/usr/bin/env bash
O=gpurun_out/r3f; mkdir -p $O
timeout 300 python -m pytest tests/test_ops_convsp.py tests/test_net.py -q -m gpu 2>&1 | tail -40 > $O/pytest.log; grep -E "passed|failed|Error|assert " $O/pytest.log | head -20
timeout 250 python tools/sweep_layers_sp.py --dec --only-sp > $O/sweep_sp.log 2>&1; cat $O/sweep_sp.log
for a in 1 2 4 8 7 15; do echo "== WSL_SP_ABLATE=$a"; WSL_SP_ABLATE=$a timeout 120 python tools/sweep_layers_sp.py --exp --only-sp --few 2>&1 | grep "@" | awk -F'|' '{print $2, $4, $6}'; done > $O/ablate.log 2>&1; cat $O/ablate.log
