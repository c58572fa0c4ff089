#!/usr/bin/env bash
O=gpurun_out/r3e; mkdir -p $O
timeout 300 python -m pytest tests/test_net.py -q -m gpu -x -k "split" 2>&1 | grep -E "assert|Error|error|rel_err|^E " | head -30 > $O/fail.log; cat $O/fail.log
for a in 0 1 2 4 8 3 6 7 15; do echo "== WSL_SP_ABLATE=$a"; WSL_SP_ABLATE=$a timeout 120 python tools/sweep_layers_sp.py --exp --only-sp --few 2>&1 | grep "@" | awk -F'|' '{print $2, $4, $6}'; done > $O/ablate.log 2>&1; cat $O/ablate.log
