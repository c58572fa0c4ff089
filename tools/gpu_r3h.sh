#!/usr/bin/env bash
O=gpurun_out/r3h; mkdir -p $O
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -4 $O/smoke.log
timeout 1500 python -m pytest tests/test_error_budget.py -q -m gpu 2>&1 | tail -30 > $O/pytest_budget.log; cat $O/pytest_budget.log
cp gpurun_out/fullsize_*.json $O/ 2>/dev/null
