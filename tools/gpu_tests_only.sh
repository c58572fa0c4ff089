#!/usr/bin/env bash
# all GPU tests except the 5-minute fp64 error-budget module (run separately, once per record)
set -u
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
(timeout 1200 python -m pytest tests -m gpu -q --tb=short --deselect tests/test_error_budget.py 2>&1 | tail -25) > "$O/pytest.log"
tail -4 "$O/pytest.log"
