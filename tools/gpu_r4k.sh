#!/usr/bin/env bash
# EXPERIMENT (not in the product): operand reads of the next MFMA group pinned behind the MFMAs of the current one (-DWSL_SP_PIPE=1,
# blocks of <= 32 output channels): numerics of that build, layer sweep and the split step against the product, one box
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
python -c "
import os, sys
from wsl4mis_amd import _lib
_lib.LIB_PATH = os.path.abspath('tools/exp/libwslhip_pipe.so')
import pytest
sys.exit(pytest.main(['tests/test_ops_convsp.py', 'tests/test_concurrency.py', '-x', '-q', '-m', 'gpu']))" 2>&1 | tail -2 | tee -a "$O/pipe.log"
export SWEEP_BEST=3 SWEEP_REPS=30
for v in product pipe; do
  lib=""; [ "$v" != product ] && lib="tools/exp/libwslhip_$v.so"
  echo "== $v" | tee -a "$O/pipe.log"
  WSL_LIB=$lib timeout 300 python tools/sweep_layers_sp.py --dec --only-sp 2>&1 | grep "@\|sum" | cut -d'|' -f2,4,6 | tee -a "$O/pipe.log"
done
VARIANTS="product pipe" bash tools/gpu_r4j.sh "$O" | tee -a "$O/pipe.log"
