#!/usr/bin/env bash
# EXPERIMENT (not in the product): conv_wino2r_kernel -- the f32 data-gradient kernel -- with its LDS DMAs issued from inline assembly
# (-DWSL_WINO2R_UNTRACKED=1), so that hipcc does not wait for the NEXT chunk's DMA in front of THIS chunk's first LDS reads: numerics of
# that build, then the f32 step against the product on one box
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
python -c "
import os, sys
from wsl4mis_amd import _lib
_lib.LIB_PATH = os.path.abspath('tools/exp/libwslhip_w2r.so')
import pytest
sys.exit(pytest.main(['tests/test_ops_conv.py', 'tests/test_net.py', '-x', '-q', '-m', 'gpu']))" 2>&1 | tail -2 | tee -a "$O/w2r.log"
PREC=f32 VARIANTS="product w2r" bash tools/gpu_r4j.sh "$O" | tee -a "$O/w2r.log"
