#!/usr/bin/env bash
# the measured-but-not-yet-shipped kernel switches (-DWSL_SP_PIPE=1, -DWSL_WINO2R_UNTRACKED=1) as ONE library: the GPU tests that exercise
# the kernels they touch (ops, nets, kernel pairs on two streams, whole-step reproducibility in both precisions), then both steps
O="$1"; R="${GRAFT_REPO_ROOT:-$(pwd)}"; mkdir -p "$R/$O"; cd "$R"
python -c "
import os, sys
from wsl4mis_amd import _lib
_lib.LIB_PATH = os.path.abspath('tools/exp/libwslhip_next.so')
import pytest
sys.exit(pytest.main(['tests/test_ops_conv.py', 'tests/test_ops_convsp.py', 'tests/test_concurrency.py', 'tests/test_net.py', 'tests/test_fullsize.py', '-x', '-q', '-m', 'gpu']))" 2>&1 | tail -3 | tee -a "$O/next.log"
for prec in f32 split_f16x3; do for v in product next; do
  lib=""; [ "$v" != product ] && lib="--lib tools/exp/libwslhip_$v.so"
  python bench.py --conv-precision $prec --steps 40 --warmup 10 --no-split-record --no-cpu-baseline $lib 2>/dev/null | tail -1 | python -c "
import json, sys
d = json.loads(sys.stdin.read()); print('$prec', '$v', d['value'], d['ms_per_step'])" | tee -a "$O/next.log"
done; done
