#!/usr/bin/env python3
"""bench.py on the experiments build (tools/exp/libwslhip_exp.so) so that WSL_* tuning knobs take effect for a whole step:
   WSL_WGRAD_XCD=0 python tools/bench_exp.py --steps 20 --warmup 5 --no-cpu-baseline
Numbers from this script are tuning evidence only; records come from bench.py on the product library."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import explib  # noqa: E402

explib.use()
sys.path.insert(0, explib.ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    bench.main()
