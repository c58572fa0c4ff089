#!/usr/bin/env python3
"""Times every built tile shape of the conv kernel on the layer shapes of the UNet (batch 64, 256x256 input) and prints the
best per shape.  Used to fill the static plan table in wsl_conv.hip (fwd_plan)."""
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
PLANS = [(8, 64, 16), (8, 64, 32), (8, 32, 16), (8, 32, 32), (8, 32, 64), (16, 16, 16), (16, 16, 32), (16, 16, 64)]
SHAPES = []
for lvl, (c, hw) in enumerate([(16, 256), (32, 128), (64, 64), (128, 32), (256, 16)]):
    SHAPES += [(c, c, hw), (2 * c, c, hw)]
    if lvl:
        SHAPES += [(c // 2, c, hw), (c, c // 2, hw), (c, 2 * c, hw) if False else (c, c, hw)]
SHAPES = sorted(set(SHAPES), key=lambda t: (-t[2], t[0], t[1]))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for ci, co, hw in SHAPES:
    res = []
    for th, tw, ct in PLANS:
        if hw % tw or hw % th or co % ct:
            continue
        env = dict(os.environ, WSL_CONV_PLAN=f"{th},{tw},{ct}")
        out = subprocess.run([sys.executable, os.path.join(here, "microbench_conv.py"), str(N), str(ci), str(co), str(hw), str(hw)],
                             env=env, capture_output=True, text=True).stdout
        us = [float(l.split("k3:")[1].split("us")[0]) for l in out.splitlines() if "k3:" in l]
        if us:
            res.append((us[0], (th, tw, ct)))
    res.sort()
    print(f"{ci:4d}->{co:4d} @{hw:3d}: " + "  ".join(f"{p[0]}x{p[1]}x{p[2]}:{u:6.1f}" for u, p in res), flush=True)
