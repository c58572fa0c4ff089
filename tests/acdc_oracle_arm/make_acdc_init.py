#!/usr/bin/env python3
"""The initial state the oracle arm of profiles/r3_acdc_short_schedule.md started from (oracle_acdc_short.py next to this file: torch.manual_seed(seed),
then nn.Conv2d / nn.BatchNorm2d default initialisation in construction order), saved as a state_dict the HIP arm loads with --resume.
Build container only (imports the oracle's layout).     python tests/acdc_oracle_arm/make_acdc_init.py 2022 11"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle_acdc_short import default_init  # noqa: E402

for seed in sys.argv[1:]:
    torch.manual_seed(int(seed))
    sd = {k: v.detach().clone() for k, v in default_init("unet").items()}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools", "exp", f"acdc_init_unet_seed{seed}.pth")
    torch.save(sd, out)
    print(out, sum(v.numel() for v in sd.values()), "values")
