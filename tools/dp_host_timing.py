#!/usr/bin/env python3
"""Where does the data-parallel route lose time on ONE GPU?  Host-side durations of the engine's pieces in a 1-rank RCCL group
(is dist.all_reduce blocking the host until the comm stream reaches it?) next to the step time.  python tools/dp_host_timing.py"""
import os
import random
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.cuda.set_device(0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from wsl4mis_amd.engine import TrainEngine
from wsl4mis_amd.synthetic import batch

for force in (False, True):
    torch.manual_seed(1)
    eng = TrainEngine("unet_cct", 1, 4, loss="pce_gatedcrf", force_dp=force)
    x, lab = batch(64, 256, 256, 3, torch.device("cuda", 0))
    host = {"allreduce": 0.0, "bwd_phase": 0.0, "fb": 0.0, "opt": 0.0}
    if force:
        orig_ar, orig_bw = eng._allreduce, eng.model._run_backward

        def ar(flat):
            t = time.perf_counter(); orig_ar(flat); host["allreduce"] += time.perf_counter() - t

        def bw(*a, **k):
            t = time.perf_counter(); r = orig_bw(*a, **k); host["bwd_phase"] += time.perf_counter() - t; return r
        eng._allreduce, eng.model._run_backward = ar, bw
    for _ in range(5):
        eng.step(x, lab, 0.4)
    torch.cuda.synchronize()
    for k in host: host[k] = 0.0
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        t = time.perf_counter(); eng.forward_backward(x, lab, 0.4); host["fb"] += time.perf_counter() - t
        t = time.perf_counter(); eng.optimizer_step(); host["opt"] += time.perf_counter() - t
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"force_dp={force}: {1e3 * dt / n:.3f} ms/step; host enqueue of the 20 steps took {1e3 * t_enq / n:.3f} ms/step; per step host ms: "
          + ", ".join(f"{k} {1e3 * v / n:.3f}" for k, v in host.items()), flush=True)
dist.destroy_process_group()
