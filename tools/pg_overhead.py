#!/usr/bin/env python3
"""Does an initialised RCCL process group by itself change the single-GPU step time?  One subprocess per mode."""
import os
import socket
import subprocess
import sys
import time

MODES = ["nopg", "pg_eager", "pg_eager_q8", "pg_eager_q16", "pg_eager_presiding", "pg_eager_coll_dp_presiding"]
if len(sys.argv) > 1:
    mode = sys.argv[1]
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    torch.cuda.set_device(0)
    if "presiding" in mode:      # create the library's side stream BEFORE RCCL creates its own streams
        from wsl4mis_amd.networks.net_factory import net_factory
        m0 = net_factory("unet_cct", 1, 4)
        with torch.no_grad():
            m0(torch.rand(2, 1, 16, 16, device="cuda"))
        torch.cuda.synchronize()
    if mode != "nopg":
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        kw = {"device_id": torch.device("cuda", 0)} if "eager" in mode else {}
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, **kw)
        if "coll" in mode:
            t = torch.ones(1024, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
    from wsl4mis_amd.engine import TrainEngine
    from wsl4mis_amd.synthetic import batch
    torch.manual_seed(1)
    eng = TrainEngine("unet_cct", 1, 4, loss="pce_gatedcrf", force_dp=mode.endswith("_dp"))
    x, lab = batch(64, 256, 256, 3, torch.device("cuda", 0))
    for _ in range(5):
        eng.step(x, lab, 0.4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.step(x, lab, 0.4)
    torch.cuda.synchronize()
    print(f"MODE {mode}: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms/step", flush=True)
    if mode != "nopg":
        dist.destroy_process_group()
else:
    for rep in range(1):
        for m in MODES:
            env = dict(os.environ)
            if "_q8" in m:
                env["GPU_MAX_HW_QUEUES"] = "8"
            if "_q16" in m:
                env["GPU_MAX_HW_QUEUES"] = "16"
            r = subprocess.run([sys.executable, os.path.abspath(__file__), m], capture_output=True, text=True, timeout=200, env=env)
            print([l for l in r.stdout.splitlines() if l.startswith("MODE")] or r.stderr[-300:], flush=True)
