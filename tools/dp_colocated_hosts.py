#!/usr/bin/env python3
"""Host-side enqueue cost of the data-parallel step when EIGHT rank processes share one node (SURVEY 8e's stated risk: at 8 GPUs the
step is ~15 ms of device time, so >= 6x scaling is a host-overhead / straggler question, not a bandwidth one).  The builder has one
GPU: the eight processes all use cuda:0 (the device time is then shared and meaningless), the process group is gloo, and what is
measured is what each rank's Python thread spends ENQUEUEING one step -- forward, loss head, split backward, the two bucketed
all-reduce calls, SGD -- while seven other ranks do the same on the same host, against one process alone.
    python tools/dp_colocated_hosts.py [world]          (record: profiles/r4_dp_colocated_hosts.md)"""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, prec, q):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    torch.cuda.set_device(0)
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from wsl4mis_amd.engine import TrainEngine
    from wsl4mis_amd.synthetic import batch
    torch.manual_seed(1)
    eng = TrainEngine("unet_cct", 1, 4, loss="pce_gatedcrf", conv_precision=prec)
    x, lab = batch(8, 256, 256, 3 + rank, torch.device("cuda", 0))       # 8 slices per rank: the launch COUNT per step is that of batch 64
    for _ in range(3):
        eng.step(x, lab, 0.4)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    n, host = 15, 0.0
    t0 = time.perf_counter()
    for _ in range(n):
        t = time.perf_counter()
        eng.step(x, lab, 0.4)
        host += time.perf_counter() - t            # returns when the step is enqueued (gloo all-reduce of CUDA tensors: see note)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    q.put((rank, 1e3 * host / n, 1e3 * wall / n))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run(world, prec):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, port, prec, q)) for r in range(world)]
    for p in ps:
        p.start()
    rows = sorted(q.get(timeout=600) for _ in ps)
    for p in ps:
        p.join(60)
    return rows


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    print(f"host cores: {os.cpu_count()}")
    for prec in ("f32", "split_f16x3"):
        one = run(1, prec)
        many = run(world, prec)
        print(f"[{prec}] 1 process: host enqueue {one[0][1]:.2f} ms/step (step wall {one[0][2]:.2f} ms at batch 8)")
        print(f"[{prec}] {world} processes on one GPU (gloo): host enqueue per rank min {min(r[1] for r in many):.2f} / max {max(r[1] for r in many):.2f} ms/step; "
              f"wall per step {max(r[2] for r in many):.2f} ms (device shared by {world})")
    print("note: with gloo the all-reduce of device tensors copies through the host and blocks the calling thread until the bucket "
          "is reduced, so 'host enqueue' here INCLUDES waiting for the backward kernels in front of each bucket -- an upper bound on what "
          "the RCCL route (asynchronous, on the comm stream) costs the host.")
