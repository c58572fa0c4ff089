#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): training slices/sec of the dual-branch UNet (unet_cct) with partial-CE +
GatedCRF on synthetic 256x256 4-class scribble slices, batch 64 per GPU, data-parallel over N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = masks -> forward -> loss head (+GatedCRF) -> backward -> gradient all-reduce -> SGD, i.e. one optimiser
step on one batch of 64 slices per GPU, inputs resident in HBM.  Prints ONE JSON line on rank 0 with the `roofline`
(HIP events around every launch of the dominant kernel family, recorded on the launch stream) and, at N=1, the
`cpu_baseline` (the oracle's torch-CPU restatement of the same step, timed on this box's host cores).

unet_cct runs its two decoders on two streams (+5 % step rate; mean teacher: the teacher's forward), so in the timed region launches of the dominant kernel
overlap each other and a per-launch duration no longer measures the kernel.  The `roofline` object is therefore taken
from a short second segment of the same workload in the same process with the decoders serialised
(`wsl_debug_net_concurrent(0)`), where launches do not overlap; the timed region's own (overlapping) per-launch figures
are reported next to it as `roofline.timed_region_overlapped`.  `--serial-decoders` serialises the timed region too.
"""
import argparse
import ctypes as C
import json
import os
import random
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="slices per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--loss", default="pce_gatedcrf", choices=["pce_gatedcrf", "ours_proposed", "pce", "mean_teacher", "ustm", "pce_tv", "pce_ms", "pce_entropy", "ce_dice"])
    ap.add_argument("--crf-radius", type=int, default=5, help="reference default 5 (11x11); 2 = the 5x5 of BASELINE.json")
    ap.add_argument("--net", default="unet_cct", choices=["unet_cct", "unet"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true")
    ap.add_argument("--prof-timed", action="store_true",
                    help="also record per-launch events in the (overlapped) timed region -> roofline.timed_region_overlapped")
    ap.add_argument("--serial-decoders", action="store_true",
                    help="run the two decoders of unet_cct on ONE stream in the timed region too (per-launch timings do "
                         "not overlap; the command behind profiles/*serial* rocprofv3 summaries)")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(host cores, 32)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) run the CPU leg and print its JSON")
    return ap.parse_args()


def cpu_baseline(args):
    """The oracle (torch-CPU restatement of the reference step) on a bounded sample of the same workload."""
    from oracle import torch_ref as R
    n_thr = args.cpu_threads or min(os.cpu_count() or 1, 32)
    torch.set_num_threads(n_thr)
    B, S = args.cpu_batch, args.size
    g = torch.Generator().manual_seed(1)
    sd = {}
    for k, shp in R.state_layout(args.net, 1, 4):
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith("running_var") or (k.split(".")[-2] in ("1", "5") and k.endswith("weight")):
            sd[k] = torch.ones(shp)
        elif len(shp) == 4:
            sd[k] = torch.randn(shp, generator=g) * (1.0 / (shp[1] * shp[2] * shp[3]) ** 0.5)
        else:
            sd[k] = torch.zeros(shp)
    tr = R.RefTrainer(sd, args.net)
    from wsl4mis_amd.synthetic import scribble_labels
    x = torch.rand((B, 1, S, S), generator=g)
    lab = torch.from_numpy(scribble_labels(B, S, S, 5))
    em = [(torch.rand((B, 16 << l, S >> l, S >> l), generator=g) >= R.DROP[l]).to(torch.uint8) for l in range(5)]
    cm = [(torch.rand((B, 16 << l), generator=g) >= 0.5).float() * 2 for l in range(5)]
    crf = args.crf_radius if args.loss == "pce_gatedcrf" else None
    if args.loss in ("pce", "mean_teacher"):
        raise SystemExit("cpu baseline is built for ours_proposed and pce_gatedcrf")
    tr.step(x, lab, 0.4, em, cm, crf)                      # warm-up (thread pool, oneDNN primitives)
    iters, t0 = 0, time.perf_counter()
    while iters < max(2, args.cpu_iters) or (time.perf_counter() - t0 < 15.0 and iters < 12):   # ~15-30 s of CPU work
        tr.step(x, lab, 0.4, em, cm, crf)
        iters += 1
    dt = time.perf_counter() - t0
    return {"value": round(B * iters / dt, 3), "unit": "slices/s", "cores": n_thr, "kind": "port",
            "sample": f"oracle/torch_ref.py RefTrainer (stock torch CPU ops), {args.net} {args.loss}"
                      + (f" r={args.crf_radius}" if crf else "") + f", batch {B} at {S}x{S}, 1 warm-up + {iters} timed "
                      f"steps, {n_thr} threads of {os.cpu_count()} host cores"}


def cpu_baseline_subprocess(args):
    """Run the CPU leg in its own interpreter with a hard time limit, so a slow host can never cost the GPU result."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--loss", args.loss, "--net", args.net,
           "--size", str(args.size), "--crf-radius", str(args.crf_radius), "--cpu-batch", str(args.cpu_batch),
           "--cpu-iters", str(args.cpu_iters), "--cpu-threads", str(args.cpu_threads)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "slices/s", "cores": 0, "kind": "port", "sample": "failed: " + r.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "slices/s", "cores": 0, "kind": "port", "sample": "timed out after 240 s"}


def main():
    args = parse()
    if args.loss in ("mean_teacher", "ustm", "pce_tv", "pce_ms", "pce_entropy", "ce_dice"):
        args.net, args.no_cpu_baseline = "unet", True      # config 4 (and USTM): single-decoder student + EMA teacher
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args)), flush=True)
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path in the product)")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from wsl4mis_amd import _lib
    from wsl4mis_amd.engine import TrainEngine
    from wsl4mis_amd.synthetic import batch
    dev = torch.device("cuda", local)
    torch.manual_seed(2022)                                   # same initial weights on every rank
    eng = TrainEngine(args.net, 1, 4, base_lr=0.01, max_iterations=60000, loss=args.loss, crf_radius=args.crf_radius)
    torch.manual_seed(2022 + 1000 * rank)                     # different dropout masks / data per rank
    x, lab = batch(args.batch, args.size, args.size, 2022 + rank, dev)
    random.seed(2022)                                         # identical beta stream on all ranks
    L = _lib.lib()
    if args.serial_decoders:
        L.wsl_debug_net_concurrent(0)
        eng.concurrent = False
    for _ in range(args.warmup):
        eng.step(x, lab, random.random() + 1e-10)
    overlapped = (args.net == "unet_cct" or args.loss == "mean_teacher") and not args.serial_decoders and \
        os.environ.get("WSL_NET_CONCURRENT") != "0"
    # per-launch HIP events cost ~2 % of the step rate: when the roofline comes from its own serialised segment anyway
    # (overlapped run) the timed region stays uninstrumented unless --prof-timed asks for its overlapping figures too
    prof_timed = not args.no_prof and (not overlapped or args.prof_timed)
    if prof_timed:
        L.wsl_prof_enable(1)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step(x, lab, random.random() + 1e-10)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    losses = eng.losses()

    def report():
        rows = (_lib.WslProfRow * 8)()
        L.wsl_prof_report(rows, 8)
        L.wsl_prof_enable(0)
        fams = {}
        for r in rows:
            if r.calls:
                fams[r.name.decode()] = {"calls": int(r.calls), "ms": round(r.ms, 3),
                                         "avg_us": round(1e3 * r.ms / r.calls, 2),
                                         "tflops": round(r.flops / (r.ms * 1e-3) / 1e12, 2) if r.flops else None,
                                         "algo_GBps": round(r.bytes / (r.ms * 1e-3) / 1e9, 1)}
        return rows, fams

    def roofline_of(rows, nsteps):
        conv = [r for r in rows if r.calls and r.flops > 0]
        if not conv:
            return None
        # one entry per kernel family: the direct MFMA convolution (1x1 layers, first conv, 4-channel classifiers), the
        # Winograd F(2x2,3x3) convolution (every other 3x3 layer, forward + data-gradient launches) and the weight gradient
        # (Winograd form for the 3x3 layers; the 1x1 / first / classifier layers ride in the same event family).  The
        # dominant one = most time per step.
        # Flops are the ALGORITHMIC ones (direct convolution: 2 * 9 * Ci * Co per pixel) for all of them -- the Winograd
        # kernel issues 2.25x fewer matrix instructions for the same result, so its fraction can pass what a direct kernel
        # could reach.
        kern = {"conv_mfma2l_kernel (fwd + data-gradient launches)": [r for r in rows[:2] if r.calls],
                "conv_wino2_kernel (fwd + data-gradient launches)": [r for r in rows[6:8] if r.calls],
                "wgrad_wino_kernel": [r for r in rows[2:3] if r.calls]}
        per_kernel = {k: {"ms_per_step": round(sum(r.ms for r in v) / nsteps, 3), "launches": int(sum(r.calls for r in v)),
                          "avg_launch_us": round(1e3 * sum(r.ms for r in v) / sum(r.calls for r in v), 2),
                          "achieved": round(sum(r.flops for r in v) / (sum(r.ms for r in v) * 1e-3) / 1e12, 2),
                          "frac": round(sum(r.flops for r in v) / (sum(r.ms for r in v) * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
                      for k, v in kern.items() if v}
        name, grp = max(((k, v) for k, v in kern.items() if v), key=lambda kv: sum(r.ms for r in kv[1]))
        ms, fl, calls = sum(r.ms for r in grp), sum(r.flops for r in grp), sum(r.calls for r in grp)
        ach = fl / (ms * 1e-3) / 1e12
        traffic = None     # HBM bytes per launch of that kernel family from the committed rocprofv3 PMC passes
        try:               # (FETCH_SIZE / WRITE_SIZE, separate runs of this same command -- tools/pmc_traffic.py)
            tfile = next(f for f in ("r1z_pmc_traffic.json", "r1t_pmc_traffic.json", "r1h_pmc_traffic.json")
                         if os.path.exists(os.path.join(ROOT, "profiles", f)))
            with open(os.path.join(ROOT, "profiles", tfile)) as fh:
                tj = json.load(fh)["kernels"]
            keys = ("conv_wino2_kernel", "conv_wino_kernel") if name.startswith("conv_wino") else \
                   ("conv_mfma2l_kernel", "conv_mfma2_kernel") if name.startswith("conv") else \
                   ("wgrad_wino_kernel", "wgrad_mfma2s_kernel", "wgrad_mfma2l_kernel", "wgrad_mfma2_kernel")
            nl = sum(tj[k]["launches_sampled"] for k in keys if k in tj)
            traffic = {"hbm_bytes_per_launch": sum(tj[k]["hbm_bytes_per_launch"] * tj[k]["launches_sampled"]
                                                   for k in keys if k in tj) / nl,
                       "algorithmic_bytes_per_launch": sum(r.bytes for r in grp) / calls,
                       "source": f"profiles/{tfile} (rocprofv3 --pmc, FETCH_SIZE x2 per MI355X_MICROARCH.md)"}
        except (OSError, KeyError, ValueError, ZeroDivisionError, StopIteration):
            pass
        return {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,     # HBM bytes per launch (PMC)
                "traffic_detail": traffic,
                "launches": int(calls), "avg_launch_us": round(1e3 * ms / calls, 2), "flops_per_launch": fl / calls,
                "flops_counted": "algorithmic (direct convolution)", "kernels": per_kernel,
                "all_mfma_kernels_tflops": round(sum(r.flops for r in conv) / (sum(r.ms for r in conv) * 1e-3) / 1e12, 2),
                "all_mfma_kernels_ms_per_step": round(sum(r.ms for r in conv) / nsteps, 3)}

    roof, fams = None, {}
    if not args.no_prof:
        timed = None
        if prof_timed:
            rows, fams = report()
            roof = roofline_of(rows, args.steps)
        if overlapped:
            # second segment, decoders serialised: launches of the dominant kernel no longer overlap each other
            if roof:
                timed = {k: roof[k] for k in ("achieved", "frac", "launches", "avg_launch_us", "all_mfma_kernels_tflops")}
            seg = max(1, min(args.steps, 5))
            L.wsl_debug_net_concurrent(0)
            eng.concurrent = False
            eng.step(x, lab, random.random() + 1e-10)
            L.wsl_prof_enable(1)
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(seg):
                eng.step(x, lab, random.random() + 1e-10)
            torch.cuda.synchronize()
            seg_ms = 1e3 * (time.perf_counter() - ts) / seg
            rows2, fams = report()
            L.wsl_debug_net_concurrent(1)
            eng.concurrent = True
            roof = roofline_of(rows2, seg)
            if roof:
                roof["measured"] = (f"{seg} extra steps of the same workload right after the timed region, two decoder streams "
                                    f"serialised ({round(seg_ms, 3)} ms/step incl. event overhead); in the timed region "
                                    "launches of this kernel overlap each other, so a per-launch duration does not measure it")
                if timed:
                    roof["timed_region_overlapped"] = timed
    if rank == 0:
        gflop = 28.98 if args.net == "unet_cct" else 17.68     # conv-stack training GFLOP/slice
        if args.loss == "mean_teacher":
            gflop += 5.899                                     # + the teacher's forward
        if args.loss == "ustm":
            gflop += 9 * 5.899                                 # + 1 + 4 x 2 teacher forwards per student slice
        value = args.batch * world * args.steps / dt
        out = {"metric": "training slices/sec (256x256, bs64, unet_cct pCE+GatedCRF)", "value": round(value, 2),
               "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{args.net} {args.loss}" + (f" r={args.crf_radius}" if args.loss == "pce_gatedcrf" else "")
                          + f", {args.size}x{args.size}x1 4-class synthetic scribble slices, batch {args.batch}/GPU, SGD+poly LR",
                          "global_batch": args.batch * world, "parallelism": f"dp{world}", "crf_radius": args.crf_radius},
               "whole_step_conv_mfma_frac": round(value / world * gflop * 1e9 / (PEAK_F32_MFMA_TFLOPS * 1e12), 4),
               "roofline": roof, "kernels": fams, "last_losses": {k: round(v, 5) for k, v in losses.items()}}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_subprocess(args)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
