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
(`wsl_net_concurrent(0)`), where launches do not overlap; the timed region's own (overlapping) per-launch figures
are reported next to it as `roofline.timed_region_overlapped`.  `--serial-decoders` serialises the timed region too.
"""
import argparse
import ctypes as C
import json
import os
import random
import sys
import time

# Before the HIP runtime loads: ROCclr multiplexes HIP streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues.  Once RCCL has
# created its streams, the decoder side stream of libwslhip.so lands on the SAME hardware queue as the main stream and the two
# decoders of unet_cct stop overlapping: measured 17.96 instead of 16.99 ms/step in a 1-rank RCCL group (tools/pg_overhead.py,
# profiles/r2_pg_overhead.md).  Eight queues keep them apart.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_F16_MFMA_TFLOPS = 2500.0    # same guide: dense f16 / bf16 MFMA peak (the split-precision record's matrix roofline)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # >= 50 timed steps after >= 10 warm-up steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--conv-precision", default="f32", choices=["f32", "split_f16x3"],
                    help="f32 (the headline) | split_f16x3: the split-precision conv path (f16 hi / lo operands, three MFMA "
                         "passes, fp32 accumulate) as the primary record (dtype f32-split-f16x3)")
    ap.add_argument("--no-split-record", action="store_true",
                    help="by default an f32 run is followed by the SAME workload on the split-precision conv path (own engine, own "
                         "warm-up, >= 20 timed steps, own serialised roofline segment), nested as \"split_f16x3\" in the one JSON line")
    ap.add_argument("--lib", default=None, help="A/B timing only: another hipcc build of libwslhip.so (e.g. tools/exp/libwslhip_prev.so) instead "
                    "of the in-tree product library; the line says so in config.library")
    ap.add_argument("--batch", type=int, default=64, help="slices per GPU")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--loss", default="pce_gatedcrf", choices=["pce_gatedcrf", "ours_proposed", "pce", "mean_teacher", "ustm", "pce_tv", "pce_ms", "pce_entropy", "ce_dice"])
    ap.add_argument("--crf-radius", type=int, default=5, help="reference default 5 (11x11); 2 = the 5x5 of BASELINE.json")
    ap.add_argument("--net", default="unet_cct", choices=["unet_cct", "unet"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true")
    ap.add_argument("--no-pmc-refresh", action="store_true", help="do not collect roofline.traffic in this run (then the newest committed "
                    "profiles/r*_pmc_traffic.json record is quoted, marked as such)")
    ap.add_argument("--pmc-refresh", action="store_true", help="(the default since round 5 when N=1) collect roofline.traffic IN THIS RUN -- two "
                    "rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; counters only, no trace domain) of this same command at "
                    "--steps 2 --warmup 1, aggregated by tools/pmc_traffic.py (adds about a minute)")
    ap.add_argument("--prof-timed", action="store_true",
                    help="also record per-launch events in the (overlapped) timed region -> roofline.timed_region_overlapped")
    ap.add_argument("--serial-decoders", action="store_true",
                    help="run the two decoders of unet_cct on ONE stream in the timed region too (per-launch timings do "
                         "not overlap; the command behind profiles/*serial* rocprofv3 summaries)")
    ap.add_argument("--force-dp", action="store_true", help="N=1 only: take the data-parallel route (split backward, bucketed RCCL "
                    "all-reduce in a 1-rank group on the comm stream) to measure its overhead on one GPU")
    ap.add_argument("--cpu-batch", type=int, default=4)
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = sweep 16 / 32 / 64 / 128 / all host cores and report the best")
    ap.add_argument("--cpu-one-batch", action="store_true", help="CPU leg: only --cpu-batch, not batch 16 as well")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) run the CPU leg and print its JSON")
    ap.add_argument("--path", default="engine", choices=["engine", "modules"],
                    help="engine (default, the headline): the fused TrainEngine.  modules: the DROP-IN module path INTEGRATION.md section 2 hands a "
                         "maintainer -- net_factory() + torch.softmax + PartialCrossEntropyLoss + ModelLossSemsegGatedCRF + loss.backward() + "
                         "torch.optim.SGD, the loop of train_weakly_supervised_pCE_GatedCRFLoss_2D.py:108-130 -- same batch and size; prints its own "
                         "JSON line.  A default N=1 run nests that line as \"modules_path\" (VERDICT r5 item 5)")
    ap.add_argument("--no-modules-record", action="store_true", help="do not run / nest the --path modules record in the default line")
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of --steps steps each (same engine, barrier + synchronize around each); "
                    "`value` / `ms_per_step` are the MEDIAN region's, min / max / all travel in `repeats` (VERDICT r5 item 4b)")
    ap.add_argument("--selftest-emulator", action="store_true",
                    help="TEST HARNESS ONLY (tests/test_dp.py), never a measurement: the launch / rendezvous / timing / reporting logic of this "
                         "script on the CPU -- ranks over gloo, the kernel sources in the test-only host emulator (tests/emul), value = null")
    return ap.parse_args()


def cpu_baseline(args):
    """The oracle (torch-CPU restatement of the reference step) on bounded samples of the same workload: batch 4 and batch
    16, ~10-15 s of CPU work each.  Built for BASELINE.json's configs 0-4: unet pce (config 0),
    unet_cct pce / pce_gatedcrf / ours_proposed, unet mean_teacher."""
    from oracle import torch_ref as R
    from wsl4mis_amd.synthetic import scribble_labels
    ncpu = os.cpu_count() or 1
    S = args.size
    if args.loss not in ("pce", "pce_gatedcrf", "ours_proposed", "mean_teacher") or (args.loss == "ours_proposed" and args.net != "unet_cct"):
        raise SystemExit(f"cpu baseline is not built for {args.net} {args.loss}")
    crf = args.crf_radius if args.loss == "pce_gatedcrf" else None

    def leg(B, budget_s):
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
        x = torch.rand((B, 1, S, S), generator=g)
        lab = torch.from_numpy(scribble_labels(B, S, S, 5))
        em = [(torch.rand((B, 16 << l, S >> l, S >> l), generator=g) >= R.DROP[l]).to(torch.uint8) for l in range(5)]
        cm = [(torch.rand((B, 16 << l), generator=g) >= 0.5).float() * 2 for l in range(5)]
        if args.loss == "mean_teacher":
            tr = R.RefMeanTeacher(sd)
            noise = torch.clamp(torch.randn((B, 1, S, S), generator=g) * 0.1, -0.2, 0.2)
            one = lambda: tr.step(x, lab, em, em, noise)                                      # noqa: E731
        else:
            tr = R.RefTrainer(sd, args.net)
            one = lambda: tr.step(x, lab, 0.4, em, cm, crf, kind="pce" if args.loss == "pce" else None)   # noqa: E731
        one()                                                  # warm-up (thread pool, oneDNN primitives)
        iters, t0 = 0, time.perf_counter()
        while iters < 2 or (time.perf_counter() - t0 < budget_s and iters < 12):
            one()
            iters += 1
        dt = time.perf_counter() - t0
        return {"batch": B, "value": round(B * iters / dt, 3), "timed_steps": iters, "seconds": round(dt, 2)}

    # the host's best.  Thread sweep up to every core (torch's intra-op pool; oneDNN convolutions stop scaling long before
    # 256 threads on these shapes, which is why the sweep exists) at the small batch, starting from the count that was best in earlier
    # rounds; then batch 16 and -- if the budget allows -- the GPU line's own batch 64 at the best thread count.  Bounded: a leg is
    # started only while the whole CPU part is under ~70 s, and every finished leg is printed at once (the parent keeps what arrived
    # if the time-out cuts the rest).
    if args.cpu_threads:
        sweep = [args.cpu_threads]
    else:
        sweep = [t for t in (32, 64, 128, ncpu, 16, 8, 4) if t <= ncpu] or [ncpu]
        sweep = list(dict.fromkeys(sweep))
    legs = []
    t_start = time.perf_counter()

    def run_leg(B, threads, budget):
        torch.set_num_threads(threads)
        r = leg(B, budget)
        r["threads"] = threads
        legs.append(r)
        print("LEG " + json.dumps(r), flush=True)

    # (measured on the GPU box's 256-core host, round 5: 32 threads 14.4 slices/s, 64: 6.5, 128: 2.0, 256: minutes per step -- the sweep
    #  walks up from 32 only while it still gains, then down -- 16, 8, 4 -- on the same rule)
    for t in sweep:
        if legs and time.perf_counter() - t_start > 50.0:
            break
        best_so_far = max((r["value"] for r in legs), default=0.0)
        if legs and t > legs[-1]["threads"] and legs[-1]["value"] < 0.95 * best_so_far:
            continue                                            # more threads already lost: do not go further up
        if legs and t < 32 and t < legs[-1]["threads"] and legs[-1]["threads"] < 32 and legs[-1]["value"] < 0.95 * best_so_far:
            continue                                            # ... nor further down once fewer threads lost
        run_leg(args.cpu_batch, t, 2.5)
    n_thr = max(legs, key=lambda r: r["value"])["threads"]
    if not args.cpu_one_batch:
        rate = max(r["value"] for r in legs)
        for B in (16, 64):     # larger batches at the best thread count, while three steps of them fit what is left of ~70 s
            if B != args.cpu_batch and (time.perf_counter() - t_start) + 3.0 * B / rate < 70.0:
                run_leg(B, n_thr, 3.0)
                if legs[-1]["value"] < 0.9 * rate:
                    break          # a larger batch already lost (the host is cache-bound): batch 64 would only cost half a minute
    return cpu_result(args, legs, sweep, ncpu, crf)


def cpu_result(args, legs, sweep, ncpu, crf):
    S = args.size
    best = max(legs, key=lambda r: r["value"])
    return {"value": best["value"], "unit": "slices/s", "cores": best["threads"], "kind": "port",
            "sample": f"oracle/torch_ref.py (stock torch CPU ops = what the reference's CPU path executes), {args.net} {args.loss}"
                      + (f" r={args.crf_radius}" if crf else "") + f", batch {best['batch']} at {S}x{S}, 1 warm-up + "
                      f"{best['timed_steps']} timed steps ({best['seconds']} s), {best['threads']} threads of {ncpu} host cores: the best of a "
                      f"thread sweep {sweep} at batch {args.cpu_batch} and of the larger batches in `legs` at the best thread count"
                      + ("; GatedCRF in the oracle is a tap loop over shifted views, which is KINDER to the CPU than the reference's "
                         "two F.unfold materialisations (127 MB per slice each)" if crf else ""),
            "host_cores": ncpu, "legs": legs}


def cpu_baseline_subprocess(args):
    """Run the CPU leg in its own interpreter with a hard time limit, so a slow host can never cost the GPU result."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--loss", args.loss, "--net", args.net,
           "--size", str(args.size), "--crf-radius", str(args.crf_radius), "--cpu-batch", str(args.cpu_batch),
           "--cpu-threads", str(args.cpu_threads)] + (["--cpu-one-batch"] if args.cpu_one_batch else [])
    def partial(out, why):
        legs = [json.loads(ln[4:]) for ln in (out or "").splitlines() if ln.startswith("LEG ")]
        if not legs:
            return {"value": None, "unit": "slices/s", "cores": 0, "kind": "port", "sample": why}
        crf = args.crf_radius if args.loss == "pce_gatedcrf" else None
        res = cpu_result(args, legs, sorted({r["threads"] for r in legs}), os.cpu_count() or 1, crf)
        res["sample"] += f" [{why}: the legs that had finished]"
        return res
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=200, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return partial(r.stdout, "failed: " + r.stderr[-300:])
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else e.stdout
        return partial(out, "timed out after 200 s")


def median_region(times):
    """(median seconds, index of the median region) of the repeated timed regions (odd counts: the middle one; even: the upper middle)"""
    order = sorted(range(len(times)), key=lambda i: times[i])
    i = order[len(order) // 2]
    return times[i], i


def modules_path(args):
    """The drop-in MODULE path: what a maintainer gets by swapping the reference trainer's three imports and
    keeping its loop -- ref: train_weakly_supervised_pCE_GatedCRFLoss_2D.py:108-130 (`unet`), and for `unet_cct` the dual-branch form of
    SURVEY 8d config 2 (0.5 (ce1 + ce2) + 0.1 GatedCRF(beta s1 + (1 - beta) s2)), the composition the fused engine runs.  torch.softmax,
    the mix, the loss sum, zero_grad, the p.grad copies of the autograd node and torch.optim.SGD are ATen kernels here; the networks, the
    partial cross-entropy and the GatedCRF module are this library's.  Same batch, size, lr schedule, timing protocol as the engine line."""
    from wsl4mis_amd.networks.net_factory import net_factory
    from wsl4mis_amd.synthetic import batch
    from wsl4mis_amd.utils import losses
    from wsl4mis_amd.utils.gate_crf_loss import ModelLossSemsegGatedCRF
    if args.loss != "pce_gatedcrf":
        raise SystemExit("--path modules is built for the headline composition (pce_gatedcrf)")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    torch.manual_seed(2022)
    random.seed(2022)
    model = net_factory(args.net, 1, 4, conv_precision=args.conv_precision).train()
    base_lr, max_it = 0.01, 60000
    opt = torch.optim.SGD(model.parameters(), lr=base_lr, momentum=0.9, weight_decay=0.0001)
    ce = losses.PartialCrossEntropyLoss(ignore_index=4)
    crf = ModelLossSemsegGatedCRF()
    desc = [{"weight": 1, "xy": 6, "rgb": 0.1}]
    x, lab = batch(args.batch, args.size, args.size, 2022, dev)
    lab = lab.long()                                      # (the reference passes label_batch[:].long())
    dual = args.net == "unet_cct"
    it = [0]
    last = {}

    def step():
        beta = random.random() + 1e-10
        out = model(x)
        if dual:
            o1, o2 = out
            s1, s2 = torch.softmax(o1, dim=1), torch.softmax(o2, dim=1)
            loss_ce = 0.5 * (ce(o1, lab) + ce(o2, lab))
            y = beta * s1 + (1.0 - beta) * s2
        else:
            y = torch.softmax(out, dim=1)
            loss_ce = ce(out, lab)
        g = crf(y, desc, args.crf_radius, x, args.size, args.size)["loss"]
        loss = loss_ce + 0.1 * g
        opt.zero_grad()
        loss.backward()
        opt.step()
        it[0] += 1
        lr_ = base_lr * (1.0 - it[0] / max_it) ** 0.9
        for pg in opt.param_groups:
            pg["lr"] = lr_
        last["loss"], last["ce"], last["crf"] = loss, loss_ce, g

    for _ in range(args.warmup):
        step()
    times = []
    for _ in range(max(1, args.repeats)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt, _ = median_region(times)
    rates = [args.batch * args.steps / t for t in times]
    # device kernels of ONE step by name: the library's (wsl::) against ATen's / the runtime's
    aten = wslk = None
    top = []
    try:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        names = [e.name for e in prof.events() if getattr(e, "device_type", None) is not None and "CUDA" in str(e.device_type).upper()]
        wslk = sum(1 for n in names if "wsl::" in n)
        aten = len(names) - wslk
        cnt = {}
        for e in prof.events():
            if getattr(e, "device_type", None) is not None and "CUDA" in str(e.device_type).upper() and "wsl::" not in e.name:
                c = cnt.setdefault(e.name[:80], [0, 0.0])
                c[0] += 1
                c[1] += float(getattr(e, "device_time", 0.0) or getattr(e, "cuda_time", 0.0) or 0.0)
        top = [{"kernel": k, "launches": v[0], "us": round(v[1], 1)} for k, v in sorted(cnt.items(), key=lambda kv: -kv[1][1])[:8]]
    except Exception as e:                                 # noqa: BLE001  (the count is a diagnosis, never the measurement)
        top = [{"profiler_failed": repr(e)[:200]}]
    return {"path": "modules", "value": round(args.batch * args.steps / dt, 2), "unit": "slices/s", "ms_per_step": round(1e3 * dt / args.steps, 3),
            "steps": args.steps, "warmup": args.warmup,
            "repeats": {"n": len(times), "values": [round(r, 2) for r in rates], "min": round(min(rates), 2), "max": round(max(rates), 2)},
            "what": f"{args.net} via net_factory() (one autograd node per network forward), torch.softmax, PartialCrossEntropyLoss, "
                    "ModelLossSemsegGatedCRF, loss.backward(), torch.optim.SGD(momentum 0.9, wd 1e-4) + poly lr: the reference trainer's loop "
                    "(train_weakly_supervised_pCE_GatedCRFLoss_2D.py:108-130) on the drop-in modules, batch "
                    f"{args.batch} at {args.size}x{args.size}, conv precision {args.conv_precision}",
            "aten_kernels_per_step": aten, "wsl_kernels_per_step": wslk, "largest_non_library_kernels_of_one_step": top,
            "last_losses": {k: round(float(v), 5) for k, v in last.items()}}


def modules_path_subprocess(args):
    """--path modules in its own interpreter with a hard time limit (the kernel count uses torch.profiler: it must never cost the line)"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--path", "modules", "--loss", args.loss, "--net", args.net, "--size", str(args.size),
           "--batch", str(args.batch), "--crf-radius", str(args.crf_radius), "--steps", str(max(10, min(args.steps, 20))),
           "--warmup", str(max(3, min(args.warmup, 5))), "--repeats", "3", "--conv-precision", args.conv_precision]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"path": "modules", "value": None, "failed": (r.stderr or r.stdout)[-400:]}
    except subprocess.TimeoutExpired:
        return {"path": "modules", "value": None, "failed": "timed out after 240 s"}


def main():
    args = parse()
    if args.loss in ("mean_teacher", "ustm", "pce_tv", "pce_ms", "pce_entropy", "ce_dice"):
        args.net = "unet"                                  # config 4 (and USTM): single-decoder student + EMA teacher
        if args.loss != "mean_teacher":
            args.no_cpu_baseline = True
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args)), flush=True)
        return
    if args.path == "modules":
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (there is no CPU path in the product)")
        from wsl4mis_amd import _lib as _l
        if _l.library_sha256() != _l.source_sha256():
            raise SystemExit("libwslhip.so was not built from this tree: run wsl4mis_amd/csrc/build.sh")
        print(json.dumps(modules_path(args)), flush=True)
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, 127.0.0.1 rendezvous on a
        # free port) -- the command the driver documents for N > 1, with this script's own arguments.  A rank that fails takes the
        # others down (torch.distributed.run kills the group) and the exit status is passed on.
        import socket
        import subprocess
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree")
    emu = args.selftest_emulator
    from wsl4mis_amd import _lib
    if emu:
        # test harness: same control flow, CPU tensors, gloo, the test-only emulator library; nothing below measures anything
        args.no_prof = args.no_split_record = args.no_cpu_baseline = True
        torch.set_num_threads(1)
        _lib.use_library_for_tests(C.CDLL(os.path.join(ROOT, "tests", "emul", "libwslhip_emul.so")))
        torch.cuda.synchronize = lambda *a, **k: None
        torch.cuda.empty_cache = lambda: None
    elif not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU path in the product)")
    else:
        torch.cuda.set_device(local)
    backend = "gloo" if emu else "nccl"
    pg_kw = {} if emu else {"device_id": torch.device("cuda", local)}
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, **pg_kw)
    elif args.force_dp:
        import socket
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
        dist.init_process_group(backend, init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, **pg_kw)
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    from wsl4mis_amd.engine import TrainEngine
    from wsl4mis_amd.synthetic import batch
    dev = torch.device("cpu") if emu else torch.device("cuda", local)
    L = _lib.lib()
    # the binary that is measured names the sources it was built from (csrc/build.sh compiles their SHA-256 into wsl_build_info()); a
    # library that was not built from THIS tree is refused -- A/B builds given with --lib are other trees' binaries by definition and only
    # reported
    lib_sha, tree_sha = (None, None) if emu else (_lib.library_sha256(L), _lib.source_sha256())
    if not emu and not args.lib and lib_sha != tree_sha:
        raise SystemExit(f"libwslhip.so carries source hash {lib_sha} but this tree hashes to {tree_sha}: run wsl4mis_amd/csrc/build.sh")
    x, lab = batch(args.batch, args.size, args.size, 2022 + rank, dev)
    overlapped = (args.net == "unet_cct" or args.loss == "mean_teacher") and not args.serial_decoders
    # per-launch HIP events cost ~2 % of the step rate: when the roofline comes from its own serialised segment anyway
    # (overlapped run) the timed region stays uninstrumented unless --prof-timed asks for its overlapping figures too
    prof_timed = not args.no_prof and (not overlapped or args.prof_timed)

    def timed_region(precision, steps, warmup, repeats=1):
        """engine of this conv precision, `warmup` untimed steps, then `repeats` regions of exactly `steps` steps between barrier +
        synchronize on both sides; returns (engine, max-over-ranks seconds of the median region, this rank's own seconds of that region,
        every region's max-over-ranks seconds)"""
        torch.manual_seed(2022)                                   # same initial weights on every rank
        eng = TrainEngine(args.net, 1, 4, base_lr=0.01, max_iterations=60000, loss=args.loss, crf_radius=args.crf_radius,
                          force_dp=args.force_dp, conv_precision=precision)
        torch.manual_seed(2022 + 1000 * rank)                     # different dropout masks / data per rank
        random.seed(2022)                                         # identical beta stream on all ranks
        if args.serial_decoders:
            L.wsl_net_concurrent(0)
            eng.concurrent = False
        for _ in range(warmup):
            eng.step(x, lab, random.random() + 1e-10)
        if world > 1 or args.force_dp:
            torch.cuda.synchronize()
            C.CDLL(None).fflush(None)     # RCCL prints its version banner through C stdio: out now, not after the JSON line
        if prof_timed:
            L.wsl_prof_enable(1)
        if eng.dp:
            eng.comm_diag(True)                                   # two events per step around the wait for the comm stream
        # `repeats` timed regions of exactly `steps` steps each, same engine, each bracketed by barrier + synchronize on both sides and
        # reduced with MAX over the ranks; the caller reports the MEDIAN region (one 0.3 s sample cannot resolve the
        # per-cent steps the kernels are tuned in, and boxes differ by 1-3 %)
        regions = []
        for rep in range(repeats):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i_ in range(steps):
                if emu and rep == 0 and os.environ.get("WSL_SELFTEST_FAIL") == f"{rank}:{i_}":      # (test harness: tests/test_dp.py's rank-failure case)
                    raise RuntimeError(f"injected failure of rank {rank} in timed step {i_}")
                eng.step(x, lab, random.random() + 1e-10)
            torch.cuda.synchronize()
            dt_own = time.perf_counter() - t0                         # this rank's own time, before it waits for the others
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            regions.append((float(tt.item()), dt_own))
        dt, imed = median_region([r[0] for r in regions])
        return eng, dt, regions[imed][1], [r[0] for r in regions]

    n_rep = 1 if emu else max(1, args.repeats)
    eng, dt, dt_own, region_s = timed_region(args.conv_precision, args.steps, args.warmup, n_rep)
    losses = eng.losses()
    dp_diag = None
    if eng.dp:
        # self-diagnosing N > 1 line: what each rank saw, how long the main stream sat waiting for the
        # all-reduce stream, and whether the replicas are still bit-identical after the timed region
        wait_ms = eng.comm_diag(False) / max(1, n_rep)          # (the events cover every timed region: per region)
        bucket_us = eng.bucket_diag()
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if not emu else None
        except Exception:                                       # noqa: BLE001
            rccl = None
        p = eng.model.flat_params()
        digest = torch.stack([p.view(torch.int32).to(torch.int64).sum(), (p.view(torch.int32).to(torch.int64) * torch.arange(
            1, p.numel() + 1, device=dev, dtype=torch.int64) % 1000003).sum()])
        mine = torch.tensor([dt_own * 1e3 / args.steps, wait_ms / max(1, args.steps), float(digest[0].item()), float(digest[1].item()),
                             float(losses["loss"])], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        if world > 1:
            dist.all_gather(allr, mine)
        else:
            allr = [mine]
        rows_ = [[float(v) for v in t.tolist()] for t in allr]
        dp_diag = {"backend": dist.get_backend(), "world_size_seen_by_the_process_group": dist.get_world_size(),
                   "per_rank_ms_per_step": [round(r[0], 3) for r in rows_],
                   "ms_per_step_min": round(min(r[0] for r in rows_), 3), "ms_per_step_max": round(max(r[0] for r in rows_), 3),
                   "per_rank_ms_per_step_main_stream_blocked_on_allreduce": [round(r[1], 4) for r in rows_],
                   "replicas_bit_identical_after_timed_region": all(r[2] == rows_[0][2] and r[3] == rows_[0][3] for r in rows_),
                   "per_rank_last_loss": [round(r[4], 5) for r in rows_],
                   "rccl_version": rccl, "hip_visible_devices": os.environ.get("HIP_VISIBLE_DEVICES"), "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                   "allreduce_us_per_bucket_on_the_comm_stream_rank0": bucket_us,     # {elements: mean us}, HIP events on the comm stream
                   "allreduce": "flat fp32 gradient arena in 2 buckets (decoders, then encoder) on a side stream; bytes per step "
                                f"{4 * eng.model.n_param}", "semantics": "DDP-equivalent (per-rank BN statistics and loss normalisation, gradients averaged)"}

    def report():
        rows = (_lib.WslProfRow * _lib.WSL_PROF_FAMILIES)()
        L.wsl_prof_report(rows, _lib.WSL_PROF_FAMILIES)
        L.wsl_prof_enable(0)
        fams = {}
        for r in rows:
            if r.calls:
                fams[r.name.decode()] = {"calls": int(r.calls), "ms": round(r.ms, 3),
                                         "avg_us": round(1e3 * r.ms / r.calls, 2),
                                         "tflops": round(r.flops / (r.ms * 1e-3) / 1e12, 2) if r.flops else None,
                                         "issued_tflops": round(r.issued_flops / (r.ms * 1e-3) / 1e12, 2) if r.flops else None,
                                         "algo_GBps": round(r.bytes / (r.ms * 1e-3) / 1e9, 1)}
        return rows, fams

    fresh_traffic = [None]

    def pmc_refresh():
        """two counter passes of this command in child processes (the GPU is still untouched by this process)"""
        import shutil
        import subprocess
        import tempfile
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pmc_traffic as agg
        tmp = tempfile.mkdtemp(prefix="wsl_pmc_", dir="/tmp")
        child = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-prof", "--no-split-record", "--no-pmc-refresh",
                 "--serial-decoders", "--conv-precision", args.conv_precision, "--loss", args.loss, "--net", args.net, "--batch", str(args.batch),
                 "--size", str(args.size), "--crf-radius", str(args.crf_radius)]
        try:
            for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
                subprocess.run(["rocprofv3", "--pmc", ctr, "--output-format", "csv", "-d", os.path.join(tmp, ctr), "--"] + child,
                               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=150, check=True)
            fresh_traffic[0] = agg.aggregate(os.path.join(tmp, "FETCH_SIZE"), os.path.join(tmp, "WRITE_SIZE"))
        except (OSError, subprocess.SubprocessError, AssertionError) as e:
            print(f"[bench] --pmc-refresh failed ({e}); falling back to the committed record", file=sys.stderr)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)

    def pmc_traffic(name, algo_bytes_per_launch):
        """HBM bytes per launch of the dominant kernel family from the newest committed rocprofv3 PMC record
        (FETCH_SIZE / WRITE_SIZE in separate runs of this same command -- tools/pmc_traffic.py); the record's file name,
        SHA-256 and modification date travel in the line, so a stale record is visible."""
        import glob
        import hashlib
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), key=os.path.basename)          # r1z < r2a < ...: the newest round's record
        if not cands and fresh_traffic[0] is None:
            return None
        tfile = cands[-1] if cands else "(none)"
        try:
            if fresh_traffic[0] is not None:
                tdoc = fresh_traffic[0]
                raw = json.dumps(tdoc, sort_keys=True).encode()
            else:
                raw = open(tfile, "rb").read()
                tdoc = json.loads(raw)
            tj = tdoc["kernels"]
            keys = ("conv_sp_kernel",) if name.startswith("conv_sp") else ("wgrad_sp_kernel",) if name.startswith("wgrad_sp") else \
                   ("conv_wino2_kernel", "conv_wino_kernel") if name.startswith("conv_wino") else \
                   ("conv_mfma2l_kernel", "conv_mfma2_kernel", "conv_nk16_kernel") if name.startswith("conv") else \
                   ("wgrad_wino_kernel",)
            nl = sum(tj[k]["launches_sampled"] for k in keys if k in tj)
            return {"hbm_bytes_per_launch": sum(tj[k]["hbm_bytes_per_launch"] * tj[k]["launches_sampled"] for k in keys if k in tj) / nl,
                    "algorithmic_bytes_per_launch": algo_bytes_per_launch,
                    "algorithmic_bytes_counted": "4 (Ci + Co) B per output pixel, plus -- on the data-gradient launches that carry the BatchNorm-backward "
                                                 "statistics epilogue -- the epilogue's reads of the consumer layer's y (4 B) and keep mask (1 B) per "
                                                 "output element (VERDICT r5 weak 6: those reads are part of the launch's algorithm)",
                    "source": ("collected in THIS run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this command at --steps 2, FETCH_SIZE x2 per "
                               "MI355X_MICROARCH.md)") if fresh_traffic[0] is not None else
                              (f"profiles/{os.path.basename(tfile)} (rocprofv3 --pmc, FETCH_SIZE x2 per MI355X_MICROARCH.md; a committed "
                               "record of an earlier run of this command, not collected in this run)"),
                    "source_sha256": hashlib.sha256(raw).hexdigest(),
                    "source_collected_utc": tdoc.get("collected_utc")}
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            return None

    def sq_counters(split):
        """SQ counters of the MFMA kernel families from the newest committed profiles/r*_pmc_sq_{f32,split}.md (rocprofv3 --pmc passes of
        this command, tools/record_round.sh): the matrix pipe's busy share is what `issued_frac` estimates from flops."""
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_sq_{'split' if split else 'f32'}.md")), key=os.path.basename)
        if not cands:
            return None
        out = {}
        try:
            for line in open(cands[-1]):
                c = [t.strip() for t in line.strip().strip("|").split("|")]
                if len(c) >= 5 and c[0].startswith("`") and c[1] not in ("–", "-"):
                    out[c[0].strip("`")] = {"mfma_pipe_busy": float(c[1]), "vector_instruction_active": None if c[2] == "–" else float(c[2]),
                                            "lds_bank_conflict_share_of_lds_active": None if c[3] == "–" else float(c[3]),
                                            "valu_instructions_per_mfma": None if c[4] == "–" else float(c[4])}
        except (OSError, ValueError):
            return None
        return {"source": f"profiles/{os.path.basename(cands[-1])} (a committed record; SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES etc. per kernel family)",
                "kernels": out}

    def roofline_of(rows, nsteps, ms_per_step):
        byname = {r.name.decode(): r for r in rows if r.calls}
        conv = [r for r in byname.values() if r.flops > 0]
        if not conv:
            return None
        # one entry per MFMA kernel family.  Flops are the ALGORITHMIC ones (direct convolution: 2 * 9 * Ci * Co per pixel,
        # SURVEY 8d) -- `frac`; the Winograd F(2x2,3x3) kernels issue 2.25x fewer matrix instructions for the same result, so
        # `issued_frac` (matrix-core flops actually issued / peak) is reported next to it: THAT is the matrix-pipe utilisation.
        kern = {"conv_mfma2l_kernel (direct: 1x1, first conv, classifiers; fwd + data-gradient launches)":
                    [byname[k] for k in ("conv_mfma2l_kernel(fwd)", "conv_mfma2l_kernel(dgrad)") if k in byname],
                "conv_wino2_kernel (Winograd; fwd + data-gradient launches)":
                    [byname[k] for k in ("conv_wino2_kernel(fwd)", "conv_wino2_kernel(dgrad)") if k in byname],
                "wgrad_wino_kernel (Winograd weight gradient)": [byname[k] for k in ("wgrad_wino_kernel",) if k in byname],
                "wgrad_direct_kernels (1x1, first conv, classifiers)": [byname[k] for k in ("wgrad_direct_kernels",) if k in byname],
                "conv_sp_kernel (split precision; fwd + data-gradient launches)":
                    [byname[k] for k in ("conv_sp_kernel(fwd)", "conv_sp_kernel(dgrad)") if k in byname],
                "wgrad_sp_kernel (split-precision weight gradient)": [byname[k] for k in ("wgrad_sp_kernel",) if k in byname]}

        def fam(v):
            ms, fl, iss, calls = sum(r.ms for r in v), sum(r.flops for r in v), sum(r.issued_flops for r in v), sum(r.calls for r in v)
            return {"ms_per_step": round(ms / nsteps, 3), "launches": int(calls), "avg_launch_us": round(1e3 * ms / calls, 2),
                    "achieved": round(fl / (ms * 1e-3) / 1e12, 2), "frac": round(fl / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                    "issued": round(iss / (ms * 1e-3) / 1e12, 2), "issued_frac": round(iss / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                    "algo_GBps": round(sum(r.bytes for r in v) / (ms * 1e-3) / 1e9, 1),
                    "algo_frac_of_hbm_peak": round(sum(r.bytes for r in v) / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
        per_kernel = {k: fam(v) for k, v in kern.items() if v}
        for k, v in per_kernel.items():      # the split kernels issue f16 MFMAs: their matrix roofline is the f16 peak
            if "_sp_kernel" in k:
                v["peak"] = round(PEAK_F16_MFMA_TFLOPS / 3.0, 1)      # algorithmic flops at three f16 passes per product
                v["frac"] = round(v["achieved"] / (PEAK_F16_MFMA_TFLOPS / 3.0), 4)
                v["issued_frac"] = round(v["issued"] / PEAK_F16_MFMA_TFLOPS, 4)
            else:
                v["peak"] = PEAK_F32_MFMA_TFLOPS
        name, grp = max(((k, v) for k, v in kern.items() if v), key=lambda kv: sum(r.ms for r in kv[1]))
        d = per_kernel[name]
        calls = sum(r.calls for r in grp)
        traffic = None      # attached by attach_traffic() once the counter passes have run (after every engine of this process is freed)
        # HBM-bound kernel families: SURVEY 8d's algorithmic bytes / measured time / 8 TB/s
        hbm = {}
        for k in ("bnact_bwd(reduce+finalize+apply)", "bilinear_up2(fwd+bwd)", "pool2_fwd+feat_grad_combine", "gatedcrf_fwd_kernel",
                  "loss_head(reduce+finalize+bwd+mix)", "sgd_kernel", "bn_finalize_kernel", "wgrad_reduce_kernel", "masks+filter_images"):
            if k in byname:
                r = byname[k]
                gbs = r.bytes / (r.ms * 1e-3) / 1e9
                hbm[k] = {"ms_per_step": round(r.ms / nsteps, 3), "launches_per_step": round(r.calls / nsteps, 1),
                          "algorithmic_bytes_per_step": round(r.bytes / nsteps), "achieved_GBps": round(gbs, 1),
                          "frac": round(gbs / PEAK_HBM_GBS, 4)}
        # (split-precision kernels: algorithmic flops against what three f16 passes can deliver, 2500 / 3 TFLOP/s, issued flops against
        #  the f16 peak; the HBM side of the same launches is in kernels[*].algo_frac_of_hbm_peak.)  whole_step_issued_frac: every MFMA
        #  kernel's issued flops over the peak of the instruction it issues = the matrix pipe's busy share of the timed step
        pipe_s = sum(r.issued_flops / ((PEAK_F16_MFMA_TFLOPS if "_sp_kernel" in r.name.decode() else PEAK_F32_MFMA_TFLOPS) * 1e12) for r in conv) / nsteps
        return {"bound": "mfma", "kernel": name, "achieved": d["achieved"], "peak": d["peak"],
                "unit": "TFLOP/s", "frac": d["frac"],
                "issued": d["issued"], "issued_frac": d["issued_frac"],
                "whole_step_issued_frac": round(pipe_s / (ms_per_step * 1e-3), 4),
                "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,     # HBM bytes per launch (PMC)
                "traffic_detail": traffic,
                "_traffic_key": (name, sum(r.bytes for r in grp) / calls),
                "launches": int(calls), "avg_launch_us": d["avg_launch_us"], "flops_per_launch": sum(r.flops for r in grp) / calls,
                "flops_counted": "frac/achieved: algorithmic (direct convolution); issued/issued_frac: matrix-core flops issued "
                                 "(= algorithmic / 2.25 for the Winograd kernels, x 3 for the split-precision kernels, against the f32 resp. f16 MFMA "
                                 "peak); whole_step_issued_frac: all MFMA kernels' issued flops / the peak of the instruction each issues, per "
                                 "step / the timed region's ms_per_step",
                "kernels": per_kernel,
                "sq_counters": sq_counters(any("_sp_kernel" in r.name.decode() for r in conv)),
                "all_mfma_kernels_tflops": round(sum(r.flops for r in conv) / (sum(r.ms for r in conv) * 1e-3) / 1e12, 2),
                "all_mfma_kernels_ms_per_step": round(sum(r.ms for r in conv) / nsteps, 3),
                "hbm_roofline": {"peak_GBps": PEAK_HBM_GBS, "kernels": hbm,
                                 "all_hbm_kernels_ms_per_step": round(sum(v["ms_per_step"] for v in hbm.values()), 3)}}

    def attach_traffic(roof):
        if roof and "_traffic_key" in roof:
            name, algo = roof.pop("_traffic_key")
            t = pmc_traffic(name, algo)
            roof["traffic"] = t["hbm_bytes_per_launch"] if t else None
            roof["traffic_detail"] = t

    def roofline_segment(eng, dt, steps):
        """(roofline object, per-family rows) of the engine that just ran a timed region of `steps` steps in `dt` seconds"""
        roof, fams = None, {}
        if args.no_prof:
            return roof, fams
        timed = None
        if prof_timed:
            rows, fams = report()
            roof = roofline_of(rows, steps, 1e3 * dt / steps)
        if overlapped:
            # second segment, decoders serialised: launches of the dominant kernel no longer overlap each other
            if roof:
                timed = {k: roof[k] for k in ("achieved", "frac", "launches", "avg_launch_us", "all_mfma_kernels_tflops")}
            seg = max(1, min(steps, 5))
            L.wsl_net_concurrent(0)
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
            L.wsl_net_concurrent(1)
            eng.concurrent = True
            roof = roofline_of(rows2, seg, 1e3 * dt / steps)
            if roof:
                roof["measured"] = (f"{seg} extra steps of the same workload right after the timed region, two decoder streams "
                                    f"serialised ({round(seg_ms, 3)} ms/step incl. event overhead); in the timed region "
                                    "launches of this kernel overlap each other, so a per-launch duration does not measure it")
                if timed:
                    roof["timed_region_overlapped"] = timed
        return roof, fams

    roof, fams = roofline_segment(eng, dt, args.steps)
    # the SAME workload on the split-precision conv path, as a nested record: its own engine, warm-up, timed region
    # (barrier + synchronize on both sides, max over ranks) and serialised roofline segment.  Headline fields stay the f32 path's.
    split_rec = None
    if args.conv_precision == "f32" and not args.no_split_record and args.loss != "ustm":
        del eng
        torch.cuda.empty_cache()
        s_steps, s_warm = max(20, min(args.steps, 50)), max(5, min(args.warmup, 10))
        eng_s, dt_s, _, region_s_split = timed_region("split_f16x3", s_steps, s_warm, n_rep)
        if eng_s.dp:
            eng_s.comm_diag(False)
        losses_s = eng_s.losses()
        roof_s, _ = roofline_segment(eng_s, dt_s, s_steps)
        split_rec = {"value": round(args.batch * world * s_steps / dt_s, 2), "unit": "slices/s", "ms_per_step": round(1e3 * dt_s / s_steps, 3),
                     "steps": s_steps, "warmup": s_warm, "dtype": "f32-split-f16x3",
                     "repeats": {"n": len(region_s_split), "values": [round(args.batch * world * s_steps / t, 2) for t in region_s_split]},
                     "what": "the same workload with the 3x3 convolutions (forward, data gradient, weight gradient of the layers with >= 32 "
                             "channels) on v_mfma_f32_16x16x32_f16: every operand split into f16 hi + lo while a tile is staged, three passes, "
                             "fp32 accumulation; fp32 storage everywhere.  Same parity tests and tolerances as the f32 path "
                             "(tests/test_ops_convsp.py, test_net.py, test_error_budget.py, test_fullsize.py, test_concurrency.py)",
                     "roofline": roof_s, "last_losses": {k: round(v, 5) for k, v in losses_s.items()}}
        del eng_s
    else:
        del eng
    import shutil as _sh
    # roofline.traffic, collected in this run (the full default record only: the tuning scripts' `--no-cpu-baseline` lines, A/B builds and
    # runs under a profiler skip it; N = 1 only -- ADVICE r5: every rank of a larger job would spawn its own children).  AFTER every engine of
    # this process has been freed: the two counter passes are child processes on the same GPU.
    if world == 1 and (args.pmc_refresh or (not args.no_cpu_baseline and not args.lib)) and not args.no_prof and not emu \
            and not args.no_pmc_refresh and _sh.which("rocprofv3"):
        torch.cuda.empty_cache()
        pmc_refresh()                  # (about 40 s, bounded by time-outs)
    attach_traffic(roof)
    if split_rec:
        attach_traffic(split_rec["roofline"])
    if rank == 0:
        gflop = 28.98 if args.net == "unet_cct" else 17.68     # conv-stack training GFLOP/slice
        if args.loss == "mean_teacher":
            gflop += 5.899                                     # + the teacher's forward
        if args.loss == "ustm":
            gflop += 9 * 5.899                                 # + 1 + 4 x 2 teacher forwards per student slice
        value = args.batch * world * args.steps / dt
        split = args.conv_precision != "f32"
        headline = args.net == "unet_cct" and args.loss == "pce_gatedcrf" and args.size == 256 and args.batch == 64
        wl = {"pce_gatedcrf": "pCE+GatedCRF", "ours_proposed": "pCE+mixed pseudo-label Dice", "pce": "pCE", "mean_teacher": "mean teacher pCE+TV+consistency",
              "ustm": "USTM", "pce_tv": "pCE+TV", "pce_ms": "pCE+Mumford-Shah", "pce_entropy": "pCE+entropy", "ce_dice": "CE+Dice"}[args.loss]
        out = {"metric": ("training slices/sec (256x256, bs64, unet_cct pCE+GatedCRF)" if headline else
                          f"training slices/sec ({args.size}x{args.size}, bs{args.batch}, {args.net} {wl}) -- not BASELINE.json's headline workload")
                         + (" -- SPLIT-PRECISION RECORD" if split else ""),
               "value": None if emu else round(value, 2),
               "unit": "slices/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32-split-f16x3" if split else "f32",
               "data": "SELFTEST on the host emulator: control flow only, NOT a measurement" if emu else "synthetic",
               "config": {"workload": f"{args.net} {args.loss}" + (f" r={args.crf_radius}" if args.loss == "pce_gatedcrf" else "")
                          + f", {args.size}x{args.size}x1 4-class synthetic scribble slices, batch {args.batch}/GPU, SGD+poly LR",
                          "global_batch": args.batch * world, "parallelism": f"dp{world}", "crf_radius": args.crf_radius,
                          "conv_precision": args.conv_precision, **({"library": args.lib} if args.lib else {}),
                          "library_sha": lib_sha, "tree_sha": tree_sha},
               "repeats": {"n": len(region_s), "what": f"{len(region_s)} timed regions of {args.steps} steps each on one engine (barrier + synchronize "
                           "around each, max over ranks); value / ms_per_step are the median region's",
                           "values": [round(args.batch * world * args.steps / t, 2) for t in region_s],
                           "ms_per_step": [round(1e3 * t / args.steps, 3) for t in region_s],
                           "min": round(args.batch * world * args.steps / max(region_s), 2),
                           "max": round(args.batch * world * args.steps / min(region_s), 2)},
               "whole_step_conv_mfma_frac": round(value / world * gflop * 1e9 / (PEAK_F32_MFMA_TFLOPS * 1e12), 4),
               "roofline": roof, "kernels": fams, "last_losses": {k: round(v, 5) for k, v in losses.items()}}
        if split_rec:
            out["split_f16x3"] = split_rec
        if dp_diag:
            out["dp"] = dp_diag
        if world == 1 and not args.no_cpu_baseline and not args.no_modules_record and not args.lib and not emu and args.loss == "pce_gatedcrf":
            mp = modules_path_subprocess(args)               # the drop-in module path's own rate, next to the fused engine's
            if mp.get("value"):
                mp["ratio_to_engine"] = round(mp["value"] / value, 4)
            out["modules_path"] = mp
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_subprocess(args)
        if world > 1 or args.force_dp:
            C.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if world > 1 or args.force_dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
