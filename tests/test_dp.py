"""Data-parallel path on the CPU: two `gloo` ranks, each running the engine's split backward + bucketed all-reduce on
its shard (kernel sources in the host emulator), against the DDP-emulation fixture generated from the reference
(per-shard BatchNorm statistics and loss normalisation, gradients averaged -- SURVEY 8e)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, get_backend, golden, grad_tol


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_matches_ddp_fixture(tmp_path):
    get_backend("emul")                         # builds the emulation library if needed
    port = str(free_port())
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(r), "2", port, str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=1500)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    g = golden("g8_ddp")
    r0, r1 = (np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(2))
    assert abs(float(r0["loss"]) - float(g["r0_loss"])) < 1e-4 * abs(float(g["r0_loss"]))
    assert abs(float(r1["loss"]) - float(g["r1_loss"])) < 1e-4 * abs(float(g["r1_loss"]))
    bad = []
    for k in g.files:
        if k.startswith("g."):
            assert np.array_equal(r0[k], r1[k]), k                  # both ranks hold the same averaged gradient
            err = float(np.max(np.abs(r0[k] - g[k])))
            if err > grad_tol(k[2:], g[k]):
                bad.append((k, err, float(np.max(np.abs(g[k])))))
    assert not bad, bad[:6]
    assert np.array_equal(r0["params_after"], r1["params_after"])   # replicas stay bit-identical after the step


@pytest.mark.gpu
@pytest.mark.timeout(1200)
@pytest.mark.parametrize("kind", ["pce_gatedcrf", "mean_teacher"])
def test_two_rank_gloo_on_one_gpu_match_the_oracle(tmp_path, kind):
    """the data-parallel route with world_size 2 on the REAL library -- two processes sharing cuda:0, gloo
    carrying the broadcasts and the bucketed gradient all-reduce (RCCL cannot put two ranks on one device; the driver's 8-GPU
    run covers that transport) -- against the oracle's per-shard computation averaged, replicas bit-identical."""
    _two_rank_against_the_oracle(tmp_path, kind, gpu=True)


@pytest.mark.parametrize("kind", ["pce_gatedcrf", "mean_teacher", "pce_gatedcrf_split"])
def test_two_rank_gloo_other_compositions_match_the_oracle(tmp_path, kind):
    """(`_split`: the headline composition with the eligible layers on the split-precision conv path -- same oracle, same criteria)"""
    _two_rank_against_the_oracle(tmp_path, kind, gpu=False)


def _two_rank_against_the_oracle(tmp_path, kind, gpu):
    """The headline composition (unet_cct pCE + GatedCRF) and BASELINE.json config 4 (mean teacher: teacher forward + gradient
    all-reduce in one step) through the data-parallel route on two gloo ranks: the averaged gradient, the SGD step and the
    EMA teacher equal the oracle's per-shard computation averaged (DDP-equivalent semantics), and the replicas
    stay bit-identical."""
    import torch
    from detinit import det_state
    from oracle import torch_ref as R
    import dp_worker
    worker_kind, kind = kind, (kind[:-6] if kind.endswith("_split") else kind)
    if not gpu:
        get_backend("emul")
    port = str(free_port())
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(r), "2", port, str(tmp_path), worker_kind] +
                              (["gpu"] if gpu else []), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=1500)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    r0, r1 = (np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(2))
    assert np.array_equal(r0["grads"], r1["grads"]) and np.array_equal(r0["params_after"], r1["params_after"])
    # ---- oracle: each shard on its own (own BN statistics, own loss normalisation), gradients averaged
    net = "unet_cct" if kind == "pce_gatedcrf" else "unet"
    layout = {k: tuple(shp) for k, shp in R.state_layout(net, 1, 4)}

    _det = {}

    def det(seed):                                  # (a fresh copy per call; the deterministic state itself is generated once per seed)
        if seed not in _det:
            _det[seed] = {k: torch.from_numpy(np.asarray(v)) for k, v in det_state(layout, seed).items()}
        return {k: v.clone() for k, v in _det[seed].items()}
    pk = [k for k in layout if R.is_param(k)]
    from netutil import KinkMargins
    shard_grads, losses, it = [], [], 4500
    for r in range(2):
        d = dp_worker.shard_inputs(r, kind)
        with KinkMargins() as km, torch.no_grad():         # the shard is clear of gradient discontinuities (see dp_worker.SHARD_SEEDS)
            R.net_forward(det(9), d["x"], net, d["em"], d["cm"] if net == "unet_cct" else None, True)
            if kind == "mean_teacher":
                R.net_forward(det(22), d["x"] + d["noise"], net, d["em_t"], None, True)
        assert km.leaky > 4e-6 and km.pool > 1e-6, (r, km.leaky, km.pool)
        sd = det(9)
        for k in pk:
            sd[k].requires_grad_(True)
        if kind == "pce_gatedcrf":
            o1, o2 = R.net_forward(sd, d["x"], net, d["em"], d["cm"], True)
            s1, s2 = torch.softmax(o1, 1), torch.softmax(o2, 1)
            lce = 0.5 * (R.ce_ignore(o1, d["lab"]) + R.ce_ignore(o2, d["lab"]))
            loss = lce + 0.1 * R.gatedcrf(d["beta"] * s1 + (1.0 - d["beta"]) * s2, d["x"], 2)[0]
        else:
            with torch.no_grad():
                zt = R.net_forward(det(22), d["x"] + d["noise"], net, d["em_t"], None, True)
            loss = R.mean_teacher_loss(R.net_forward(sd, d["x"], net, d["em"], None, True), zt, d["lab"], it)[0]
        loss.backward()
        losses.append(float(loss.detach()))
        shard_grads.append(np.concatenate([sd[k].grad.numpy().ravel() for k in pk]).astype(np.float64))
    for r, rr in enumerate((r0, r1)):
        assert abs(float(rr["loss"]) - losses[r]) <= 1e-4 * abs(losses[r]), (r, float(rr["loss"]), losses[r])
    ref = 0.5 * (shard_grads[0] + shard_grads[1])
    off, bad = 0, []
    for k in pk:
        n = int(np.prod(layout[k])) if layout[k] else 1
        err = float(np.max(np.abs(r0["grads"][off:off + n] - ref[off:off + n])))
        if err > grad_tol(k, ref[off:off + n]):
            bad.append((k, err, float(np.max(np.abs(ref[off:off + n])))))
        off += n
    assert not bad, bad[:6]
    # SGD with the averaged gradient (it = 4500: momentum buffer starts at zero -> buf = g), then the EMA teacher
    d9 = det(9)
    p0 = np.concatenate([d9[k].numpy().ravel() for k in pk]).astype(np.float64)
    lr = 0.01
    p1 = p0 - lr * (ref + 1e-4 * p0)
    assert np.max(np.abs(r0["params_after"] - p1)) <= 1e-6 * np.max(np.abs(p1))
    if kind == "mean_teacher":
        assert np.array_equal(r0["teacher_after"], r1["teacher_after"])
        d22 = det(22)
        t0 = np.concatenate([d22[k].numpy().ravel() for k in pk]).astype(np.float64)
        a = min(1.0 - 1.0 / (it + 1), 0.99)
        assert np.max(np.abs(r0["teacher_after"] - (a * t0 + (1 - a) * p1))) <= 1e-6 * np.max(np.abs(t0))


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_single_rank_rccl_group_takes_the_dp_route_and_changes_nothing():
    """The N>1 route of bench.py on the real box: RCCL ("nccl") process group, split backward, both gradient buckets
    all-reduced on the side stream -- in a 1-rank group the result must be bit-identical to the plain step."""
    code = r"""
import os, sys, random, torch, torch.distributed as dist
sys.path.insert(0, %r)
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from wsl4mis_amd.engine import TrainEngine
from wsl4mis_amd.synthetic import batch
for prec in ("f32", "split_f16x3"):       # both conv precisions take the same route (VERDICT r3 item 8)
    outs = []
    for force in (False, True):
        torch.manual_seed(5)
        eng = TrainEngine("unet_cct", 1, 4, loss="pce_gatedcrf", force_dp=force, conv_precision=prec)
        assert eng.dp == force and (eng.comm is not None) == force
        x, lab = batch(4, 64, 64, 11, torch.device("cuda", 0))
        random.seed(3)
        for _ in range(3):
            eng.step(x, lab, random.random() + 1e-10)
        outs.append((eng.model.flat_params().clone(), eng.losses()))
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1], (prec, outs[0][1], outs[1][1])
dist.barrier()
t = torch.ones(1, device="cuda") * 3
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t) == 3.0
dist.destroy_process_group()
print("DP_ROUTE_OK")
""" % (ROOT, free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "DP_ROUTE_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


# ---------------------------------------------------------------------------------------------- bench.py --gpus N launches itself
def _bench_selftest(gpus, extra=(), env_extra=None, timeout=900, steps=2, warmup=0):
    """`python bench.py --gpus N` WITHOUT a launcher (no WORLD_SIZE in the environment): the script starts its own ranks.  Test
    harness mode: gloo ranks on the CPU, kernels in the host emulator, 16 x 16 slices -- control flow only, the line's value is null."""
    import json
    get_backend("emul")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1", **(env_extra or {}))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--selftest-emulator", "--batch", "2", "--size", "16",
           "--steps", str(steps), "--warmup", str(warmup), "--loss", "ours_proposed"] + list(extra)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, [json.loads(ln) for ln in lines]


def test_bench_launches_its_own_two_ranks():
    r, docs = _bench_selftest(2)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert len(docs) == 1, r.stdout                                  # rank 0 prints ONE line
    d = docs[0]
    assert d["n_gpus"] == 2 and d["value"] is None and "SELFTEST" in d["data"] and d["config"]["global_batch"] == 4
    dp = d["dp"]
    assert dp["backend"] == "gloo" and dp["world_size_seen_by_the_process_group"] == 2
    assert dp["replicas_bit_identical_after_timed_region"] is True
    assert len(dp["per_rank_last_loss"]) == 2 and dp["per_rank_last_loss"][0] != dp["per_rank_last_loss"][1]   # different shards


def test_bench_four_ranks_stay_bit_identical():
    """world size 4: both gradient buckets (decoders, then encoder) through four ranks whose shards have unequal numbers of labelled
    pixels (per-rank loss normalisation), three timed steps, replicas bit-identical afterwards"""
    r, docs = _bench_selftest(4, steps=3)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    dp = docs[0]["dp"]
    assert dp["world_size_seen_by_the_process_group"] == 4 and dp["replicas_bit_identical_after_timed_region"] is True
    assert len(set(dp["per_rank_last_loss"])) == 4
    assert all(np.isfinite(v) for v in dp["per_rank_last_loss"])


def test_bench_one_rank_through_the_force_dp_route():
    r, docs = _bench_selftest(1, extra=["--force-dp"], steps=1)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert docs[0]["dp"]["world_size_seen_by_the_process_group"] == 1


def test_bench_rank_failure_takes_the_job_down():
    """SURVEY 5 (fail fast on any rank error): rank 1 raises in its second timed step (the first one ran: the process group, both buckets and the optimiser have been through); the other rank sits in an all-reduce -- the job
    must end with a non-zero status well inside the timeout, without a result line"""
    import time
    t0 = time.time()
    r, docs = _bench_selftest(2, env_extra={"WSL_SELFTEST_FAIL": "1:1"}, timeout=600)
    assert r.returncode != 0
    assert not docs
    assert "injected failure of rank 1" in r.stderr
    assert time.time() - t0 < 300
