"""Worker of tests/test_dp.py: one rank of a world-size-2 data-parallel step on the CPU (gloo) with the kernel sources
running in the host emulator.  Writes the all-reduced gradient samples / parameters to an .npz for the parent test."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)


# generator seeds per (composition, rank): searched so that no pre-activation / pooling window of the shard lies within fp32
# noise of a LeakyReLU / max-pool kink; tests/test_dp.py re-checks the margins with the oracle
SHARD_SEEDS = {("pce_gatedcrf", 0): 106, ("pce_gatedcrf", 1): 137, ("mean_teacher", 0): 176, ("mean_teacher", 1): 159}


def shard_inputs(rank, kind, S=32, N=2, seed=None):
    """seeded inputs of one rank for the compositions that have no reference-generated DDP fixture (the parent test feeds
    the same tensors to the oracle)"""
    from oracle import torch_ref as R
    from wsl4mis_amd.synthetic import scribble_labels
    gen = torch.Generator().manual_seed(SHARD_SEEDS[(kind, rank)] if seed is None else seed)
    x = torch.rand((N, 1, S, S), generator=gen)
    lab = torch.from_numpy(scribble_labels(N, S, S, 40 + rank, share=0.08))
    em = [(torch.rand((N, 16 << l, S >> l, S >> l), generator=gen) >= R.DROP[l]).to(torch.uint8) for l in range(5)]
    cm = [(torch.rand((N, 16 << l), generator=gen) >= 0.5).float() * 2 for l in range(5)]
    em_t = [(torch.rand((N, 16 << l, S >> l, S >> l), generator=gen) >= R.DROP[l]).to(torch.uint8) for l in range(5)]
    noise = torch.clamp(torch.randn((N, 1, S, S), generator=gen) * 0.1, -0.2, 0.2)
    return {"x": x, "lab": lab, "em": em, "cm": cm, "em_t": em_t, "noise": noise, "beta": 0.41}


def run_kind(rank, world, outdir, kind, dev=None):
    """pce_gatedcrf (unet_cct, headline composition) / mean_teacher (config 4: the teacher forward and the gradient
    all-reduce in the same step) through the engine's data-parallel route"""
    from detinit import det_state
    from wsl4mis_amd.engine import TrainEngine
    split = kind.endswith("_split")                      # the same composition on the opt-in split-precision conv path
    kind = kind[:-6] if split else kind
    net = "unet_cct" if kind == "pce_gatedcrf" else "unet"
    eng = TrainEngine(net, 1, 4, base_lr=0.01, loss=kind, crf_radius=2, conv_precision="split_f16x3" if split else "f32")
    assert eng.dp and eng.world == world
    models = [eng.model] + ([eng.teacher] if eng.teacher is not None else [])
    for i, m in enumerate(models):
        vals = det_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, (9 + 13 * i) if rank == 0 else 777 + i)
        m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in vals.items()})
        dist.broadcast(m._param_arena, src=0)          # rank 1 was given OTHER weights: what the engine does at construction
        dist.broadcast(m._buf_arena, src=0)
    d = shard_inputs(rank, kind)
    if dev is not None:                                 # both ranks on ONE MI355X (gloo carries the collectives through the host)
        d = {k: ([t.to(dev) for t in v] if isinstance(v, list) else (v.to(dev) if torch.is_tensor(v) else v)) for k, v in d.items()}
    eng.it = 4500                                       # mean teacher: a non-trivial consistency weight and EMA alpha
    eng.model.set_dropout_masks(d["em"], d["cm"] if net == "unet_cct" else None)
    if eng.teacher is not None:
        eng.teacher.set_dropout_masks(d["em_t"], None)
    eng.forward_backward(d["x"], d["lab"], d["beta"], noise=d["noise"] if kind == "mean_teacher" else None)
    out = {"loss": np.float32(eng.losses()["loss"]), "grads": (eng.model.flat_grads() / world).cpu().numpy().copy()}
    eng.optimizer_step()
    out["params_after"] = eng.model.flat_params().cpu().numpy().copy()
    if eng.teacher is not None:
        out["teacher_after"] = eng.teacher.flat_params().cpu().numpy().copy()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    kind = sys.argv[5] if len(sys.argv) > 5 else "ours_proposed"
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from wsl4mis_amd import _lib
    gpu = len(sys.argv) > 6 and sys.argv[6] == "gpu"    # the real library, both ranks on cuda:0
    if gpu:
        torch.cuda.set_device(0)
    else:
        _lib.use_library_for_tests(C.CDLL(os.path.join(ROOT, "tests", "emul", "libwslhip_emul.so")))
    if kind != "ours_proposed":
        run_kind(rank, world, outdir, kind, torch.device("cuda", 0) if gpu else None)
        dist.destroy_process_group()
        return
    from detinit import det_state, sample_index
    from wsl4mis_amd.engine import TrainEngine
    g = np.load(os.path.join(ROOT, "tests", "golden", "g8_ddp.npz"))
    eng = TrainEngine("unet_cct", 1, 4, base_lr=0.01, loss="ours_proposed")
    sd = eng.model.state_dict()
    seed = 9 if rank == 0 else 1234          # rank 1 starts from OTHER weights: the engine must broadcast rank 0's
    vals = det_state({k: tuple(v.shape) for k, v in sd.items()}, seed)
    eng.model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in vals.items()})
    dist.broadcast(eng.model._param_arena, src=0)
    dist.broadcast(eng.model._buf_arena, src=0)
    x = torch.from_numpy(g["x"][2 * rank:2 * rank + 2])
    lab = torch.from_numpy(g["label"][2 * rank:2 * rank + 2])
    eng.model.set_dropout_masks([torch.from_numpy(g[f"r{rank}_emask{i}"]) for i in range(5)],
                                [torch.from_numpy(g[f"r{rank}_cmask{i}"]) for i in range(5)])
    eng.forward_backward(x, lab, float(g["beta"]))
    loss = eng.losses()["loss"]
    grads = (eng.model.flat_grads() / world).numpy().copy()
    out = {"loss": np.float32(loss)}
    for (p, off, n, shape), (k, _) in zip(eng.model._plist, eng.model.named_parameters()):
        out["g." + k] = grads[off:off + n][sample_index(n)]
    eng.optimizer_step()
    out["params_after"] = eng.model.flat_params().numpy().copy()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
