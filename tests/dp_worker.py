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


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    torch.set_num_threads(1)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    from wsl4mis_amd import _lib
    _lib.use_library_for_tests(C.CDLL(os.path.join(ROOT, "tests", "emul", "libwslhip_emul.so")))
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
