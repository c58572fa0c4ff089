"""Data-parallel path on the CPU: two `gloo` ranks, each running the engine's split backward + bucketed all-reduce on
its shard (kernel sources in the host emulator), against the DDP-emulation fixture generated from the reference
(per-shard BatchNorm statistics and loss normalisation, gradients averaged -- SURVEY 8e)."""
import os
import socket
import subprocess
import sys

import numpy as np

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
