import sys, time, random, torch
sys.path.insert(0, "/root/repo")
from wsl4mis_amd.engine import TrainEngine
from wsl4mis_amd.synthetic import batch
dev = torch.device("cuda", 0)
torch.manual_seed(2022)
eng = TrainEngine("unet_cct", 1, 4, base_lr=0.01, max_iterations=60000, loss="pce_gatedcrf", crf_radius=5)
x, lab = batch(64, 256, 256, 2022, dev)
for _ in range(5): eng.step(x, lab, 0.5)
torch.cuda.synchronize()
hs = []
t0 = time.perf_counter()
for _ in range(20):
    a = time.perf_counter(); eng.step(x, lab, 0.5); hs.append(time.perf_counter() - a)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue per step: %.2f ms (min %.2f); enqueue of 20 steps %.1f ms, device drained %.1f ms later; total/step %.2f ms" % (1e3*sum(hs)/20, 1e3*min(hs), 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t2-t0)/20))
