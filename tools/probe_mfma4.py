import ctypes as C, torch, sys
sys.path.insert(0, '/root/repo')
from wsl4mis_amd import _lib
L = _lib.lib()
L.wsl_debug_mfma4_probe.argtypes = [C.c_void_p]*4
a = torch.arange(64, dtype=torch.float32, device='cuda') + 1
b = (torch.arange(64, dtype=torch.float32, device='cuda') + 1) * 100
d = torch.zeros(256, device='cuda')
L.wsl_debug_mfma4_probe(a.data_ptr(), b.data_ptr(), d.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
d = d.view(64, 4).cpu()
# hypothesis: block = lane//4; D[lane][r] = A[4*block + r] * B[lane]
ok = True
for l in range(64):
    for r in range(4):
        exp = float(a[4*(l//4) + r] * b[l])
        if abs(float(d[l, r]) - exp) > 1e-3: ok = False
print("hypothesis D[lane][r] = A[4*(lane//4)+r]*B[lane]:", ok)
print(d[:8])
