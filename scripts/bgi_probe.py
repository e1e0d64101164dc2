"""dev probe: what bounds build_grasp_input? random vs sequential vs constant ids."""
import sys, torch
sys.path.insert(0, '.')
from catgrasp_amd import ops
dev = torch.device('cuda:0')
def timed(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3
g = torch.Generator(device=dev); g.manual_seed(0)
M, G, NP = 20000, 10000, 2048
xyz = torch.randn(M, 3, device=dev, generator=g); nrm = torch.randn(M, 3, device=dev, generator=g)
obj = (torch.arange(G, device=dev) * 8 // G)[:, None]      # consecutive candidates share an object (bench / pipeline order)
obj_r = torch.randint(0, 8, (G, 1), device=dev, generator=g)
pinv = torch.randn(G, 12, device=dev, generator=g)
out = torch.empty(G, NP, 6, device=dev)
cases = {'random': (obj * 2500 + torch.randint(0, 2500, (G, NP), device=dev, generator=g)).int().contiguous(),
         'sequential': (obj * 2500 + torch.arange(NP, device=dev)[None]).int().contiguous(),
         'constant': (obj * 2500 + torch.zeros(G, NP, device=dev, dtype=torch.long)).int().contiguous(),
         'random, objects interleaved': (obj_r * 2500 + torch.randint(0, 2500, (G, NP), device=dev, generator=g)).int().contiguous(),
         'wide (fallback path)': torch.randint(0, M, (G, NP), device=dev, generator=g).int().contiguous()}
for k, ids in cases.items():
    t = timed(lambda: ops.build_grasp_input(xyz, nrm, ids, pinv, out=out))
    print(f'{k:22s} {t*1e6:8.1f} us  {G*(NP*28+48)/t/1e9:7.0f} GB/s')
mean = torch.zeros(6, device=dev); istd = torch.ones(6, device=dev)
t = timed(lambda: ops.build_grasp_input(xyz, nrm, cases['random'], pinv, mean, istd, out=out)); print(f'random+normaliser {t*1e6:8.1f} us')
