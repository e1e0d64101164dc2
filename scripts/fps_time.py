"""dev: us per round of farthest_point_sample, N = 20,000 -> 1,024, one cloud and the 8 clouds of C3 (CATGRASP_AMD_LIB selects the build)."""
import sys, time
import torch
sys.path.insert(0, '.')
from catgrasp_amd import primitives
dev = torch.device('cuda:0')
torch.manual_seed(0)
for B in (1, 8):
    pts = torch.rand(B, 20000, 3, device=dev)
    st = torch.zeros(B, dtype=torch.long, device=dev)
    primitives.farthest_point_sample(pts, 1024, start=st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        out = primitives.farthest_point_sample(pts, 1024, start=st)
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
    print(f'B={B} us_per_round {t / 1024 * 1e6:.3f} checksum {int(out.sum())}', flush=True)
