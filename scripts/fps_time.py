"""dev: us per round of farthest_point_sample for `N:S` pairs on the command line (default 20000:1024), one cloud and 8 clouds in one
launch (CATGRASP_AMD_LIB selects the build)."""
import sys, time
import torch
sys.path.insert(0, '.')
from catgrasp_amd import primitives
dev = torch.device('cuda:0')
torch.manual_seed(0)
sizes = [(int(a.split(':')[0]), int(a.split(':')[1])) for a in sys.argv[1:]] or [(20000, 1024)]
for N, S in sizes:
    for B in (1, 8):
        pts = torch.rand(B, N, 3, device=dev)
        st = torch.zeros(B, dtype=torch.long, device=dev)
        primitives.farthest_point_sample(pts, S, start=st)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            out = primitives.farthest_point_sample(pts, S, start=st)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
        print(f'N={N} S={S} B={B} us_per_round {t / S * 1e6:.3f} checksum {int(out.sum())}', flush=True)
