"""sa_tile_kernel alone for rocprofv3 --pmc passes: the single-scale second level (131 -> 128-128-256, K = 64, S = 128) at 64 clouds.
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY ... --kernel-include-regex sa_tile -- python scripts/pmc_sa_tile.py"""
import sys

import torch

sys.path.insert(0, '.')
from catgrasp_amd import pointnet2 as p2            # noqa: E402
from catgrasp_amd import primitives as prim        # noqa: E402

dev = torch.device('cuda:0')
B, N, S, K, D, mlp = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 512, 128, 64, 128, [128, 128, 256]
g = torch.Generator(device=dev); g.manual_seed(0)
xyz = torch.rand(B, N, 3, device=dev, generator=g); pts = torch.randn(B, N, D, device=dev, generator=g)
new_xyz = xyz[:, :S].contiguous(); idx = torch.randint(0, N, (B, S, K), device=dev, generator=g)
torch.manual_seed(0)
sa = p2.PointNetSetAbstraction(S, 0.2, K, 3 + D, mlp).to(dev).eval()
W = prim.SetAbstractionWeights(p2._sa_layers_from_state(sa.state_dict(), 'mlp_', 3), 3 + D, dev, kind='tile')
out = torch.empty(B, S, 256, device=dev)
for _ in range(6):
    prim.group_mlp_max(xyz, pts, new_xyz, idx, W, check_indices=False, channels_last=True, out=out)
torch.cuda.synchronize()
print('pmc_sa_tile done')
