"""dev: the FC-tail layers (1024->512, 512->256, 256->4096) for M = 1 .. 16384 rows through cg_gemm_bias_act, HIP events, 50 launches;
CATGRASP_AMD_GEMM_SMALL_TILES=0 forces the tile kernel, a large value the wavefront-per-tile kernel: the crossover behind SMALL_M."""
import sys
import numpy as np, torch
sys.path.insert(0, '.')
from catgrasp_amd import folding, ops
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
for K, N in ((1024, 512), (512, 256), (256, 4096)):
    wp = torch.from_numpy(folding.pack_b(rng.normal(0, 0.05, (N, K)).astype(np.float32))).to(dev)
    b = torch.zeros(N, device=dev)
    row = []
    for M in (1, 16, 64, 256, 1024, 2048, 4096, 16384):
        x = torch.randn(M, K, device=dev)
        for _ in range(3):
            y = ops.gemm_bias_act(x, wp, N, b, relu=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            y = ops.gemm_bias_act(x, wp, N, b, relu=True)
        e1.record(); torch.cuda.synchronize()
        row.append(f'M={M}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us (sum {float(y.double().sum()):.4f})')
    print(f'{K}->{N}: ' + '  '.join(row), flush=True)
