"""HIP-event time of the device resampling draw (cg_draw_resample_ids), the step's launch shape: 6,250 rows of 2,048 of 2,500."""
import sys
import torch
sys.path.insert(0, '.')
from catgrasp_amd import transforms
dev = torch.device('cuda:0')
for n_valid, n_pts, rows in ((2500, 2048, 6250), (2500, 2048, 50000), (2048, 2048, 16), (20000, 2048, 6250)):
    out = torch.empty((rows, n_pts), dtype=torch.int32, device=dev)
    for _ in range(3):
        transforms.draw_ids_device(n_valid, n_pts, rows, dev, seed=5, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        transforms.draw_ids_device(n_valid, n_pts, rows, dev, seed=5, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'n_valid={n_valid} n_pts={n_pts} rows={rows}: {ms * 1e3:.1f} us per launch, {rows * n_pts / ms / 1e6:.1f} G indices/s, {rows * n_pts * 4 / ms / 1e6:.0f} GB/s written')
