"""Quick GPU timing of cls_forward (dev tool)."""
import sys, time
sys.path.insert(0, '.')
import torch
from catgrasp_amd import engine, folding
dev = torch.device('cuda:0')
from catgrasp_amd import synth
sd = synth.make_state_dict('cls', 6, 10, seed=11)
W = folding.prepare_cls(sd, dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
if len(sys.argv) > 2:
    engine.set_precision(sys.argv[2])
x = (torch.randn(B, 2048, 6) * 0.5).to(dev)
for _ in range(2):
    engine.cls_forward(W, x)
torch.cuda.synchronize()
t = time.time(); n = 3
for _ in range(n):
    engine.cls_forward(W, x)
torch.cuda.synchronize()
dt = (time.time() - t) / n
print(f'{engine.PRECISION} tp={engine.TILE_POINTS} B={B} {dt*1e3:.2f} ms  {B/dt:.0f} cand/s  {B*1.7541e9/dt/1e12:.1f} TFLOP/s')
