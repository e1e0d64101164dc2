import torch
dev=torch.device('cuda:0')
def timed(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/iters*1e-3
for mb in (491, 2000):
    n=mb*1000*1000//4
    x=torch.empty(n,device=dev); y=torch.empty(n,device=dev)
    t=timed(lambda: x.fill_(1.0)); print(f'fill {mb} MB: {t*1e6:.1f} us {n*4/t/1e9:.0f} GB/s')
    t=timed(lambda: y.copy_(x)); print(f'copy {mb} MB: {t*1e6:.1f} us {2*n*4/t/1e9:.0f} GB/s (r+w)')
    t=timed(lambda: x.sum()); print(f'sum  {mb} MB: {t*1e6:.1f} us {n*4/t/1e9:.0f} GB/s')
