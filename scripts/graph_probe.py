"""Probe: the exact-f32 PointNetCls forward of a few poses (12 launches from one C call) eagerly vs replayed from a HIP graph
(torch.cuda.CUDAGraph capturing the same ctypes launches): wall clock per forward incl. the synchronisation, and bit equality."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
from catgrasp_amd import engine, folding, ops, synth

dev = torch.device('cuda:0')
W = folding.prepare_cls(synth.make_state_dict('cls', 6, 10, seed=0), dev)
for B in (1, 16, 64):
    x = (torch.randn(B, 2048, 6) * 0.5).to(dev)
    with torch.no_grad():
        for _ in range(5):
            ref = engine.cls_forward(W, x)[0].clone()
        torch.cuda.synchronize()

        def eager():
            lg = engine.cls_forward(W, x)[0]
            return ops.softmax_pg(lg)[0]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = eager()
        torch.cuda.synchronize()

        def timeit(fn, n=200):
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn(); torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        te = timeit(eager)
        tg = timeit(g.replay)
        g.replay(); torch.cuda.synchronize()
        same = torch.equal(out, ops.softmax_pg(ref)[0])
        print(f'B={B}: eager {te:.3f} ms, graph replay {tg:.3f} ms per forward+softmax (synchronised each), identical: {same}')
