"""The PointNet++ encoder (catgrasp_amd.pointnet2.PointNet2Encoder) on 20,000-point clouds: HIP-event time per level and per stage,
algorithmic flops per level, fraction of the f32 MFMA peak of every fused kernel.

    python scripts/pp_encoder_profile.py out.json [--msg] [--trace]       (on the GPU box)
--trace: only run the forwards (3 per batch size) so that `rocprofv3 --kernel-trace --stats` sees them."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
from catgrasp_amd import pointnet2 as p2            # noqa: E402
from catgrasp_amd import primitives as prim        # noqa: E402

PEAK = 157.3e12
dev = torch.device('cuda:0')
msg = '--msg' in sys.argv
trace = '--trace' in sys.argv
out_path = next((a for a in sys.argv[1:] if not a.startswith('--')), None)


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def mlp_flops(cin, mlp):
    f, prev = 0, cin
    for c in mlp:
        f += 2 * prev * c; prev = c
    return f


N = 20000
torch.manual_seed(0)
enc = p2.PointNet2Encoder(channel=6, msg=msg).to(dev).eval()
p2.VALIDATE_INPUTS = False
rows = []
for B in (1, 8, 16):
    rng = np.random.default_rng(B)
    x = torch.from_numpy(rng.uniform(-0.6, 0.6, size=(B, N, 6)).astype(np.float32)).to(dev)
    start = (torch.randint(0, N, (B,)), torch.randint(0, 512, (B,)))
    with torch.no_grad():
        if trace:
            for _ in range(3):
                enc(x, start=start)
            torch.cuda.synchronize()
            continue
        xyz, feats = x[:, :, :3].contiguous(), x[:, :, 3:].contiguous()
        whole = timed(lambda: enc(x, start=start))
        r = {'clouds': B, 'points': N, 'msg': msg, 'encoder_ms': round(whole * 1e3, 4)}
        if not msg:
            _, l1_xyz = p2.farthest_point_sample(xyz, 512, start[0], return_xyz=True)
            idx1 = p2.query_ball_point(0.2, 32, xyz, l1_xyz)
            W1, W2, W3 = enc.sa1._weights(dev), enc.sa2._weights(dev), enc.sa3._weights(dev)
            l1 = prim.group_mlp_max(xyz, feats, l1_xyz, idx1, W1, channels_last=True)
            _, l2_xyz = p2.farthest_point_sample(l1_xyz, 128, start[1], return_xyz=True)
            idx2 = p2.query_ball_point(0.4, 64, l1_xyz, l2_xyz)
            l2 = prim.group_mlp_max(l1_xyz, l1, l2_xyz, idx2, W2, channels_last=True)
            st = {
                'fps1_20000_to_512': timed(lambda: p2.farthest_point_sample(xyz, 512, start[0], return_xyz=True)),
                'ball1_r0.2_k32': timed(lambda: p2.query_ball_point(0.2, 32, xyz, l1_xyz)),
                'sa1_9_64_64_128': timed(lambda: prim.group_mlp_max(xyz, feats, l1_xyz, idx1, W1, check_indices=False, channels_last=True)),
                'fps2_512_to_128': timed(lambda: p2.farthest_point_sample(l1_xyz, 128, start[1], return_xyz=True)),
                'ball2_r0.4_k64': timed(lambda: p2.query_ball_point(0.4, 64, l1_xyz, l2_xyz)),
                'sa2_131_128_128_256': timed(lambda: prim.group_mlp_max(l1_xyz, l1, l2_xyz, idx2, W2, check_indices=False, channels_last=True)),
                'sa3_all_259_256_512_1024_gemm_chain': timed(lambda: prim.group_all_mlp_max(l2_xyz, l2, W3, fused=False)),
                'sa3_all_259_256_512_1024_fused_tile': timed(lambda: prim.group_all_mlp_max(l2_xyz, l2, W3, fused=True)),
            }
            fl = {'sa1_9_64_64_128': B * 512 * 32 * mlp_flops(9, [64, 64, 128]), 'sa2_131_128_128_256': B * 128 * 64 * mlp_flops(131, [128, 128, 256]),
                  'sa3_all_259_256_512_1024_gemm_chain': B * 128 * mlp_flops(259, [256, 512, 1024])}
            fl['sa3_all_259_256_512_1024_fused_tile'] = fl['sa3_all_259_256_512_1024_gemm_chain']
            r['stages_us'] = {k: round(v * 1e6, 2) for k, v in st.items()}
            r['algorithmic_gflop'] = {k: round(v / 1e9, 4) for k, v in fl.items()}
            r['tflops'] = {k: round(fl[k] / st[k] / 1e12, 2) for k in fl}
            r['frac_of_f32_mfma_peak'] = {k: round(fl[k] / st[k] / PEAK, 4) for k in fl}
        rows.append(r)
        print(json.dumps(r), flush=True)
if out_path and not trace:
    with open(out_path, 'w') as f:
        json.dump({'what': 'PointNet2Encoder, HIP events around each stage as the module issues it (20 launches back to back)', 'rows': rows}, f, indent=1)
