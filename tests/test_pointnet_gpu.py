"""GPU parity: HIP PointNetCls / PointNetSeg forward vs the CPU oracle (oracle/pointnet_ref.py),
which is itself pinned to the reference pointnet2.py by tests/golden.  Tolerance: 1e-4 absolute on
logits (BASELINE.json north_star: "within 1e-4 fp32")."""
import numpy as np
import pytest
import torch

from catgrasp_amd import synth
from oracle import pointnet_ref as oref

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _close(a, b, what):
    """|a-b| <= 1e-4 * max(1, |b|): 1e-4 absolute for O(1) values, 1e-4 relative for large ones."""
    err = ((a - b).abs() / b.abs().clamp(min=1.0)).max().item()
    assert err <= TOL, f'{what}: max scaled err {err}'


def _inputs(B, N, seed, scale=0.5):
    rng = np.random.default_rng(seed)
    x = rng.normal(0, scale, (B, N, 6)).astype(np.float32)
    return torch.from_numpy(x)


@pytest.mark.parametrize('B,N,gain', [(3, 256, 1.6), (2, 2048, 1.6), (5, 100, 1.0), (1, 64, 1.6), (70, 130, 1.7), (3, 700, 1.7), (1, 31, 1.6), (2, 257, 1.6)])
def test_cls_forward_matches_oracle(cuda_device, mlp_precision, B, N, gain):
    from catgrasp_amd import engine, folding
    sd = synth.make_state_dict('cls', 6, 10, seed=11, gain=gain)
    x = _inputs(B, N, 5)
    ref_logits, ref_tf = oref.pointnet_cls_forward(sd, x)
    W = folding.prepare_cls(sd, cuda_device)
    logits, tf = engine.cls_forward(W, x.to(cuda_device))
    torch.cuda.synchronize()
    assert logits.shape == (B, 10) and tf.shape == (B, 64, 64)
    _close(logits.cpu(), ref_logits, 'logits')
    _close(tf.cpu(), ref_tf, 'trans_feat')
    # the grasp-Q *scores* (softmax probabilities, predicter.py:86) within 1e-4 absolute
    perr = (torch.softmax(logits.cpu(), 1) - torch.softmax(ref_logits, 1)).abs().max().item()
    assert perr <= TOL, f'probs max abs err {perr}'
    # against the float64 evaluation: the exact-f32 HIP path is as close to the truth as the reference dtype, and so is the
    # split-half path (22 significant bits per operand); the split-bf16 path (16 bits) and the 2-unit f16 + e4m3 path stay within a
    # small multiple of it
    y64, _ = oref.pointnet_cls_forward(sd, x, torch.float64)
    e_hip = ((logits.cpu().double() - y64).abs() / y64.abs().clamp(min=1)).max().item()
    e_ref = ((ref_logits.double() - y64).abs() / y64.abs().clamp(min=1)).max().item()
    assert e_hip <= {'f32': max(4 * e_ref, 1e-5), 'f16x3': max(8 * e_ref, 1e-5), 'bf16x3': 5e-5, 'f16fp8x2': 6e-5}[mlp_precision], (e_hip, e_ref)


@pytest.mark.parametrize('B,N', [(1, 512), (2, 1000), (1, 8192)])
def test_seg_forward_matches_oracle(cuda_device, B, N):
    from catgrasp_amd import engine, folding
    sd = synth.make_state_dict('seg', 6, 300, seed=12)
    x = _inputs(B, N, 6)
    ref_y, ref_tf = oref.pointnet_seg_forward(sd, x)
    W = folding.prepare_seg(sd, cuda_device)
    y, tf = engine.seg_forward(W, x.to(cuda_device))
    torch.cuda.synchronize()
    assert y.shape == (B, N, 300)
    _close(y.cpu(), ref_y, 'seg logits')
    _close(tf.cpu(), ref_tf, 'trans_feat')


def test_dense_layer_bits_do_not_depend_on_the_kernel_or_the_batch(cuda_device):
    """cg_gemm_bias_act takes a wavefront-per-tile kernel for few output tiles (<= 2,048 of 32 x 32) and the workgroup-tile kernel
    beyond; both issue the same sequence of exact-f32 MFMAs per output element, so a row's result must not depend on how many rows
    were computed with it -- bit for bit, against a float64 product within float32 rounding, for the FC shapes of the networks, odd K
    tails (K = 72: not a multiple of the 8-k-step pipeline) and every epilogue option.  And the one-call PointNetCls forward
    (cg_pointnet_cls_forward, batches <= 1,024) == the launch-by-launch chain."""
    from catgrasp_amd import engine, folding, ops
    rng = np.random.default_rng(3)
    for K, N, relu, eye in ((1024, 512, True, 0), (512, 256, True, 0), (256, 4096, False, 64), (256, 10, False, 0), (72, 96, True, 0)):
        w = rng.normal(0, 0.05, (N, K)).astype(np.float32); b = rng.normal(0, 0.1, N).astype(np.float32)
        wp = torch.from_numpy(folding.pack_b(w)).to(cuda_device); bd = torch.from_numpy(b).to(cuda_device)
        nb = (N + 31) // 32
        m_small = max(32, (2048 // nb) * 32)                  # the largest row count the wavefront-per-tile kernel takes
        M = m_small + 64                                       # ... and one the tile kernel takes
        x = torch.from_numpy(rng.normal(0, 1, (M, K)).astype(np.float32)).to(cuda_device)
        y_big = ops.gemm_bias_act(x, wp, N, bd, relu=relu, eye_k=eye)
        for m in (1, 5, 33, m_small):
            y = ops.gemm_bias_act(x[:m].contiguous(), wp, N, bd, relu=relu, eye_k=eye)
            assert torch.equal(y, y_big[:m]), (K, N, m)
        ref = x[:64].double().cpu().numpy() @ w.astype(np.float64).T + b
        if eye:
            ref += np.eye(eye).reshape(-1)
        if relu:
            ref = np.maximum(ref, 0)
        assert np.abs(y_big[:64].double().cpu().numpy() - ref).max() < 2e-4
    sd = synth.make_state_dict('cls', 6, 10, seed=9)
    W = folding.prepare_cls(sd, cuda_device)
    x = torch.from_numpy(rng.normal(0, 0.3, (37, 300, 6)).astype(np.float32)).to(cuda_device)
    with engine.precision('f32'):
        one_call = engine.cls_forward(W, x)
        old, engine.FUSED_LAUNCH_MAX_B = engine.FUSED_LAUNCH_MAX_B, 0
        try:
            chain = engine.cls_forward(W, x)
        finally:
            engine.FUSED_LAUNCH_MAX_B = old
    assert torch.equal(one_call[0], chain[0]) and torch.equal(one_call[1], chain[1])


@pytest.mark.parametrize('N', [2048, 64, 300])
def test_a_pose_scored_alone_has_the_bits_it_has_inside_a_large_batch(cuda_device, N):
    """calls of a few poses divide the 1024 output channels of the 128 -> 1024 layer over 2 / 4 / 8 workgroups per point
    tile (pointmlp_max_kernel<MID, CS>), large batches do not; per output element the sequence of matrix FMAs is the same, so logits,
    feature transform and the NUNOCS head's input are bit-identical whatever the batch a cloud is scored in (1, 3, 5, 20 clouds cover
    CS = 8, 4, 2 and 1 at 2,048 points; 1 ... 200 clouds at 64 points)."""
    from catgrasp_amd import engine, folding
    sd = synth.make_state_dict('cls', 6, 10, seed=4)
    W = folding.prepare_cls(sd, cuda_device)
    x = _inputs(600, N, 9).to(cuda_device)
    big_logits, big_tf = engine.cls_forward(W, x)
    for B in (1, 3, 5, 20, 100, 200):
        logits, tf = engine.cls_forward(W, x[:B].contiguous())
        assert torch.equal(logits, big_logits[:B]) and torch.equal(tf, big_tf[:B]), B
    sds = synth.make_state_dict('seg', 6, 30, seed=5)
    Ws = folding.prepare_seg(sds, cuda_device)
    big = engine.seg_forward(Ws, x[:40].contiguous())[0]
    one = engine.seg_forward(Ws, x[:1].contiguous())[0]
    assert torch.equal(one, big[:1])
