"""Host-side weight preparation: eval-mode BatchNorm folding and MFMA B-fragment packing.

BN fold (torch defaults eps=1e-5, pointnet2.py:164-168):
    W' = W * g / sqrt(var + eps),   b' = (b - mean) * g / sqrt(var + eps) + beta
computed in float64 and rounded once to float32.
"""
import numpy as np
import torch

BN_EPS = 1e-5
HALF_MAX, HALF_LOW = 65504.0, 2.0 ** -6        # csrc/cg_split.hpp


def fold_bn(w, b, bn=None):
    """w:(O,I) b:(O); bn = (weight, bias, running_mean, running_var) or None -> float64 arrays."""
    w = np.asarray(w, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if bn is None:
        return w, b
    g, beta, mu, var = [np.asarray(t, dtype=np.float64) for t in bn]
    s = g / np.sqrt(var + BN_EPS)
    return w * s[:, None], (b - mu) * s + beta


def pack_b(w):
    """W:(N,K) (K % 8 == 0) -> flat float32 array in B-fragment order
    Wp[nb][ks][lane][j] = W[nb*32 + (lane&31)][ks*8 + (lane>>5)*4 + j], rows zero padded to 32."""
    w = np.asarray(w, dtype=np.float32)
    n, k = w.shape
    assert k % 8 == 0, k
    nb = (n + 31) // 32
    wp = np.zeros((nb * 32, k), dtype=np.float32)
    wp[:n] = w
    # [nb, 32(l31), ks, 2(lhi), 4(j)] -> [nb, ks, lhi, l31, j]
    wp = wp.reshape(nb, 32, k // 8, 2, 4).transpose(0, 2, 3, 1, 4)
    return np.ascontiguousarray(wp).reshape(-1)


def bf16_split(w):
    """float32 array -> (hi, lo) uint16 bf16 bit patterns, round-to-nearest-even, lo = bf16(w - hi)."""
    def rne(x):
        bits = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
        return (((bits + 0x7fff + ((bits >> 16) & 1)) >> 16) & 0xffff).astype(np.uint16)
    w = np.asarray(w, dtype=np.float32)
    hi = rne(w)
    hi_f = (hi.astype(np.uint32) << 16).view(np.float32)
    lo = rne(w - hi_f)
    return hi, lo


def f16_split(w):
    """float32 array -> (hi, lo) uint16 IEEE-half bit patterns, round-to-nearest-even, lo = half(w - hi)."""
    w = np.asarray(w, dtype=np.float32)
    hi = w.astype(np.float16)
    lo = (w - hi.astype(np.float32)).astype(np.float16)
    return hi.view(np.uint16), lo.view(np.uint16)


def pack_b_bf16x3(w):
    return pack_b_split(w, 'bf16')


def pack_b_split(w, elem='bf16'):
    """W:(N,K) (K % 16 == 0) -> flat uint16 array Wp[nb][kc][2 (hi,lo)][lane][8] of 16-bit pieces (elem 'bf16' or 'f16'),
    element e of lane l = W[nb*32 + (l&31)][kc*16 + (l>>5)*8 + e], rows zero padded to 32 (see pointmlp_split.hip)."""
    w = np.asarray(w, dtype=np.float32)
    n, k = w.shape
    assert k % 16 == 0, k
    nb = (n + 31) // 32
    wp = np.zeros((nb * 32, k), dtype=np.float32)
    wp[:n] = w
    hi, lo = bf16_split(wp) if elem == 'bf16' else f16_split(wp)
    out = np.stack([hi, lo], 0)                                   # (2, nb*32, k)
    # [2, nb, 32(l31), kc, 2(lhi), 8(e)] -> [nb, kc, 2, lhi, l31, e]
    out = out.reshape(2, nb, 32, k // 16, 2, 8).transpose(1, 3, 0, 4, 2, 5)
    return np.ascontiguousarray(out).reshape(-1)


MX_NB_BYTES = 16640        # csrc/gen_l3_mx_asm.py: f16 fragments 8192 | lo8 4096 | hi8 4096 | scales 256


def _e4m3_bits(x):
    """float32 array -> uint8 OCP e4m3 bit patterns, round-to-nearest-even (|x| <= 448 by construction of the scales)."""
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.float8_e4m3fn).view(torch.uint8).numpy()


def pack_b_f16fp8x2(w):
    """W:(N,128) -> flat uint8 image of the 128 -> 1024 layer for the f16fp8x2 mode (csrc/pointmlp_split.hip, gen_l3_mx_asm.py).
    Per 32 output channels (16,640 B):
      [0,8192)       f16 hi pieces,   [kc 8][lane 64][8]: element e of lane l = half(W[nb*32 + (l&31)][kc*16 + (l>>5)*8 + e])
      [8192,12288)   lo8 = e4m3((W - half(W)) / 2^(e-19)),  [kh 2][u 2][lane 64][16 B]
      [12288,16384)  hi8 = e4m3(W / 2^(e-8)),               same layout
      [16384,16640)  scales, [lane 64][4 B] = E8M0 of (hi8 kh0, hi8 kh1, lo8 kh0, lo8 kh1) of unit u = l>>5
    One MX unit = 32 consecutive input channels of one output channel: unit u of k-half kh = channels [32*(2kh+u), +32); e = its
    frexp exponent (largest |w| in [2^(e-1), 2^e)), so the largest hi8 value lies in [128, 256) and every residual in (0, 128].
    Byte 4q+i of lane-half g = l>>5 within the unit holds channel 32*(2kh+u) + 8q + 4g + i (the order the producing layer's lanes
    hold them; the matrix instruction only needs both operands to agree)."""
    w = np.asarray(w, dtype=np.float32)
    n, k = w.shape
    assert k == 128 and n % 32 == 0, (n, k)
    nb = n // 32
    with np.errstate(over='ignore'):
        hi = w.astype(np.float16)
    lo = w - hi.astype(np.float32)
    blk = np.abs(w).reshape(n, 4, 32).max(axis=2)                       # (n, 4) unit maxima, unit index = 2*kh + u
    e = np.clip(np.frexp(blk)[1], -100, 100).astype(np.int32)           # frexp(0) = (0, 0)
    sc_hi = np.ldexp(np.float32(1), e - 8)[:, :, None]
    sc_lo = np.ldexp(np.float32(1), e - 19)[:, :, None]
    hi8 = _e4m3_bits((w.reshape(n, 4, 32) / sc_hi).reshape(n, 128))
    lo8 = _e4m3_bits((lo.reshape(n, 4, 32) / sc_lo).reshape(n, 128))

    def frag8(img):
        # channel index c = 32*(2kh+u) + 8q + 4g + i  ->  [nb, j, kh, u, q, g, i] -> [nb, kh, u, g, j, q, i]
        v = img.reshape(nb, 32, 2, 2, 4, 2, 4).transpose(0, 2, 3, 5, 1, 4, 6)
        return np.ascontiguousarray(v).reshape(nb, 4096)
    # f16 fragments: [nb, j, kc, g, e] -> [nb, kc, g, j, e]
    f16 = np.ascontiguousarray(hi.view(np.uint16).reshape(nb, 32, 8, 2, 8).transpose(0, 2, 3, 1, 4)).reshape(nb, 4096).view(np.uint8)
    # scales: lane (g, j): bytes (hi8 kh0, hi8 kh1, lo8 kh0, lo8 kh1) of unit u = g
    eb = e.reshape(nb, 32, 2, 2)                                       # [nb, j, kh, u]
    sc = np.stack([119 + eb[:, :, 0, :], 119 + eb[:, :, 1, :], 108 + eb[:, :, 0, :], 108 + eb[:, :, 1, :]], axis=-1)   # [nb, j, u, 4]
    sc = np.ascontiguousarray(sc.transpose(0, 2, 1, 3)).astype(np.uint8).reshape(nb, 256)
    out = np.concatenate([f16, frag8(lo8), frag8(hi8), sc], axis=1)
    assert out.shape == (nb, MX_NB_BYTES)
    return np.ascontiguousarray(out).reshape(-1)


def _get(sd, name):
    return sd[name].detach().cpu().double().numpy()


def _bn(sd, p):
    return (_get(sd, p + '.weight'), _get(sd, p + '.bias'), _get(sd, p + '.running_mean'), _get(sd, p + '.running_var'))


def _conv(sd, name):
    w = _get(sd, name + '.weight')
    return w.reshape(w.shape[0], -1), _get(sd, name + '.bias')


class DeviceWeights:
    """Folded + packed weights of one PointNet encoder (+ head) resident in HBM."""

    def __init__(self, device):
        self.device = device
        self.t = {}
        self.half_ok = {}       # name of a half ('.h') image -> usable by the f16x3 kernels (see put_half)

    def put(self, name, arr):
        self.t[name] = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)).to(self.device)
        self.__dict__.pop('_cls_c', None)      # engine._cls_forward_one_call caches raw device pointers of these tensors

    def put_split(self, name, w):
        """Split images of a (N,K) weight matrix (int16 storage of the 16-bit patterns): `name` ('....s') holds the bf16 pieces,
        the same name ending in '.h' the IEEE-half pieces."""
        assert name.endswith('.s'), name
        self.t[name] = torch.from_numpy(pack_b_split(w, 'bf16').view(np.int16)).to(self.device)
        self.put_half(name[:-2] + '.h', w)

    def put_half(self, name, w):
        """IEEE-half split image of a folded weight matrix + the pre-screen of its range: the image is used by the f16x3
        kernels only if every weight is finite in half (|w| < 65504 after BN folding -- a blown-up BN scale can exceed it) and
        the matrix is not uniformly tiny (max |w| >= 2^-6: below that the lo pieces sit in the half subnormals).  Layers that
        fail the screen run with bf16 pieces / exact f32 instead (engine._dense, engine._encoder_forward_split)."""
        w = np.asarray(w, dtype=np.float32)
        if not np.isfinite(w).all():
            raise ValueError(f'non-finite folded weights in layer {name[:-2]} (checkpoint or BatchNorm statistics are corrupt)')
        amax = float(np.abs(w).max()) if w.size else 0.0
        self.half_ok[name] = bool(amax < HALF_MAX and amax >= HALF_LOW)
        with np.errstate(over='ignore'):
            self.t[name] = torch.from_numpy(pack_b_split(w, 'f16').view(np.int16)).to(self.device)

    def put_mx(self, name, w):
        """f16fp8x2 image ('....q') of a 128 -> 1024 per-point layer; usable under the same range screen as its half image."""
        self.t[name] = torch.from_numpy(pack_b_f16fp8x2(w)).to(self.device)

    def __contains__(self, k):
        return k in self.t

    def __getitem__(self, k):
        return self.t[k]


IN_CH = 6      # input width of the fused per-point passes (xyz + normal); narrower inputs (the reference's default channel=3 ... 5)
               # ride zero-padded: a zero weight column times a zero input adds exactly +0 to an fmaf chain


def _pad_first_layer(w, what):
    """First-layer weights (64, cin) of an STN3d / encoder conv1 -> (64, 6): cin = 3..6 input channels (pointnet2.py:153,227: the
    reference default is channel=3), the missing ones as zero columns (the input is zero-padded to match, engine.pad_points)."""
    cin = w.shape[1]
    if not 3 <= cin <= IN_CH:
        raise NotImplementedError(f'{what}: the fused HIP passes take 3..6 input channels (xyz [+ up to 3 features]), got {cin}')
    return np.concatenate([w, np.zeros((w.shape[0], IN_CH - cin))], axis=1) if cin < IN_CH else w


def _prepare_tnet(W, sd, q, tag, k):
    """Fold/pack one STN3d (k=3) / STNkd (k=64) (pointnet2.py:153-223): parameters under prefix q, stored under tag."""
    w, b = fold_bn(*_conv(sd, q + 'conv1'), _bn(sd, q + 'bn1'))
    if k == 3:
        W.cin = w.shape[1]
        W.put(tag + '.w1', _pad_first_layer(w, 'STN3d.conv1')); W.put(tag + '.b1', b)          # 6 -> 64 on VALU, unpacked
    else:
        W.put(tag + '.wm', pack_b(w)); W.put(tag + '.bm', b)   # 64 -> 64, packed
        W.put_split(tag + '.wm.s', w)
    w, b = fold_bn(*_conv(sd, q + 'conv2'), _bn(sd, q + 'bn2'))
    W.put(tag + '.w2', pack_b(w)); W.put(tag + '.b2', b); W.put_split(tag + '.w2.s', w)
    w, b = fold_bn(*_conv(sd, q + 'conv3'), _bn(sd, q + 'bn3'))
    W.put(tag + '.w3', pack_b(w)); W.put(tag + '.b3', b); W.put_split(tag + '.w3.s', w); W.put_mx(tag + '.w3.q', w)
    w, b = fold_bn(_get(sd, q + 'fc1.weight'), _get(sd, q + 'fc1.bias'), _bn(sd, q + 'bn4'))
    W.put(tag + '.fc1', pack_b(w)); W.put(tag + '.fc1b', b); W.put_half(tag + '.fc1.h', w)
    w, b = fold_bn(_get(sd, q + 'fc2.weight'), _get(sd, q + 'fc2.bias'), _bn(sd, q + 'bn5'))
    W.put(tag + '.fc2', pack_b(w)); W.put(tag + '.fc2b', b); W.put_half(tag + '.fc2.h', w)
    w, b = _get(sd, q + 'fc3.weight'), _get(sd, q + 'fc3.bias')
    if k == 64:
        # emit the 64x64 feature transform transposed (column j*64+i holds T[i][j]) so the point kernels read a
        # lane's consecutive-k operand elements with 16-byte loads; the identity stays on the diagonal.
        perm = np.arange(4096).reshape(64, 64).T.reshape(-1)
        w, b = w[perm], b[perm]
    W.put(tag + '.fc3', pack_b(w)); W.put(tag + '.fc3b', b)
    if k == 64:
        W.put_half(tag + '.fc3.h', w)       # 256 -> 4096; the 9-wide stn.fc3 stays exact f32


def prepare_stnkd(sd, device):
    """A free-standing STNkd(k) (pointnet2.py:189-223) on an arbitrary k-channel tensor: every layer as a packed GEMM (the fused
    pass <1> that contains STNkd inside the encoder starts from the 6-channel points, so it cannot serve this module).  k is
    zero-padded to a multiple of 8 (the GEMM kernel's K granularity); exact-f32 images only."""
    sd = {k_.replace('module.', ''): v for k_, v in sd.items()}
    W = DeviceWeights(device)
    w, b = fold_bn(*_conv(sd, 'conv1'), _bn(sd, 'bn1'))
    k = w.shape[1]
    W.k, W.k_pad = k, (k + 7) // 8 * 8
    W.put('c1', pack_b(np.concatenate([w, np.zeros((64, W.k_pad - k))], axis=1))); W.put('c1b', b)
    w, b = fold_bn(*_conv(sd, 'conv2'), _bn(sd, 'bn2')); W.put('c2', pack_b(w)); W.put('c2b', b)
    w, b = fold_bn(*_conv(sd, 'conv3'), _bn(sd, 'bn3')); W.put('c3', pack_b(w)); W.put('c3b', b)
    w, b = fold_bn(_get(sd, 'fc1.weight'), _get(sd, 'fc1.bias'), _bn(sd, 'bn4')); W.put('fc1', pack_b(w)); W.put('fc1b', b)
    w, b = fold_bn(_get(sd, 'fc2.weight'), _get(sd, 'fc2.bias'), _bn(sd, 'bn5')); W.put('fc2', pack_b(w)); W.put('fc2b', b)
    W.put('fc3', pack_b(_get(sd, 'fc3.weight'))); W.put('fc3b', _get(sd, 'fc3.bias'))
    return W


def prepare_stn3d(sd, device):
    """A standalone STN3d(channel=3..6) (pointnet2.py:153-185)."""
    sd = {k.replace('module.', ''): v for k, v in sd.items()}
    W = DeviceWeights(device)
    _prepare_tnet(W, sd, '', 'stn', 3)
    return W


def prepare_encoder(sd, prefix, device, out=None):
    """Fold/pack PointNetEncoder weights (pointnet2.py:226-238): channel = 3..6, with or without the feature transform (the
    reference default is feature_transform=False: no `fstn` parameters -- W.has_fstn tells the engine to skip that pass)."""
    sd = {k.replace('module.', ''): v for k, v in sd.items()}
    W = out or DeviceWeights(device)
    p = prefix
    _prepare_tnet(W, sd, p + 'stn.', 'stn', 3)
    W.has_fstn = (p + 'fstn.conv1.weight') in sd
    if W.has_fstn:
        _prepare_tnet(W, sd, p + 'fstn.', 'fstn', 64)
    w, b = fold_bn(*_conv(sd, p + 'conv1'), _bn(sd, p + 'bn1'))
    if w.shape[1] != W.cin:
        raise ValueError(f'PointNetEncoder: conv1 takes {w.shape[1]} channels but its STN3d {W.cin}')
    W.put('enc.w1', _pad_first_layer(w, 'PointNetEncoder.conv1')); W.put('enc.b1', b)
    w, b = fold_bn(*_conv(sd, p + 'conv2'), _bn(sd, p + 'bn2'))
    W.put('enc.w2', pack_b(w)); W.put('enc.b2', b); W.put_split('enc.w2.s', w)
    w, b = fold_bn(*_conv(sd, p + 'conv3'), _bn(sd, p + 'bn3'))
    W.put('enc.w3', pack_b(w)); W.put('enc.b3', b); W.put_split('enc.w3.s', w); W.put_mx('enc.w3.q', w)
    return W


def prepare_cls(sd, device):
    """PointNetCls (pointnet2.py:275-299)."""
    sd = {k.replace('module.', ''): v for k, v in sd.items()}
    W = prepare_encoder(sd, 'feat.', device)
    w, b = fold_bn(_get(sd, 'fc1.weight'), _get(sd, 'fc1.bias'), _bn(sd, 'bn1'))
    W.put('head.fc1', pack_b(w)); W.put('head.fc1b', b); W.put_half('head.fc1.h', w)
    w, b = fold_bn(_get(sd, 'fc2.weight'), _get(sd, 'fc2.bias'), _bn(sd, 'bn2'))
    W.put('head.fc2', pack_b(w)); W.put('head.fc2b', b); W.put_half('head.fc2.h', w)
    w, b = _get(sd, 'fc3.weight'), _get(sd, 'fc3.bias')
    W.put('head.fc3', pack_b(w)); W.put('head.fc3b', b)
    W.n_out = w.shape[0]
    return W


def prepare_seg(sd, device):
    """PointNetSeg (pointnet2.py:302-329).  conv1 (1088->512) is split into the global-feature part
    (first 1024 input channels, evaluated once per cloud and used as a per-cloud bias) and the
    point-feature part (last 64 channels), exactly the cat([global, pointfeat]) order of :271."""
    sd = {k.replace('module.', ''): v for k, v in sd.items()}
    W = prepare_encoder(sd, 'feat.', device)
    w, b = fold_bn(*_conv(sd, 'conv1'), _bn(sd, 'bn1'))
    W.put('seg.c1g', pack_b(w[:, :1024])); W.put('seg.c1b', b); W.put_split('seg.c1g.s', w[:, :1024])
    W.put('seg.c1p', pack_b(w[:, 1024:])); W.put_split('seg.c1p.s', w[:, 1024:])
    w, b = fold_bn(*_conv(sd, 'conv2'), _bn(sd, 'bn2'))
    W.put('seg.c2', pack_b(w)); W.put_split('seg.c2.s', w); W.put('seg.c2b', b)
    w, b = fold_bn(*_conv(sd, 'conv3'), _bn(sd, 'bn3'))
    W.put('seg.c3', pack_b(w)); W.put_split('seg.c3.s', w); W.put('seg.c3b', b)
    w, b = _conv(sd, 'conv4')
    W.put('seg.c4', pack_b(w)); W.put_split('seg.c4.s', w); W.put('seg.c4b', b)
    W.n_out = w.shape[0]
    return W
