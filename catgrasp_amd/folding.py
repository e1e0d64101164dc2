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

    def __contains__(self, k):
        return k in self.t

    def __getitem__(self, k):
        return self.t[k]


def _prepare_tnet(W, sd, q, tag, k):
    """Fold/pack one STN3d (k=3) / STNkd (k=64) (pointnet2.py:153-223): parameters under prefix q, stored under tag."""
    w, b = fold_bn(*_conv(sd, q + 'conv1'), _bn(sd, q + 'bn1'))
    if k == 3:
        W.put(tag + '.w1', w); W.put(tag + '.b1', b)          # 6 -> 64 on VALU, unpacked
    else:
        W.put(tag + '.wm', pack_b(w)); W.put(tag + '.bm', b)   # 64 -> 64, packed
        W.put_split(tag + '.wm.s', w)
    w, b = fold_bn(*_conv(sd, q + 'conv2'), _bn(sd, q + 'bn2'))
    W.put(tag + '.w2', pack_b(w)); W.put(tag + '.b2', b); W.put_split(tag + '.w2.s', w)
    w, b = fold_bn(*_conv(sd, q + 'conv3'), _bn(sd, q + 'bn3'))
    W.put(tag + '.w3', pack_b(w)); W.put(tag + '.b3', b); W.put_split(tag + '.w3.s', w)
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


def prepare_stn3d(sd, device):
    """A standalone STN3d(channel=6) (pointnet2.py:153-185)."""
    sd = {k.replace('module.', ''): v for k, v in sd.items()}
    W = DeviceWeights(device)
    _prepare_tnet(W, sd, '', 'stn', 3)
    return W


def prepare_encoder(sd, prefix, device, out=None):
    """Fold/pack PointNetEncoder(feature_transform=True) weights (pointnet2.py:226-238)."""
    sd = {k.replace('module.', ''): v for k, v in sd.items()}
    W = out or DeviceWeights(device)
    p = prefix
    _prepare_tnet(W, sd, p + 'stn.', 'stn', 3)
    _prepare_tnet(W, sd, p + 'fstn.', 'fstn', 64)
    w, b = fold_bn(*_conv(sd, p + 'conv1'), _bn(sd, p + 'bn1'))
    W.put('enc.w1', w); W.put('enc.b1', b)
    w, b = fold_bn(*_conv(sd, p + 'conv2'), _bn(sd, p + 'bn2'))
    W.put('enc.w2', pack_b(w)); W.put('enc.b2', b); W.put_split('enc.w2.s', w)
    w, b = fold_bn(*_conv(sd, p + 'conv3'), _bn(sd, p + 'bn3'))
    W.put('enc.w3', pack_b(w)); W.put('enc.b3', b); W.put_split('enc.w3.s', w)
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
