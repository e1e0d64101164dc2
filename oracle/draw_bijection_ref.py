"""TEST INFRASTRUCTURE ONLY (oracle) -- numpy restatement of the device resampling draw (csrc/hostprep.hip: draw_ids_bijection_kernel),
the counter-based replacement of `np.random.choice(M, n_pts, replace=False)` per candidate (dataset_grasp.py:72-73) on the
rng='device' path.  Same integer arithmetic, so the device rows must equal these bit for bit (tests/test_predicter_gpu.py); the CPU
suite runs the statistics on this restatement (tests/test_host_properties.py).  Nothing here is imported by the product."""
import numpy as np

ROUNDS = 12
M32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c, k0, k1):
    """c: (..., 4) uint32 counters; k0, k1: python ints.  -> (..., 4) uint32 (Salmon et al., SC'11)."""
    c = np.asarray(c, dtype=np.uint64).copy()
    k0, k1 = np.uint64(k0 & 0xFFFFFFFF), np.uint64(k1 & 0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[..., 0]
        p1 = np.uint64(0xCD9E8D57) * c[..., 2]
        n = np.empty_like(c)
        n[..., 0] = ((p1 >> np.uint64(32)) ^ c[..., 1] ^ k0) & M32
        n[..., 1] = p1 & M32
        n[..., 2] = ((p0 >> np.uint64(32)) ^ c[..., 3] ^ k1) & M32
        n[..., 3] = p0 & M32
        c = n
        k0 = (k0 + np.uint64(0x9E3779B9)) & M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & M32
    return c.astype(np.uint32)


M24 = np.uint64(0xFFFFFF)


def _mul24(a, b):
    """v_mul_u32_u24: the low 32 bits of the product of the operands' low 24 bits."""
    return ((a & M24) * (b & M24)) & M32


def _mix(x, k):
    """the round function's hash: two 24-bit multiplies (full-rate on gfx950; a 32-bit integer multiply is quarter rate)"""
    x = x + (k & np.uint64(0x7FFFFF))              # half < 2^15, key 23 bits: < 2^24
    h = _mul24(x, np.uint64(0x9E3779))
    h ^= h >> np.uint64(15)
    h = _mul24(h >> np.uint64(8), np.uint64(0x85EBCB))
    h ^= h >> np.uint64(13)
    return h


def half_bits(n_valid):
    bits = 2
    while bits < 30 and (1 << bits) < n_valid:
        bits += 2
    return bits // 2


def draw_rows(n_valid, n_pts, count, seed, base=0, row_offset=0):
    """-> (count, n_pts) int32: row r = the first n_pts values of the keyed permutation of [0, n_valid) of global row row_offset + r."""
    assert n_valid >= n_pts
    hb = np.uint64(half_bits(n_valid))
    hmask = np.uint64((1 << int(hb)) - 1)
    rows = np.arange(count, dtype=np.uint64) + np.uint64(row_offset)
    ctr = np.zeros((count, ROUNDS // 4, 4), dtype=np.uint64)
    ctr[..., 0] = (rows & M32)[:, None]
    ctr[..., 1] = np.arange(ROUNDS // 4, dtype=np.uint64)[None, :]
    ctr[..., 2] = (rows >> np.uint64(32))[:, None]
    ctr[..., 3] = np.uint64(0x42494A43)
    rk = philox4x32_10(ctr, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF).reshape(count, ROUNDS).astype(np.uint64)
    v = np.broadcast_to(np.arange(n_pts, dtype=np.uint64), (count, n_pts)).copy()
    todo = np.ones((count, n_pts), dtype=bool)
    while todo.any():
        rr, cc = np.nonzero(todo)
        x = v[rr, cc]
        l, r = x >> hb, x & hmask
        for q in range(ROUNDS):
            f = _mix(r, rk[rr, q]) >> (np.uint64(32) - hb)
            l, r = r, l ^ f
        x = (l << hb) | r
        v[rr, cc] = x
        todo[rr, cc] = x >= np.uint64(n_valid)
    return (v.astype(np.int64) + base).astype(np.int32)
