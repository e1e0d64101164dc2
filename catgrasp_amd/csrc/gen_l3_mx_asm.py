"""Generates l3_mx_asm.inc: the hand-scheduled gfx950 instruction stream of the 128 -> 1024 layer of pointmlp_split.hip in the
2-unit split mode "f16fp8x2", for ONE 256-point tile and one wave (its 4 channel blocks x 8 row tiles of 32 points).

Arithmetic of one 32x32 output block (K = 128), f32 accumulate throughout:
    x.w ~=  x_hi8.w_lo8 + x_lo8.w_hi8   2 x 2 v_mfma_scale_f32_32x32x64_f8f6f4  (e4m3 pieces, one power-of-two scale per 32 K values,
                                         applied by the instruction; 16 passes each -- half the cost per K of a 16-bit MFMA)
          + x_hi.w_hi                    8 x v_mfma_f32_32x32x16_f16               (IEEE-half pieces, 8 passes each)
= 128 MFMA passes per block instead of the 192 of the f16x3 split.

Operand layout of the scaled instruction (measured, scripts/probe/mx_probe2.hip): lane l holds 32 bytes of row l&31 (A) / column
l&31 (B); byte p of lane-half g = l>>5 multiplies byte p of lane-half g on the other side; the scale byte selected by op_sel from
the scale register of lane i + 32u covers bytes [16u, 16u+16) of BOTH lanes of row i.  So one MX unit = bytes [16u,16u+16) of the
lane pair: here unit u of k-half kh holds channel block 2*kh+u, byte 4q+i of lane-half g = channel 32*(2kh+u) + 8q + 4g + i -- exactly
the 16 + 16 values a lane pair of the producing 64 -> 128 layer holds, which therefore share one scale and are stored with one
ds_write_b128.

Registers are PHYSICAL (declared as clobbers): operand tuples of 8 registers cannot be sub-addressed through inline-asm operands.
    v[128:255]  accumulators c0..c7 (row tiles)          v[80:111]  A ring, 4 slots x 8 (an fp8 fragment or two f16 fragments)
    v[112:123]  f16 weight fragments, ring of 3          v[58:65], v[66:73]  fp8 weight fragments U, V
    v[76:79], v[124:127]  A scales, one per row tile     v74  weight scales (4 bytes: hi8 kh0, hi8 kh1, lo8 kh0, lo8 kh1)
Step = 64 cycles of the matrix pipe: one scaled MFMA (row tile rt) or two f16 MFMAs (row tiles 2P, 2P+1 at k chunk kc).  Per channel block:
    s  0.. 7  x_hi8[rt][kh0] . U = w_lo8[kh0]        s  8..15  x_hi8[rt][kh1] . V = w_lo8[kh1]       s 16..31  f16 kc 0..3
    s 32..39  x_lo8[rt][kh0] . U = w_hi8[kh0]        s 40..47  x_lo8[rt][kh1] . V = w_hi8[kh1]       s 48..63  f16 kc 4..7
Every A fill (2 ds_read_b128 [+ the ds_read_u16 of the row tile's scale pair, k-half 0 only]) is issued 3 steps ahead, every weight fragment >= 8 steps ahead (the
next block's first fragments during the current block's tail), all wait counts are exact.  The max over the 128 accumulator
registers of block q is interleaved, row tile by row tile, with the first 8 steps of block q+1 (which overwrite them in that order), and
folds straight into the wave's running per-lane maxima %[m0..3] (in/out operands; the lane^32 exchange happens once, after the last tile).
"""
import os

ACC0 = 128
RING0, NRING = 80, 4
M0, NM = 112, 3
U, V = 58, 66
SCR = [76, 77, 78, 79, 124, 125, 126, 127]      # A scale registers, one per row tile
BSC = 74
CLOBBER_LO, CLOBBER_HI = 58, 255

NB_BYTES = 16640                   # one channel block of the packed weights: f16 8192 | lo8 4096 | hi8 4096 | scales 256
OFF_LO8, OFF_HI8 = 8192, 12288
RT16 = 32 * 272                    # one 32-row tile of the f16 image (row stride 272 B)
RT8 = 32 * 144                     # one 32-row tile of an fp8 image (row stride 144 B: 128 data + 16 pad holding the scales)


def vr(a, n):
    return f'v[{a}:{a + n - 1}]' if n > 1 else f'v{a}'


def acc(rt):
    return vr(ACC0 + 16 * rt, 16)


class Gen:
    def __init__(self, nblocks=4):
        self.L = []
        self.ds = []          # issue order of LDS ops: tags
        self.vm = []          # issue order of global loads: tags
        self.nblocks = nblocks

    def emit(self, s):
        self.L.append(s)

    # ---- A side (LDS) ----
    def fill(self, q, s):
        """ds ops that fetch the A operand (and its scale) of step s of block q into ring slot s % NRING"""
        if q >= self.nblocks:
            return []
        slot = RING0 + 8 * (s % NRING)
        tag = ('a', q, s)
        out = []
        if s < 16 or 32 <= s < 48:
            lo = s >= 32
            t = s - 32 if lo else s
            kh, rt = divmod(t, 8)
            base = '%[a8l]' if lo else '%[a8h]'
            off = rt * RT8 + kh * 64
            out.append((f'ds_read_b128 {vr(slot, 4)}, {base} offset:{off}', tag))
            out.append((f'ds_read_b128 {vr(slot + 4, 4)}, {base} offset:{off + 16}', tag))
            if kh == 0:      # the row tile's scale pair (bytes: unit lhi of k-half 0, of k-half 1) serves both k-halves: steps s and s + 8
                out.append((f'ds_read_u16 v{SCR[rt]}, %[asc] offset:{rt * RT8 + (8 * 144 if lo else 0)}', tag))
        else:
            t = s - 16 if s < 32 else s - 48 + 16
            kc, P = divmod(t, 4)
            for j in range(2):
                out.append((f'ds_read_b128 {vr(slot + 4 * j, 4)}, %[a16] offset:{(2 * P + j) * RT16 + kc * 32}', tag))
        return out

    # ---- B side (global) ----
    def wload(self, q, what):
        """global loads of one weight item of block q, addressed relative to %[vw] (which stands at block `self.vw_at`)"""
        if q >= self.nblocks:
            return []
        rel = (q - self.vw_at) * NB_BYTES
        out = []
        if what[0] == 'm':                       # f16 fragment of k chunk kc -> ring slot (8q + kc) % NM
            kc = what[1]
            dst = M0 + 4 * ((8 * q + kc) % NM)
            out.append((f'v_add_u32 %[vo], {rel + kc * 1024}, %[vw]', None))
            out.append((f'global_load_dwordx4 {vr(dst, 4)}, %[vo], %[wb]', ('m', q, kc)))
        elif what[0] in ('lo8', 'hi8'):
            kh = what[1]
            dst = U if kh == 0 else V
            off = rel + (OFF_LO8 if what[0] == 'lo8' else OFF_HI8) + kh * 2048
            out.append((f'v_add_u32 %[vo], {off}, %[vw]', None))
            out.append((f'global_load_dwordx4 {vr(dst, 4)}, %[vo], %[wb]', (what[0], q, kh)))
            out.append((f'global_load_dwordx4 {vr(dst + 4, 4)}, %[vo], %[wb] offset:1024', (what[0], q, kh)))
        elif what[0] == 'sc':
            out.append((f'v_add_u32 %[vo], {(q - self.vw_at) * NB_BYTES}, %[vs]', None))
            out.append((f'global_load_dword v{BSC}, %[vo], %[wb]', ('sc', q)))
        return out

    def issue(self, items):
        abl = os.environ.get('MX_ABL', '')        # dev ablations: drop the LDS reads / the weight loads from the stream (results garbage)
        for ins, tag in items:
            if not ((abl == 'nods' and ins.startswith('ds_')) or (abl == 'novm' and ins.startswith('global_'))):
                self.emit(ins)
            if tag is not None:
                (self.ds if ins.startswith('ds_') else self.vm).append(tag)

    def wait(self, ds_tag=None, vm_tags=()):
        """s_waitcnt for: every LDS op tagged ds_tag landed, every global load with a tag in vm_tags landed"""
        parts = []
        if vm_tags:
            idx = max(i for i, t in enumerate(self.vm) if t in vm_tags)
            parts.append(f'vmcnt({min(len(self.vm) - 1 - idx, 63)})')
        if ds_tag is not None:
            idx = max(i for i, t in enumerate(self.ds) if t == ds_tag)
            parts.append(f'lgkmcnt({min(len(self.ds) - 1 - idx, 15)})')
        if parts:
            self.emit('s_waitcnt ' + ' '.join(parts))

    # ---- the max epilogue of one row tile of block q ----
    def epilogue(self, q, rt):
        m, t = f'%[m{q}]', '%[t0]'
        regs = [ACC0 + 16 * rt + r for r in range(16)]
        for dst, rr in ((m, regs[:8]), (t, regs[8:])):       # chain m continues the RUNNING max (an in/out operand), chain t is per block
            rr = list(rr)
            if rt == 0 and dst == t:
                self.emit(f'v_max3_f32 {dst}, v{rr[0]}, v{rr[1]}, v{rr[2]}')
                rr = rr[3:]
            while rr:
                if len(rr) >= 2:
                    self.emit(f'v_max3_f32 {dst}, {dst}, v{rr[0]}, v{rr[1]}')
                    rr = rr[2:]
                else:
                    self.emit(f'v_max_f32 {dst}, {dst}, v{rr[0]}')
                    rr = rr[1:]
        if rt == 7:
            self.emit(f'v_max_f32 {m}, {m}, {t}')

    def run(self):
        nb = self.nblocks
        self.vw_at = 0
        # prologue: the first block's weight fragments of the first phases, the A fills of steps 0..2
        for what in (('lo8', 0), ('lo8', 1), ('sc',), ('m', 0), ('m', 1)):
            self.issue(self.wload(0, what))
        # the workgroup barrier between the 64 -> 128 layer's LDS writes and these reads sits HERE, after the weight loads have been
        # issued: their L2 latency (the only wait of the stream that cannot be covered by earlier MFMAs) elapses while the wave waits for
        # its siblings.  lgkmcnt(0): this wave's own LDS writes have landed.
        self.emit('s_waitcnt lgkmcnt(0)')
        self.emit('s_barrier')
        for s in range(NRING - 1):
            self.issue(self.fill(0, s))
        for q in range(nb):
            for s in range(64):
                if q > 0 and s < 8:
                    self.epilogue(q - 1, s)
                # ---- what this step consumes
                need_vm = []
                if s == 0:
                    need_vm += [('lo8', q, 0), ('sc', q)]
                elif s == 8:
                    need_vm += [('lo8', q, 1)]
                elif s == 32:
                    need_vm += [('hi8', q, 0)]
                elif s == 40:
                    need_vm += [('hi8', q, 1)]
                main = 16 <= s < 32 or s >= 48
                if main:
                    t = s - 16 if s < 32 else s - 48 + 16
                    kc, P = divmod(t, 4)
                    if P == 0:
                        need_vm += [('m', q, kc)]
                self.wait(('a', q, s), need_vm)
                # ---- fillers of this step
                nq, ns = (q, s + NRING - 1) if s + NRING - 1 < 64 else (q + 1, s + NRING - 1 - 64)
                fillers = list(self.fill(nq, ns))
                w = []
                if s == 8:
                    w += self.wload(q, ('hi8', 0))
                if s == 16:
                    w += self.wload(q, ('hi8', 1))
                if main and P == 0:                        # f16 fragment two k chunks ahead (the next block's at kc 6, 7)
                    w += self.wload(q, ('m', kc + 2)) if kc + 2 < 8 else self.wload(q + 1, ('m', kc + 2 - 8))
                if s == 40:
                    w += self.wload(q + 1, ('lo8', 0))
                if s == 49:
                    w += self.wload(q + 1, ('lo8', 1))
                if s == 50:
                    w += self.wload(q + 1, ('sc',))          # the weight-scale register was last read by step 47
                # ---- MFMAs
                slot = RING0 + 8 * (s % NRING)
                if not main:
                    lo = s >= 32
                    t = s - 32 if lo else s
                    kh, rt = divmod(t, 8)
                    b = U if kh == 0 else V
                    osa = kh                                 # A scale register: bytes [unit lhi of kh0, of kh1]
                    osb = kh if lo else 2 + kh               # weight scale register: hi8 kh0, hi8 kh1, lo8 kh0, lo8 kh1
                    c_in = '0' if s < 8 else acc(rt)
                    self.emit(f'v_mfma_scale_f32_32x32x64_f8f6f4 {acc(rt)}, {vr(slot, 8)}, {vr(b, 8)}, {c_in}, v{SCR[rt]}, v{BSC} '
                              f'op_sel:[{osa & 1},{osb & 1},0] op_sel_hi:[{osa >> 1},{osb >> 1},0]')
                    self.issue(fillers + w)
                else:
                    m = M0 + 4 * ((8 * q + kc) % NM)
                    self.emit(f'v_mfma_f32_32x32x16_f16 {acc(2 * P)}, {vr(slot, 4)}, {vr(m, 4)}, {acc(2 * P)}')
                    self.issue(fillers[:1] + w)
                    self.emit(f'v_mfma_f32_32x32x16_f16 {acc(2 * P + 1)}, {vr(slot + 4, 4)}, {vr(m, 4)}, {acc(2 * P + 1)}')
                    self.issue(fillers[1:])
        # last block: plain epilogue.  c6 / c7 were written by the last two MFMAs: MFMA -> VALU read wait states first
        for rt in range(8):
            if rt in (0, 6):
                self.emit('s_nop 15')
            self.epilogue(nb - 1, rt)
        return self.L


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    g = Gen()
    lines = g.run()
    with open(os.path.join(here, 'l3_mx_asm.inc'), 'w') as f:
        f.write('// GENERATED by gen_l3_mx_asm.py -- do not edit.  The 128 -> 1024 layer of one 256-point tile for one wave, f16fp8x2 mode.\n')
        f.write('#define CG_L3_MX_ASM \\\n')
        for ln in lines:
            f.write(f'  "{ln}\\n" \\\n')
        f.write('  ""\n')
        f.write('#define CG_L3_MX_CLOBBERS ' + ', '.join(f'"v{i}"' for i in range(CLOBBER_LO, CLOBBER_HI + 1)) + '\n')
    n_mx = sum('mfma_scale' in x for x in lines)
    n_16 = sum('mfma_f32_32x32x16' in x for x in lines)
    print(len(lines), 'instructions,', n_mx, 'scaled MFMAs,', n_16, 'f16 MFMAs')


if __name__ == '__main__':
    main()
