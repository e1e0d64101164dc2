"""Generates l3_f32_asm.inc: the hand-scheduled gfx950 instruction stream of the 128 -> 1024 layer of pointmlp.hip (exact-f32 MFMA)
for ONE 64-point tile and one wave: 4 channel-block pairs x (16 k-steps x 16 v_mfma_f32_32x32x2_f32) + the max epilogues.

Why assembly: this stream is 91 % of the kernel's MFMAs.  hipcc schedules it "load, s_waitcnt, use" -- a weight fragment is waited
for 1..7 MFMAs after its global_load was issued (profiles/r2_pmc_sq_pointmlp_f32.csv: the matrix pipe idles 18 % of the kernel with
two waves per SIMD), and no source-level software pipeline survives its scheduler.  Here every operand fragment is requested one
full k-step (16 MFMAs = 1024 cycles) before it is used, so the only wait per k-step, `s_waitcnt vmcnt(0) lgkmcnt(0)`, never stalls.

Registers.  v_mfma_f32_32x32x2_f32 takes single-register A/B operands, which inline-asm tuple operands cannot name, so the stream
uses PHYSICAL accumulation registers, declared as clobbers: gfx950 lets global_load / ds_read write AGPRs and MFMAs read their A/B
operands from them.
    a[0:63]   accumulators c00 c01 c10 c11 (row tile x channel block)
    a[64:79]  operand buffer X: A fragments of row tiles 0/1 (a[64:67], a[68:71]), weight fragments of blocks 0/1 (a[72:75], a[76:79])
    a[80:95]  operand buffer Y (odd k-steps)
Stream per block pair:  for ks in 0..15: wait; MFMA; [bump offsets] 2 global_load + 2 ds_read for ks+1 into the other buffer as
fillers behind the first MFMAs (at ks = 15: the NEXT pair's ks = 0 fragments); 15 more MFMAs.  Then the epilogue: per-lane max of
the 32 accumulator registers of each channel block (v_accvgpr_read + v_max3_f32) into the named outputs; the lane^32 exchange,
bias, ReLU and the running max stay in C++.
"""
import os

ROW_TILE_BYTES = 32 * 132 * 4        # second row tile of the h2 image (row stride S128 = 132 floats)


def gen():
    L = []
    emit = L.append
    acc = {'c00': 'a[0:15]', 'c01': 'a[16:31]', 'c10': 'a[32:47]', 'c11': 'a[48:63]'}
    base = {'c00': 0, 'c01': 16, 'c10': 32, 'c11': 48}

    def buf(k):      # -> (a0, a1, b0, b1) first-register numbers
        o = 64 + 16 * (k % 2)
        return o, o + 4, o + 8, o + 12

    def loads(dst, ks, bump):
        """instructions that fetch the operands of k-step ks (of the current pair, or ks = 16 -> next pair's 0) into buffer dst"""
        a0, a1, b0, b1 = buf(dst)
        out = []
        if bump:
            out += [f'v_add_u32 %[vo0], {bump}, %[vo0]', f'v_add_u32 %[vo1], {bump}, %[vo1]']
        off = (ks % 4) * 1024
        out += [f'global_load_dwordx4 a[{b0}:{b0 + 3}], %[vo0], %[wb] offset:{off}',
                f'global_load_dwordx4 a[{b1}:{b1 + 3}], %[vo1], %[wb] offset:{off}',
                f'ds_read_b128 a[{a0}:{a0 + 3}], %[ar] offset:{(ks % 16) * 32}',
                f'ds_read_b128 a[{a1}:{a1 + 3}], %[ar] offset:{(ks % 16) * 32 + ROW_TILE_BYTES}']
        return out

    emit('v_mov_b32 %[vo0], %[voff]')
    emit('v_add_u32 %[vo1], 0x4000, %[voff]')
    for ins in loads(0, 0, 0):
        emit(ins)
    for p in range(4):
        for ks in range(16):
            a0, a1, b0, b1 = buf(ks)
            emit('s_waitcnt vmcnt(0) lgkmcnt(0)')
            fill = []
            if ks < 15:
                fill = loads(ks + 1, ks + 1, 0x1000 if (ks + 1) % 4 == 0 else 0)
            elif p < 3:
                fill = loads(0, 16, 0x5000)         # next pair: its base is 32768 bytes on, the running offsets stand at +12288
            seq = []
            for j in range(4):
                seq += [('c00', a0 + j, b0 + j), ('c01', a0 + j, b1 + j), ('c10', a1 + j, b0 + j), ('c11', a1 + j, b1 + j)]
            for i, (c, ra, rb) in enumerate(seq):
                src_c = '0' if (ks == 0 and i < 4) else acc[c]
                emit(f'v_mfma_f32_32x32x2_f32 {acc[c]}, a{ra}, a{rb}, {src_c}')
                # fillers one per MFMA gap, none between MFMAs of the same accumulator (consecutive MFMAs always differ here)
                if fill:
                    emit(fill.pop(0))
        # epilogue: MFMA results -> VALU reads need the pipeline drained (16-pass MFMA: 18 wait states)
        emit('s_nop 15')
        emit('s_nop 7')
        for half, (ca, cb) in enumerate((('c00', 'c10'), ('c01', 'c11'))):
            m = f'%[m{p}{half}]'
            regs = [base[ca] + r for r in range(16)] + [base[cb] + r for r in range(16)]
            emit(f'v_accvgpr_read_b32 {m}, a{regs[0]}')
            rest = regs[1:]
            while rest:
                if len(rest) >= 2:
                    emit(f'v_accvgpr_read_b32 %[t0], a{rest[0]}')
                    emit(f'v_accvgpr_read_b32 %[t1], a{rest[1]}')
                    emit(f'v_max3_f32 {m}, {m}, %[t0], %[t1]')
                    rest = rest[2:]
                else:
                    emit(f'v_accvgpr_read_b32 %[t0], a{rest[0]}')
                    emit(f'v_max_f32 {m}, {m}, %[t0]')
                    rest = rest[1:]
    return L


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    lines = gen()
    with open(os.path.join(here, 'l3_f32_asm.inc'), 'w') as f:
        f.write('// GENERATED by gen_l3_f32_asm.py -- do not edit.  The 128 -> 1024 layer of one 64-point tile for one wave (see the generator).\n')
        f.write('#define CG_L3_F32_ASM \\\n')
        for ln in lines:
            f.write(f'  "{ln}\\n" \\\n')
        f.write('  ""\n')
        f.write('#define CG_L3_F32_CLOBBERS ' + ', '.join(f'"a{i}"' for i in range(96)) + '\n')
    print(len(lines), 'instructions,', sum('mfma' in x for x in lines), 'MFMAs')


if __name__ == '__main__':
    main()
