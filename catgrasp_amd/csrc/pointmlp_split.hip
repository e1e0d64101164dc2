// Split-precision variant of the fused per-point MLP chain + max-pool (see pointmlp.hip for the op sequence it
// replaces: pointnet2.py:172-176, :210-214, :243-266).
//
// Every contraction is evaluated as three 16-bit MFMAs with f32 accumulation:
//     x = x_hi + x_lo,  w = w_hi + w_lo  (16-bit pieces, round-to-nearest-even; lo = round(x - x_hi))
//     x.w ~= x_lo.w_hi + x_hi.w_lo + x_hi.w_hi          (the dropped term x_lo.w_lo is below the pieces' resolution)
// on v_mfma_f32_32x32x16_{f16,bf16} (16x the f32-MFMA rate, so 16/3 = 5.3x per contraction).  The kernel is instantiated for
// both element types: "f16x3" (IEEE-half pieces, 11 + 11 significant bits; logits within ~2e-6 of the float64 evaluation, which
// is float32's own distance; activations must stay below 65504) -- the engine's default -- and "bf16x3" (8 + 8 bits, ~2e-5, no
// range limit).  Both are inside the 1e-4 parity bar (tests/test_pointnet_gpu.py runs every case under both).
//
// Layout: one workgroup = 8 waves owns one sample (or a slice of its point tiles); a tile is 256 points; 160 KB LDS,
// one workgroup per CU.
//  * FRONT LAYERS (6->64 -> [64->64] -> 64->128) ARE WAVE-PRIVATE AND REGISTER-RESIDENT.  Wave w carries points
//    [32w, 32w+32) of the tile through the chain TRANSPOSED: out^T = W . in^T, i.e. the weights are the MFMA A operand
//    and the activations the B operand.  In the 32x32 accumulator layout a lane then holds ONE point (column) and 16
//    channels (rows 8q + 4*(lane>>5) + 0..3), and the next layer's B fragment (8 consecutive channels of that point)
//    is assembled from the lane's own registers plus one v_permlane32_swap with its partner lane (lane^32) per dword:
//    no LDS round trip and no barrier between layers.  The first layer's K (6 inputs + a constant-1 bias row, padded to
//    16) rides the same MFMA.
//  * The 128-wide activation is written to LDS already split into 16-bit hi / lo images ([256][136] each; row stride
//    272 B makes the ds_read_b128 fragment reads of the next layer conflict-free), 8 bytes per store.
//  * In the 128->1024 layer (points = A operand again) wave w owns channel blocks [4w,4w+4) and ALL 8 row tiles, so
//    each packed weight fragment is fetched from L2 exactly once per workgroup tile (10.7 B/clk/CU at full MFMA rate),
//    and the max over points is a per-lane reduction over accumulator registers + one lane^32 swap.
//  * The operand fragments of the first layer and of the 64->64 layer (shared weights, or the per-sample feature
//    transform split on the fly) are staged once per workgroup in the remaining 20 KB of LDS.
#include "cg_split.hpp"
#include "../../include/catgrasp_amd.h"
#include "l3_asm.inc"
#include "l3_mx_asm.inc"

namespace {

constexpr int SH = 136;        // 16-bit elements per row of the h2 hi / lo images
constexpr int S8 = 144;        // bytes per row of the fp8 images of the f16fp8x2 mode: 128 data + 16 pad (the pads of a row tile hold its scale bytes)
constexpr int MX_NB_BYTES = 16640;   // one packed channel block of the f16fp8x2 weights (folding.pack_b_f16fp8x2; gen_l3_mx_asm.py)
template <int RT, bool MX = false> struct Geo {
  static constexpr int TP = 32 * RT;        // points per tile, one wave per 32-point row tile
  static constexpr int NT = 64 * RT;
  static constexpr int NBW = 32 / RT;       // 32-channel blocks of the 1024-wide layer owned by each wave
  // h2 hi/lo images + running max + first-layer fragments [2][2][64] + mid-layer fragments [2][4][2][64] (16 B each)
  // MX: h2 hi image (f16) + hi8 / lo8 images (e4m3); the running max lives in registers
  static constexpr size_t IMG_BYTES = MX ? (size_t)TP * SH * 2 + (size_t)2 * TP * S8 : (size_t)2 * TP * SH * 2 + 1024 * 4;
  static constexpr size_t LDS_BYTES = IMG_BYTES + 256 * 16 + 1024 * 16;
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

struct ArgsB {
  const float* x; int B; int N;
  const float* t3;
  const float* w1; const float* b1;
  const unsigned short* wm; const float* bm;     // split-packed 64->64 (MID==1)
  const float* t64;                              // (B,64,64) f32, TRANSPOSED: t64[b][n][k] = T_b[k][n] (MID==2)
  const unsigned short* w2; const float* b2;     // split-packed 64->128
  const unsigned short* w3; const float* b3;     // split-packed 128->1024
  int relu3; int nsplit;
  int n_main;            // samples [0, n_main) use `nsplit` workgroups each, samples [n_main, B) `tail_split` (tail balancing)
  int tail_split;
  float* out; float* pointfeat;
  int* status;           // optional device int (F16 only): |= CG_HALF_OVERFLOW / CG_HALF_UNDERFLOW, see cg_split.hpp
};

// split product block: c += A.B with A = ah + al, B = bh + bl (small terms first)
template <bool F16>
__device__ __forceinline__ f32x16 mfma3(frag ah, frag al, frag bh, frag bl, f32x16 c) {
  c = mfma_x<F16>(ah, bl, c);
  c = mfma_x<F16>(al, bh, c);
  return mfma_x<F16>(ah, bh, c);
}

template <bool F16>
__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, frag& hi, frag& lo, float& amax) {
  unsigned h[4], l[4];
  split2<F16>(v0[0], v0[1], h[0], l[0], amax); split2<F16>(v0[2], v0[3], h[1], l[1], amax);
  split2<F16>(v1[0], v1[1], h[2], l[2], amax); split2<F16>(v1[2], v1[3], h[3], l[3], amax);
  hi = frag{h[0], h[1], h[2], h[3]}; lo = frag{l[0], l[1], l[2], l[3]};
}

// One 32-channel x 32-point accumulator tile (lane = point l&31; register r = channel 8*(r>>2) + 4*(l>>5) + (r&3)) ->
// the two 16-deep B fragments (hi and lo images) the next layer consumes: lane (p, h) needs channels 16*kc + 8*h + 0..7.
// Quads (0,1) feed kc = 0 and (2,3) feed kc = 1; v_permlane32_swap exchanges the upper half of the even quad with the
// lower half of the odd quad, which lands exactly the partner lane's four channels next to the lane's own four.
template <bool F16>
__device__ __forceinline__ void acts_to_frags(const f32x16& c, frag& h0, frag& l0, frag& h1, frag& l1, float& amax) {
  unsigned H[4][2], L[4][2];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    split2<F16>(c[4 * q], c[4 * q + 1], H[q][0], L[q][0], amax);
    split2<F16>(c[4 * q + 2], c[4 * q + 3], H[q][1], L[q][1], amax);
  }
#pragma unroll
  for (int q = 0; q < 4; q += 2) {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      u32x2 r = __builtin_amdgcn_permlane32_swap(H[q][d], H[q + 1][d], false, false);
      H[q][d] = r[0]; H[q + 1][d] = r[1];
      r = __builtin_amdgcn_permlane32_swap(L[q][d], L[q + 1][d], false, false);
      L[q][d] = r[0]; L[q + 1][d] = r[1];
    }
  }
  h0 = frag{H[0][0], H[0][1], H[1][0], H[1][1]};
  l0 = frag{L[0][0], L[0][1], L[1][0], L[1][1]};
  h1 = frag{H[2][0], H[2][1], H[3][0], H[3][1]};
  l1 = frag{L[2][0], L[2][1], L[3][0], L[3][1]};
}

// accumulator tile initialised with the per-channel bias of channel block nb (channels = accumulator rows)
__device__ __forceinline__ f32x16 bias_tile(const float* bias, int nb, int lhi) {
  f32x16 c;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 v = *(const f32x4*)(bias + nb * 32 + 8 * q + 4 * lhi);
    c[4 * q] = v[0]; c[4 * q + 1] = v[1]; c[4 * q + 2] = v[2]; c[4 * q + 3] = v[3];
  }
  return c;
}

__device__ __forceinline__ f32x16 relu16(f32x16 c) {
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = fmaxf(c[r], 0.f);
  return c;
}

// packed split weights: Wp[nb][kc][2 (hi,lo)][lane][8] bf16
__device__ __forceinline__ void load_b(const unsigned short* wp, int nb, int kc, int nkc, int lane, frag& bhi, frag& blo) {
  const frag* p = (const frag*)wp + ((size_t)(nb * nkc + kc) * 2) * 64 + lane;
  bhi = p[0]; blo = p[64];
}

// An opaque zero: added to the (tile-invariant) front-layer weight pointers inside the tile loop so the compiler
// does not hoist 32 KB of weight-fragment loads out of the loop and spill them.
__device__ __forceinline__ int opaque_zero() { int z; asm volatile("s_mov_b32 %0, 0" : "=s"(z)); return z; }
__device__ __forceinline__ int opaque_copy(int v) { int r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(v)); return r; }

template <int MID, int RT, bool F16, bool MX>
__global__ __launch_bounds__(64 * RT, 2) void pointmlp_max_split_kernel(ArgsB a) {
  static_assert(!MX || (F16 && RT == 8), "the f16fp8x2 stream is written for half pieces and 256-point tiles");
  constexpr int TP = Geo<RT>::TP, NT = Geo<RT>::NT, NBW = Geo<RT>::NBW;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  unsigned short* h2hi = (unsigned short*)smem_raw;      // 16-bit elements (bf16 or half bit patterns)
  unsigned short* h2lo = h2hi + TP * SH;                 // (!MX)
  unsigned char* h8hi = (unsigned char*)(h2hi + TP * SH);   // (MX) e4m3 images of the hi / lo pieces, one power-of-two scale per
  unsigned char* h8lo = h8hi + TP * S8;                     //      32 channels of a point; the row's scale bytes sit in its pad
  float* rmax = (float*)(h2lo + TP * SH);                // (!MX)
  frag* w1f = (frag*)(smem_raw + Geo<RT, MX>::IMG_BYTES);     // [nb 2][hi|lo][lane]
  frag* wmf = w1f + 256;                  // [nb 2][kc 4][hi|lo][lane]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31;
  int b, split, nsp;
  if ((int)blockIdx.x < a.n_main * a.nsplit) {
    nsp = a.nsplit; b = blockIdx.x / nsp; split = blockIdx.x - b * nsp;
  } else {                 // the last B % #CU samples of a big batch: one workgroup per tile, so the final round is short
    const int r = blockIdx.x - a.n_main * a.nsplit;
    nsp = a.tail_split; b = a.n_main + r / nsp; split = r - (r / nsp) * nsp;
  }
  const int ntiles = (a.N + TP - 1) / TP;
  const int t_begin = (int)(((long)ntiles * split) / nsp);
  const int t_end = (int)(((long)ntiles * (split + 1)) / nsp);

  float amax = 0.f;        // largest magnitude this lane handed to the split since the last fold (half range check, F16 only)
  int flags = 0;           // wave-uniform CG_HALF_* bits; amax is folded into it after every layer, so no VGPR lives across L3
  auto fold = [&](bool check_low) {
    if constexpr (F16) {
      if (__builtin_amdgcn_ballot_w64(!(amax < HALF_MAX)) != 0) flags |= CG_HALF_OVERFLOW;
      // a layer output whose every value in this wave's 32-point tile is below HALF_LOW sits in / near the half subnormals:
      // the lo pieces then carry an absolute, not a relative, error (2^-25) -- report it so the caller can re-run in bf16
      if (check_low && __builtin_amdgcn_ballot_w64(amax >= HALF_LOW) == 0) flags |= CG_HALF_UNDERFLOW;
      amax = 0.f;
    }
  };
  // ---- once per workgroup: running max, first-layer fragments (W1 | b1 as the k = 6 column), mid-layer fragments
  if constexpr (!MX) { for (int i = tid; i < 1024; i += NT) rmax[i] = -INFINITY; }
  float rm[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};      // (MX) running max of the wave's 4 channel blocks, channel = lane & 31
  for (int i = tid; i < 128; i += NT) {
    const int nb = i >> 6, ln = i & 63, row = nb * 32 + (ln & 31);
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
    if ((ln >> 5) == 0) {
      const float* wr = a.w1 + row * 6;
      v0 = f32x4{wr[0], wr[1], wr[2], wr[3]};
      v1 = f32x4{wr[4], wr[5], a.b1[row], 0.f};
    }
    frag hi, lo;
    split8<F16>(v0, v1, hi, lo, amax);
    w1f[(nb * 2) * 64 + ln] = hi;
    w1f[(nb * 2 + 1) * 64 + ln] = lo;
  }
  if (MID == 1) {
    for (int i = tid; i < 1024; i += NT) wmf[i] = ((const frag*)a.wm)[i];
  }
  if (MID == 2) {   // t64 is stored TRANSPOSED (Tt[n][k] = T[k][n]): a lane's 8 consecutive k are two 16-byte loads
    for (int i = tid; i < 512; i += NT) {
      const int ln = i & 63, kc = (i >> 6) & 3, nb = i >> 8;
      const float* tp = a.t64 + (size_t)b * 4096 + (nb * 32 + (ln & 31)) * 64 + kc * 16 + (ln >> 5) * 8;
      frag hi, lo;
      split8<F16>(*(const f32x4*)tp, *(const f32x4*)(tp + 4), hi, lo, amax);
      wmf[((nb * 4 + kc) * 2) * 64 + ln] = hi;
      wmf[((nb * 4 + kc) * 2 + 1) * 64 + ln] = lo;
    }
  }
  fold(false);             // staged first-layer weights / feature transform: range only
  float t3r[9];
  if (a.t3) {
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      t3r[j] = a.t3[b * 9 + j];
      t3r[j] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(t3r[j])));     // scalar registers: nothing of it lives across the L3 asm block
    }
  }
  const float* xb = a.x + (size_t)b * a.N * 6;
  u32x4 wxh, wxl;          // weight fragments (hi / lo) of the wave's next channel block, k chunk 0
  if constexpr (!MX) {
    const u32x4* p = (const u32x4*)a.w3 + (size_t)((w * NBW * 8) * 2) * 64 + lane;
    wxh = p[0]; wxl = p[64];
  }
  __syncthreads();
  // the lane's point of the NEXT tile travels from HBM while the current tile's 128 -> 1024 stream runs (6 registers across it)
  f32x2 xn0 = {0.f, 0.f}, xn1 = xn0, xn2 = xn0;
  auto fetch_point = [&](int tile, int l31) {
    if (tile < t_end) {
      const int q = tile * TP + w * 32 + l31;
      const f32x2* src = (const f32x2*)(xb + (size_t)(q < a.N ? q : a.N - 1) * 6);       // replicate the last point: max-pool is idempotent
      xn0 = src[0]; xn1 = src[1]; xn2 = src[2];
    }
  };
  fetch_point(t_begin, l31);

  for (int tile = t_begin; tile < t_end; ++tile) {
    // every lane-dependent index / LDS address of the tile body is re-derived from an opaque copy of the lane id, so that the id alone -- not
    // the two dozen values derived from it -- is live across the L3 asm block (MX: it leaves the compiler only v0..v57)
    const int ln = opaque_copy(lane);
    const int r31 = ln & 31, hh = ln >> 5;
    // LDS byte addresses of this lane's A-fragment row in the hi / lo images (generic -> LDS address = low 32 bits)
    const unsigned ahi_t = (unsigned)(uintptr_t)(h2hi + r31 * SH + hh * 8), alo_t = (unsigned)(uintptr_t)(h2lo + r31 * SH + hh * 8);
    const unsigned a8h_t = (unsigned)(uintptr_t)(h8hi + r31 * S8 + hh * 32), a8l_t = (unsigned)(uintptr_t)(h8lo + r31 * S8 + hh * 32);
    // scale pairs of a 32-row tile: row r's hi-piece dword sits in the pad of row r>>2, slot r&3 (its lo-piece dword 8 rows further), so the
    // 32 rows of a tile fall into 32 different banks (in the row's own pad -- stride 36 dwords -- they fell 4-way into 8)
    const unsigned asc_t = (unsigned)(uintptr_t)(h8hi + (r31 >> 2) * S8 + 128 + (r31 & 3) * 4 + hh * 2);
    // ================= front layers, wave-private and register-resident: points [32w, 32w+32) =================
    const int oz = opaque_zero();
    const unsigned short* w2_t = a.w2 + oz;
    const float* b2_t = a.b2 + oz;
    const int pt = tile * TP + w * 32 + r31;
    frag fh[4], fl[4];          // the activation as B fragments (hi / lo), 4 chunks of 16 channels
    {
      const f32x2 v0 = xn0, v1 = xn1, v2 = xn2;
      float px = v0[0], py = v0[1], pz = v1[0];
      if (a.t3) {
        const float qx = px * t3r[0] + py * t3r[3] + pz * t3r[6];
        const float qy = px * t3r[1] + py * t3r[4] + pz * t3r[7];
        const float qz = px * t3r[2] + py * t3r[5] + pz * t3r[8];
        px = qx; py = qy; pz = qz;
      }
      f32x4 q0 = {px, py, pz, v1[1]}, q1 = {v2[0], v2[1], 1.f, 0.f};     // k = 6 carries the bias
      if (hh) { q0 = f32x4{0.f, 0.f, 0.f, 0.f}; q1 = q0; }              // k = 8..15: padding
      frag xh, xl;
      split8<F16>(q0, q1, xh, xl, amax);
      fold(false);
      // L0: 6(+1) -> 64
      const f32x16 z = {0};
      f32x16 c0 = mfma3<F16>(w1f[ln], w1f[64 + ln], xh, xl, z);
      f32x16 c1 = mfma3<F16>(w1f[128 + ln], w1f[192 + ln], xh, xl, z);
      acts_to_frags<F16>(relu16(c0), fh[0], fl[0], fh[1], fl[1], amax);
      acts_to_frags<F16>(relu16(c1), fh[2], fl[2], fh[3], fl[3], amax);
      fold(true);
    }
    if (MID != 0) {  // mid: 64 -> 64 (shared conv+BN+ReLU, or the per-sample 64x64 feature transform)
      f32x16 c0, c1;
      if (MID == 1) { c0 = bias_tile(a.bm + oz, 0, hh); c1 = bias_tile(a.bm + oz, 1, hh); }
      else { c0 = f32x16{0}; c1 = f32x16{0}; }
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        const frag* f0 = wmf + (kc * 2) * 64 + ln;
        const frag* f1 = wmf + ((4 + kc) * 2) * 64 + ln;
        const frag a0h = f0[0], a0l = f0[64], a1h = f1[0], a1l = f1[64];
        c0 = mfma_x<F16>(a0h, fl[kc], c0); c1 = mfma_x<F16>(a1h, fl[kc], c1);
        c0 = mfma_x<F16>(a0l, fh[kc], c0); c1 = mfma_x<F16>(a1l, fh[kc], c1);
        c0 = mfma_x<F16>(a0h, fh[kc], c0); c1 = mfma_x<F16>(a1h, fh[kc], c1);
      }
      if (MID == 1) { c0 = relu16(c0); c1 = relu16(c1); }
      if (MID == 2 && a.pointfeat && pt < a.N) {
        float* pf = a.pointfeat + ((size_t)b * a.N + pt) * 64 + 4 * hh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          *(f32x4*)(pf + 8 * q) = f32x4{c0[4 * q], c0[4 * q + 1], c0[4 * q + 2], c0[4 * q + 3]};
          *(f32x4*)(pf + 32 + 8 * q) = f32x4{c1[4 * q], c1[4 * q + 1], c1[4 * q + 2], c1[4 * q + 3]};
        }
      }
      acts_to_frags<F16>(c0, fh[0], fl[0], fh[1], fl[1], amax);
      acts_to_frags<F16>(c1, fh[2], fl[2], fh[3], fl[3], amax);
      fold(true);
    }
    __syncthreads();   // the previous tile's L3 reads of the h2 images are complete
    {  // L2: 64 -> 128, two channel blocks at a time, written split into the hi / lo images
      const int row = w * 32 + r31;
      unsigned sc_word_hi = 0, sc_word_lo = 0;
#pragma unroll
      for (int np = 0; np < 2; ++np) {
        f32x16 c0 = bias_tile(b2_t, np * 2, hh), c1 = bias_tile(b2_t, np * 2 + 1, hh);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          frag a0h, a0l, a1h, a1l;
          load_b(w2_t, np * 2, kc, 4, ln, a0h, a0l);
          load_b(w2_t, np * 2 + 1, kc, 4, ln, a1h, a1l);
          c0 = mfma_x<F16>(a0h, fl[kc], c0); c1 = mfma_x<F16>(a1h, fl[kc], c1);
          c0 = mfma_x<F16>(a0l, fh[kc], c0); c1 = mfma_x<F16>(a1l, fh[kc], c1);
          c0 = mfma_x<F16>(a0h, fh[kc], c0); c1 = mfma_x<F16>(a1h, fh[kc], c1);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const f32x16 c = relu16(h ? c1 : c0);
          if constexpr (!MX) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              unsigned h0, l0, h1, l1;
              split2<F16>(c[4 * q], c[4 * q + 1], h0, l0, amax);
              split2<F16>(c[4 * q + 2], c[4 * q + 3], h1, l1, amax);
              const int off = row * SH + (np * 2 + h) * 32 + 8 * q + 4 * hh;
              *(u32x2*)(h2hi + off) = u32x2{h0, h1};
              *(u32x2*)(h2lo + off) = u32x2{l0, l1};
            }
          } else {
            // channel block np*2+h of this point = one MX unit: the ln pair (p, 0) / (p, 1) holds its 16 + 16 values.  Scale: the
            // unit's largest value lands in [128, 256) of e4m3 (max 448); its residuals (<= 2^-12 of that) in (0, 128].
            float m = fmaxf(max16(c), 0.f);
            const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
            m = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            amax = fmaxf(amax, m);
            int e = __builtin_amdgcn_frexp_expf(m);           // m = f * 2^e, f in [0.5, 1); 0 for m = 0
            e = e < -100 ? -100 : (e > 100 ? 100 : e);
            const float sc_hi = __uint_as_float((unsigned)(e - 8 + 127) << 23), sc_lo = __uint_as_float((unsigned)(e - 19 + 127) << 23);
            u32x4 d_hi, d_lo;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              unsigned h0, h1, dl;
              d_hi[q] = split_mx(c[4 * q], c[4 * q + 1], c[4 * q + 2], c[4 * q + 3], sc_hi, sc_lo, h0, h1, dl);
              d_lo[q] = dl;
              *(u32x2*)(h2hi + row * SH + (np * 2 + h) * 32 + 8 * q + 4 * hh) = u32x2{h0, h1};
            }
            // unit u = h of k-half kh = np: bytes [16u, 16u+16) of the 32 the reading ln-half hh fetches for that k-half
            *(u32x4*)(h8hi + row * S8 + np * 64 + hh * 32 + h * 16) = d_hi;
            *(u32x4*)(h8lo + row * S8 + np * 64 + hh * 32 + h * 16) = d_lo;
            sc_word_hi |= (unsigned)(119 + e) << (8 * (2 * h + np));      // E8M0 of 2^(e-8); reader ln-half u reads bytes [2u, 2u+1] = k-halves 0, 1
            sc_word_lo |= (unsigned)(108 + e) << (8 * (2 * h + np));      // E8M0 of 2^(e-19)
          }
        }
      }
      if constexpr (MX) {
        if (hh == 0) {
          unsigned char* sp = h8hi + (w * 32 + (r31 >> 2)) * S8 + 128 + (r31 & 3) * 4;
          *(unsigned*)sp = sc_word_hi;
          *(unsigned*)(sp + 8 * S8) = sc_word_lo;
        }
      }
    }
    fold(true);
    fetch_point(tile + 1, r31);
    if constexpr (!MX) __syncthreads();        // MX: the barrier is inside the asm block, behind its first weight loads
    // ================= L3: 128 -> 1024 + running max.  wave w owns channel blocks [4w, 4w+4) =================
    // The 192-MFMA stream of one channel block is hand-scheduled assembly (gen_l3_asm.py -> l3_asm.inc): exact wait
    // counts, A fragments through a 3-deep register ring, weight fragments double buffered with the next block's first
    // fragments (wxh / wxl) already in flight when the block ends.
    if constexpr (MX) {
      // one asm block = the wave's 4 channel blocks of this tile (gen_l3_mx_asm.py -> l3_mx_asm.inc); it folds the tile into the running per-lane maxima rm[0..3]
      float t0; unsigned vo;
      const unsigned vw = (unsigned)(w * NBW * MX_NB_BYTES + ln * 16), vs = (unsigned)(w * NBW * MX_NB_BYTES + 16384 + ln * 4);
      asm volatile(CG_L3_MX_ASM
                   : [m0] "+v"(rm[0]), [m1] "+v"(rm[1]), [m2] "+v"(rm[2]), [m3] "+v"(rm[3]), [t0] "=&v"(t0), [vo] "=&v"(vo)
                   : [a16] "v"(ahi_t), [a8h] "v"(a8h_t), [a8l] "v"(a8l_t), [asc] "v"(asc_t), [vw] "v"(vw), [vs] "v"(vs), [wb] "s"(a.w3)
                   : "memory", CG_L3_MX_CLOBBERS);
    } else {
#pragma unroll 1
    for (int q = 0; q < NBW; ++q) {
      const int nb = w * NBW + q;
      const int nb_next = (q + 1 < NBW) ? nb + 1 : w * NBW;
      unsigned voff = (unsigned)((nb * 8 * 2) * 64 + ln) * 16u;
      const unsigned vnext = (unsigned)((nb_next * 8 * 2) * 64 + ln) * 16u;
      f32x16 c0, c1, c2, c3, c4, c5, c6, c7;
      u32x4 r0ah, r0al, r0bh, r0bl, r1ah, r1al, r1bh, r1bl, r2ah, r2al, r2bh, r2bl, wyh, wyl;
#define CG_L3_OPERANDS \
                   : [c0] "=&v"(c0), [c1] "=&v"(c1), [c2] "=&v"(c2), [c3] "=&v"(c3), [c4] "=&v"(c4), [c5] "=&v"(c5),\
                     [c6] "=&v"(c6), [c7] "=&v"(c7), [r0ah] "=&v"(r0ah), [r0al] "=&v"(r0al), [r0bh] "=&v"(r0bh),\
                     [r0bl] "=&v"(r0bl), [r1ah] "=&v"(r1ah), [r1al] "=&v"(r1al), [r1bh] "=&v"(r1bh), [r1bl] "=&v"(r1bl),\
                     [r2ah] "=&v"(r2ah), [r2al] "=&v"(r2al), [r2bh] "=&v"(r2bh), [r2bl] "=&v"(r2bl), [yh] "=&v"(wyh),\
                     [yl] "=&v"(wyl), [xh] "+v"(wxh), [xl] "+v"(wxl), [voff] "+v"(voff)\
                   : [vnext] "v"(vnext), [ahi] "v"(ahi_t), [alo] "v"(alo_t), [wbase] "s"(a.w3) \
                   : "memory"
      if constexpr (F16) { asm volatile(CG_L3_BLOCK_ASM_F16 CG_L3_OPERANDS); }
      else { asm volatile(CG_L3_BLOCK_ASM_BF16 CG_L3_OPERANDS); }
#undef CG_L3_OPERANDS
      float m = fmaxf(fmaxf(max16(c0), max16(c1)), max16(c2));
      m = fmaxf(fmaxf(m, max16(c3)), max16(c4));
      m = fmaxf(fmaxf(m, max16(c5)), max16(c6));
      m = fmaxf(m, max16(c7));
      m = fmaxf(m, __shfl_xor(m, 32));
      if (ln < 32) {
        const int ch = nb * 32 + ln;
        rmax[ch] = fmaxf(rmax[ch], m);
      }
    }
    }
  }
  if (F16 && flags && lane == 0 && a.status) atomicOr(a.status, flags);
  __syncthreads();
  if (t_end > t_begin) {
    if constexpr (MX) {
#pragma unroll
      for (int q = 0; q < 4; ++q) rm[q] = fmaxf(rm[q], __shfl_xor(rm[q], 32));
      if (lane < 32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ch = (w * NBW + q) * 32 + lane;
          float v = rm[q] + a.b3[ch];
          if (a.relu3) v = fmaxf(v, 0.f);
          if (nsp == 1) a.out[(size_t)b * 1024 + ch] = v;
          else atomic_max_f32(a.out + (size_t)b * 1024 + ch, v);
        }
      }
    } else {
      for (int ch = tid; ch < 1024; ch += NT) {
        float v = rmax[ch] + a.b3[ch];
        if (a.relu3) v = fmaxf(v, 0.f);
        if (nsp == 1) a.out[(size_t)b * 1024 + ch] = v;
        else atomic_max_f32(a.out + (size_t)b * 1024 + ch, v);
      }
    }
  }
}

__global__ void fill_kernel_b(float* p, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

template <int MID, int RT, bool F16, bool MX = false>
int launch(const ArgsB& a, hipStream_t s, int dev) {
  constexpr int NT = Geo<RT>::NT;
  constexpr size_t LDS_BYTES = Geo<RT, MX>::LDS_BYTES;
  auto kern = pointmlp_max_split_kernel<MID, RT, F16, MX>;
  static bool attr_set[CG_MAX_DEVICES] = {};     // per instantiation and per device (the attribute is per device)
  if (dev < 0 || dev >= CG_MAX_DEVICES) return CG_ERR_UNSUPPORTED;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.n_main * a.nsplit + (a.B - a.n_main) * a.tail_split)), dim3(NT), LDS_BYTES, s, a);
  return cg_hip_status(hipGetLastError());
}

}  // namespace

template <bool F16, bool MX = false>
static int pointmlp_max_split(const float* x, int B, int N, const float* t3, const float* w1, const float* b1,
                              int mid_mode, const unsigned short* wm_split, const float* bm, const float* t64,
                              const unsigned short* w2_split, const float* b2, const unsigned short* w3_split,
                              const float* b3, int relu3, int nsplit, int tile_points, float* out, float* pointfeat,
                              int* status, void* stream) {
  if (B < 0 || N <= 0 || mid_mode < 0 || mid_mode > 2) return CG_ERR_ARG;
  if (tile_points != 256) return CG_ERR_UNSUPPORTED;      // one geometry: 256-point tiles, 8 waves, one workgroup per CU
  if (B == 0) return CG_OK;
  if (!x || !w1 || !b1 || !w2_split || !b2 || !w3_split || !b3 || !out) return CG_ERR_ARG;
  if (mid_mode == 1 && (!wm_split || !bm)) return CG_ERR_ARG;
  if (mid_mode == 2 && !t64) return CG_ERR_ARG;
  if (pointfeat && mid_mode != 2) return CG_ERR_ARG;
  const int ntiles = (N + tile_points - 1) / tile_points;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > ntiles) nsplit = ntiles;
  hipStream_t s = (hipStream_t)stream;
  // Tail balancing: with one workgroup per sample (nsplit == 1) and B >= #CU, the last B % #CU samples would occupy a few CUs
  // for a whole sample's duration while the rest of the chip idles; they are split one workgroup per tile instead (atomic max
  // into a -inf pre-filled row), so the final round lasts one tile, not ntiles.
  int n_main = B, tail_split = 1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return CG_ERR_UNSUPPORTED;
  if (nsplit == 1 && ntiles > 1) {
    const int n_cu = cg_device_cu_count(dev);
    if (n_cu <= 0) return CG_ERR_UNSUPPORTED;
    if (B >= n_cu && (B % n_cu) != 0) { n_main = B - B % n_cu; tail_split = ntiles; }
  }
  if (nsplit > 1 || tail_split > 1) {
    const int first = (nsplit > 1) ? 0 : n_main;
    const size_t n = (size_t)(B - first) * 1024;
    hipLaunchKernelGGL(fill_kernel_b, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out + (size_t)first * 1024, n, -INFINITY);
  }
  ArgsB a{x, B, N, t3, w1, b1, wm_split, bm, t64, w2_split, b2, w3_split, b3, relu3, nsplit, n_main, tail_split, out, pointfeat, F16 ? status : nullptr};
  if (mid_mode == 0) return launch<0, 8, F16, MX>(a, s, dev);
  if (mid_mode == 1) return launch<1, 8, F16, MX>(a, s, dev);
  return launch<2, 8, F16, MX>(a, s, dev);
}

extern "C" int cg_pointmlp_max_bf16x3(const float* x, int B, int N, const float* t3, const float* w1, const float* b1,
                                      int mid_mode, const unsigned short* wm_split, const float* bm, const float* t64,
                                      const unsigned short* w2_split, const float* b2, const unsigned short* w3_split,
                                      const float* b3, int relu3, int nsplit, int tile_points, float* out, float* pointfeat,
                                      void* stream) {
  return pointmlp_max_split<false>(x, B, N, t3, w1, b1, mid_mode, wm_split, bm, t64, w2_split, b2, w3_split, b3, relu3, nsplit, tile_points, out, pointfeat, nullptr, stream);
}

extern "C" int cg_pointmlp_max_f16x3(const float* x, int B, int N, const float* t3, const float* w1, const float* b1,
                                      int mid_mode, const unsigned short* wm_split, const float* bm, const float* t64,
                                      const unsigned short* w2_split, const float* b2, const unsigned short* w3_split,
                                      const float* b3, int relu3, int nsplit, int tile_points, float* out, float* pointfeat,
                                      int* status, void* stream) {
  return pointmlp_max_split<true>(x, B, N, t3, w1, b1, mid_mode, wm_split, bm, t64, w2_split, b2, w3_split, b3, relu3, nsplit, tile_points, out, pointfeat, status, stream);
}

// f16fp8x2: as cg_pointmlp_max_f16x3 except that the 128 -> 1024 layer adds its two correction terms with block-scaled e4m3 operands;
// w3_mx is that layer's weight image in the packed format of folding.pack_b_f16fp8x2 (16,640 B per 32 output channels).
extern "C" int cg_pointmlp_max_f16fp8x2(const float* x, int B, int N, const float* t3, const float* w1, const float* b1,
                                         int mid_mode, const unsigned short* wm_split, const float* bm, const float* t64,
                                         const unsigned short* w2_split, const float* b2, const void* w3_mx,
                                         const float* b3, int relu3, int nsplit, int tile_points, float* out, float* pointfeat,
                                         int* status, void* stream) {
  return pointmlp_max_split<true, true>(x, B, N, t3, w1, b1, mid_mode, wm_split, bm, t64, w2_split, b2, (const unsigned short*)w3_mx, b3, relu3, nsplit, tile_points, out,
                                        pointfeat, status, stream);
}
