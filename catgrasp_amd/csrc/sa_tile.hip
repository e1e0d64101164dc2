// PointNet++ set-abstraction layers PAST the first one: wide inputs (3 + D up to ~590 channels), wide layers (hidden <= 512, last any
// multiple of 32), any neighbourhood size, and the group-all layer's input.
// (BASELINE.json north_star: "the PointNet++ set-abstraction encoder ... grouped per-neighbourhood MLP reductions ... LDS-staged
// neighbourhoods ... MFMA only for the dense per-point MLP GEMMs"; the primitives it stacks are pointnet2.py:101-149.)
//
// Replaces the op sequence that consumes sample_and_group's output (pointnet2.py:101-129)
//   new_points (B,S,K,3+D) = cat(xyz[idx] - new_xyz, points[idx])     -> permute -> [Conv2d(1x1) -> BatchNorm2d -> ReLU] x L -> max over K
// for layers the register-resident kernel of setabstraction.hip cannot hold (SA2 of an SSG stack: 3 + 128 inputs, 128-128-256, K = 64;
// the MSG scales: 3 + 320 inputs, K = 128, a 96-wide layer).
//
// sa_tile_kernel: one workgroup (4 wavefronts) carries a tile of 64 grouped rows through every layer inside ONE LDS strip
// (64 x (widest stored activation + 4) floats: 36 KB for SA2 -> several workgroups per CU):
//   * gather: the tile's 64 (neighbourhood, neighbour) rows -- indices, then 16-byte feature loads, rows coalesced; the channel order
//     inside the strip is [features | centred xyz | zero pad to 8] so that the feature copies are aligned 16-byte LDS stores (the host
//     permutes the first layer's weight columns the same way, catgrasp_amd/primitives.py);
//   * a layer: rows are the MFMA M dimension (A operand, ds_read_b128 from the strip), output channels the N dimension (B operand =
//     fragment-packed weights streamed from L2, one k-step ahead, each fragment feeding both 32-row halves).  A wave owns the channel
//     blocks {w, w+4, ...} and keeps ALL its accumulators until every wave has finished reading the strip, so the layer's output
//     overwrites its own input (two workgroup barriers per layer, no second buffer); bias = the accumulators' initial value;
//   * the last layer goes from the accumulators straight into the max over the neighbourhood: a per-lane reduction over accumulator
//     registers + one lane^32 exchange (K <= 32 packs 2 / 4 / 8 neighbourhoods into a tile; K > 64 takes several row tiles with a
//     running max in LDS); its bias + ReLU commute with the max and are applied once per output.
// v_mfma_f32_32x32x2_f32 throughout: exact float32 products and accumulation, like the reference's float32 convolution.
// Algorithmic work per neighbourhood: 2 K sum_l cin_l cout_l flop; HBM bytes: K (8 B index + (3 + D) 4 B gathered) + 12 B + C_L 4 B.
#include <stdlib.h>
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"

namespace {

constexpr int ST_MAX_LAYERS = 4;

struct STArgs {
  const float* xyz; const float* points; const float* new_xyz; const long long* idx;
  int B, N, S, K, D;
  int nlayers; int cin[ST_MAX_LAYERS]; int cout[ST_MAX_LAYERS];
  const float* w[ST_MAX_LAYERS]; const float* b[ST_MAX_LAYERS];
  float* out; long out_bs, out_ss, out_cs;          // out[b * out_bs + s * out_ss + c * out_cs]
  int* err_flag;
  int CS;                                           // strip row stride (floats)
  int KP, RT;                                       // rows of a tile per neighbourhood (8 / 16 / 32 / 64); row tiles per neighbourhood (KP = 64)
  int prio_div;                                     // > 0: wave priority = (blockIdx.x / prio_div) & 3 (see the kernel)
  int append_n;                                     // > 0: channels [C_last, C_last + append_n) of every output row = new_xyz (3) ++ zeros
};

// The MFMA stream of one wave for one layer: NBW channel blocks x NH row halves.  Operand fragments travel DEPTH k-steps ahead of
// their products through a ring of DEPTH + 1 register slots.  DEPTH = 1 everywhere: a lead of 2 or 3 k-steps for the narrow shares
// (one block: 8 MFMAs = 512 cycles per k-step, shorter than an L2 round trip) was measured -- once the loop really kept the ring full
// (see below) -- at 0.706 against 0.736 of the f32 peak (64 clouds): the second and third wavefront of the SIMD already cover the
// round trip, and the extra slots push the narrow instance over its 168-register budget (4 spilled registers).
// arow: this lane's A-fragment row of the first half (strip + (rowbase + l31) * CS + lhi * 4); the second half is 32 rows further.
// wp: packed weights + lane; fragment of (block nb, k-step ks) = wp[(nb * nks + ks) * 64].
template <int NBW, int NH>
__device__ __forceinline__ void st_mma(const float* arow, int CS, const f32x4* wp, int nks, const int (&blk)[NBW], f32x16 (&acc)[NBW][NH]) {
#ifndef ST_DEPTH_NARROW
#define ST_DEPTH_NARROW 1
#endif
  constexpr int DEPTH = NBW * NH <= 2 ? ST_DEPTH_NARROW : 1;
  constexpr int SLOTS = DEPTH + 1;
  f32x4 av[SLOTS][NH], bv[SLOTS][NBW];
  const int last = nks - 1;
  auto load = [&](int ks, int slot) {       // past the end: a repeated, unused load (a conditional one would put a phi on the registers)
    ks = ks < last ? ks : last;
#pragma unroll
    for (int h = 0; h < NH; ++h) av[slot][h] = *(const f32x4*)(arow + h * 32 * CS + ks * 8);
#pragma unroll
    for (int i = 0; i < NBW; ++i) bv[slot][i] = wp[(blk[i] * nks + ks) * 64];
  };
  auto mma = [&](int slot) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < NBW; ++i)
#pragma unroll
        for (int h = 0; h < NH; ++h) acc[i][h] = mfma32(av[slot][h][j], bv[slot][i][j], acc[i][h]);
  };
  auto pin = [&](int slot) {               // a use that keeps the NEXT k-step's operands in this iteration (LLVM otherwise sinks their loads
#pragma unroll                              // in front of their MFMAs); they were requested DEPTH k-steps ago
    for (int h = 0; h < NH; ++h) asm volatile("" : "+v"(av[slot][h]));
#pragma unroll
    for (int i = 0; i < NBW; ++i) asm volatile("" : "+v"(bv[slot][i]));
  };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) load(d, d);
  // Whole turns of the ring in a loop WITHOUT exits inside its body, the remaining < SLOTS k-steps as straight-line code behind it: with a
  // `break` in the unrolled body the compiler sees a path back to the loop header on which the newest load is still in flight and
  // puts `s_waitcnt vmcnt(0)` in front of every turn -- i.e. it drains the ring once per turn (seen in the ISA of the first version).
  int ks = 0;
  for (int turn = nks / SLOTS; turn > 0; --turn) {
#pragma unroll
    for (int u = 0; u < SLOTS; ++u) {
      load(ks + u + DEPTH, (u + DEPTH) % SLOTS);
      __builtin_amdgcn_sched_barrier(0);
      mma(u);
      __builtin_amdgcn_sched_barrier(0);
      pin((u + 1) % SLOTS);
    }
    ks += SLOTS;
  }
  const int rem = nks - ks;
#pragma unroll
  for (int u = 0; u < SLOTS - 1; ++u) {
    if (u < rem) {
      load(ks + u + DEPTH, (u + DEPTH) % SLOTS);
      __builtin_amdgcn_sched_barrier(0);
      mma(u);
      __builtin_amdgcn_sched_barrier(0);
      pin((u + 1) % SLOTS);
    }
  }
}

// Hidden layer for one wave: NBW blocks {nb0, nb0 + nbs, ...} x NH halves (NH = 2: both halves; NH = 1: the half at rowbase).
// Every wave of the workgroup runs exactly two barriers per hidden layer, whatever its share (st_hidden_idle for a wave without one).
template <int NBW, int NH>
__device__ __forceinline__ void st_hidden(float* strip, int CS, const float* w, const float* bias, int nks, int nb0, int nbs, int rowbase, int lane) {
  const int l31 = lane & 31, lhi = lane >> 5;
  int blk[NBW];
  f32x16 acc[NBW][NH];
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    blk[i] = nb0 + i * nbs;
    const float bq = bias[blk[i] * 32 + l31];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][h][r] = bq;
  }
  st_mma<NBW, NH>(strip + (rowbase + l31) * CS + lhi * 4, CS, (const f32x4*)w + lane, nks, blk, acc);
  __syncthreads();                         // every wave has read its last fragment of the strip: the layer may overwrite its input
  // one opaque per-lane base + wave-uniform row offsets: left to itself LLVM hoists all 16 x NBW x NH store addresses of every layer
  // shape out of the tile loop and keeps them in (spilled) registers
  int base = (rowbase + 4 * lhi) * CS + l31;
  asm volatile("" : "+v"(base));
#pragma unroll
  for (int i = 0; i < NBW; ++i)
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        strip[base + ((r & 3) + 8 * (r >> 2) + 32 * h) * CS + blk[i] * 32] = fmaxf(acc[i][h][r], 0.f);
  __syncthreads();
}
__device__ __forceinline__ void st_hidden_idle() { __syncthreads(); __syncthreads(); }

// max over the accumulator registers [r0, r0 + n) of a lane
template <int R0, int NR>
__device__ __forceinline__ float st_max_regs(const f32x16& c) {
  float m = c[R0];
#pragma unroll
  for (int r = 1; r < NR; ++r) m = fmaxf(m, c[R0 + r]);
  return m;
}

// Last layer for one wave, NBW blocks (all NHT 32-row halves of the tile) per pass: accumulators -> neighbourhood maxima -> out (+ bias,
// ReLU) or the running maxima in LDS (several row tiles per neighbourhood).  KP = rows per neighbourhood in the tile.
template <int NBW, int KP, int NHT>
__device__ __forceinline__ void st_last(const STArgs& a, const float* strip, const float* w, const float* bias, int nks, int nb0, int lane,
                                        int kt, float* rmax, const long* goff) {
  const int l31 = lane & 31, lhi = lane >> 5;
  int blk[NBW];
  f32x16 acc[NBW][NHT];
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    blk[i] = nb0 + i * 4;
#pragma unroll
    for (int h = 0; h < NHT; ++h) acc[i][h] = f32x16{0};
  }
  st_mma<NBW, NHT>(strip + l31 * a.CS + lhi * 4, a.CS, (const f32x4*)w + lane, nks, blk, acc);
  // goff[j]: offset of the tile's j-th neighbourhood inside `out`, or -1 for the padding of the last tile
  auto store = [&](int j, int ch, float m) {
    const long o = goff[j];
    if (o >= 0) a.out[o + (long)ch * a.out_cs] = fmaxf(m + bias[ch], 0.f);
  };
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    const int ch = blk[i] * 32 + l31;
    if constexpr (KP >= 32) {
      constexpr int HPN = KP / 32;         // 32-row halves per neighbourhood; NHT / HPN neighbourhoods in the tile
#pragma unroll
      for (int j = 0; j < NHT / HPN; ++j) {
        float m = max16(acc[i][j * HPN]);
#pragma unroll
        for (int h = 1; h < HPN; ++h) m = fmaxf(m, max16(acc[i][j * HPN + h]));
        m = fmaxf(m, __shfl_xor(m, 32));
        if (lane < 32) {
          if (HPN < NHT || a.RT == 1) store(j, ch, m);
          else {                           // the tile is one neighbourhood's row tile kt of RT: running maximum in LDS
            if (kt > 0) m = fmaxf(m, rmax[ch]);
            if (kt == a.RT - 1) store(0, ch, m); else rmax[ch] = m;
          }
        }
      }
    } else {
      constexpr int NPH = 32 / KP;         // neighbourhoods per 32-row half; registers [p * 16 / NPH, ...) hold the rows of neighbourhood p
      constexpr int RP = 16 / NPH;
#pragma unroll
      for (int h = 0; h < NHT; ++h) {
        float m[NPH];
        if constexpr (NPH == 2) { m[0] = st_max_regs<0, RP>(acc[i][h]); m[1] = st_max_regs<RP, RP>(acc[i][h]); }
        if constexpr (NPH == 4) {
          m[0] = st_max_regs<0, RP>(acc[i][h]); m[1] = st_max_regs<RP, RP>(acc[i][h]);
          m[2] = st_max_regs<2 * RP, RP>(acc[i][h]); m[3] = st_max_regs<3 * RP, RP>(acc[i][h]);
        }
#pragma unroll
        for (int p = 0; p < NPH; ++p) {
          const float v = fmaxf(m[p], __shfl_xor(m[p], 32));
          if (lane < 32) store(h * NPH + p, ch, v);
        }
      }
    }
  }
}

// KP: rows of a tile per neighbourhood.  WIDE: hidden layers up to 512 channels (4 blocks x 2 halves of accumulators per wave) -- the
// narrow instance (<= 256) needs half the registers and so holds twice the wavefronts.  TR: rows per tile -- 64, or 128 for launches
// with enough rows to fill the chip that way: half the barriers, gathers and pipeline fills per matrix FMA, each weight fragment
// feeding four 32-row halves (narrow instance only: a two-block share is 128 accumulator registers).
template <int KP, bool WIDE, int TR>
__global__ __launch_bounds__(256, WIDE ? 1 : (TR == 128 ? 2 : 3)) void sa_tile_kernel(STArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int CS = a.CS;
  const int c_last = a.cout[a.nlayers - 1];
  float* strip = smem;
  float* rmax = strip + TR * CS;
  long* goff = (long*)(rmax + c_last);          // (TR * CS + c_last) * 4 is a multiple of 16
  int* pbase = (int*)(goff + 16);               // row -> b * N + point index
  float* bias_s = (float*)(pbase + TR);         // the biases of every layer, staged once per (persistent) workgroup: a global load at the
                                                // head of every layer of every tile is an L2 round trip the matrix pipe waits out
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  constexpr int NPT = TR / KP;
  const int G = a.B * a.S;                      // < 2^31 / 8 (checked by the launcher); so is B * N < 2^31
  const int nslots = KP == TR ? G : (G + NPT - 1) / NPT;
  const int RT = KP == TR ? a.RT : 1;
  constexpr int NHT = TR / 32;
  const int D = a.D, cin0 = a.cin[0];

  auto row_id = [&](int slot, int kt, int r, int& g_out) -> long long {
    int g, k;
    if (KP == TR) { g = slot; k = kt * TR + r; } else { g = slot * NPT + r / KP; k = r % KP; }
    if (g >= G) g = G - 1;                     // tail slot: a repeated neighbourhood whose result is not stored
    if (k >= a.K) k = 0;                       // short neighbourhood: neighbour 0 again (the max is idempotent)
    g_out = g;
    return a.idx ? a.idx[(long)g * a.K + k] : (long long)k;      // no index list: the group-all layer, row k = point k
  };

  // XCD-aware slot order: workgroups are dealt to the 8 XCDs round robin (blockIdx % 8), each XCD has its own L2, and consecutive slots
  // gather from the same cloud.  With slot = blockIdx every XCD's L2 pulls its own copy of every cloud's rows (measured: 2 x 16.3 MB
  // fetched per launch at 16 clouds against 4.2 MB of rows); giving XCD x the x-th contiguous eighth of the slots makes each cloud's rows
  // one L2's business.  (Grids that are not a multiple of 8 -- launches smaller than the chip -- keep the plain order.)
  const bool by_xcd = (gridDim.x & 7) == 0 && nslots >= (int)gridDim.x;
  const int per_xcd = (nslots + 7) >> 3;
  const int slot_end = by_xcd ? min(nslots, ((int)(blockIdx.x & 7) + 1) * per_xcd) : nslots;
  const int slot_step = by_xcd ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  int slot = by_xcd ? (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x, kt = 0;
  if (slot >= slot_end) return;
  int b_off[ST_MAX_LAYERS];
  {
    int o = 0;
    for (int l = 0; l < a.nlayers; ++l) {
      b_off[l] = o;
      for (int i = tid; i < a.cout[l]; i += 256) bias_s[o + i] = a.b[l][i];
      o += a.cout[l];
    }
  }                                             // visible after the first tile's barriers
  // The workgroups resident on one CU run the same phases for the same time: under fair issue they stay in lockstep -- all in their
  // matrix phase together, all in their gather / store phases together -- and the matrix pipe idles through the latter.  Unequal wave
  // priorities break the tie: the k-th workgroup of a CU (dispatch order: blockIdx / #CUs) gets priority k, finishes its matrix phase
  // first and gathers while the others multiply.
  if (a.prio_div > 0) {
    switch ((blockIdx.x / a.prio_div) & 3) {
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      case 3: __builtin_amdgcn_s_setprio(3); break;
      default: break;
    }
  }
  long long id_next = 0; int g_next = 0;
  if (tid < TR) id_next = row_id(slot, 0, tid, g_next);
  for (;;) {
    __syncthreads();                           // the previous tile's last layer has read the strip (and goff)
    int g_row = 0, p_row = 0;
    if (tid < TR) {
      long long id = id_next;
      if (id < 0 || id >= a.N) { if (a.err_flag) *a.err_flag = 1; id = 0; }      // index_points raises on such an index
      g_row = g_next;
      p_row = (g_row / a.S) * a.N + (int)id;
      pbase[tid] = p_row;
    }
    if (tid >= TR && tid < TR + NPT) {          // where the tile's neighbourhoods go in `out`
      const int g = (KP == TR ? slot : slot * NPT) + (tid - TR);
      const int b = g / a.S;
      const long o = g < G ? (long)b * a.out_bs + (long)(g - b * a.S) * a.out_ss : -1;
      goff[tid - TR] = o;
      if (a.append_n > 0 && o >= 0 && kt == 0) {      // the row the NEXT level's group-all GEMM reads: [features | xyz | zero pad]
        const float* pc = a.new_xyz + (size_t)g * 3;
        for (int e = 0; e < a.append_n; ++e) a.out[o + (long)(c_last + e) * a.out_cs] = e < 3 ? pc[e] : 0.f;
      }
    }
    __syncthreads();
    int nslot = slot, nkt = kt + 1;
    if (nkt >= RT) { nkt = 0; nslot = slot + slot_step; }
    const bool more = nslot < slot_end;
    if (tid < TR && more) id_next = row_id(nslot, nkt, tid, g_next);             // in flight under this tile's gather and layers
    // ---- gather: features (the strip's first D channels), then centred xyz, then the zero pad.  The coordinate loads go out first and
    // are consumed last: they ride along with the feature loads instead of adding a round trip of their own
    float px0 = 0.f, px1 = 0.f, px2 = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
    if (tid < TR) {
      const float* px = a.xyz + (size_t)p_row * 3;
      px0 = px[0]; px1 = px[1]; px2 = px[2];
      if (a.new_xyz) { const float* pc = a.new_xyz + (size_t)g_row * 3; cx = pc[0]; cy = pc[1]; cz = pc[2]; }
    }
    if (D > 0) {
      if ((D & 3) == 0 && ((uintptr_t)a.points & 15) == 0) {
        // 16-byte pieces, GU of them in flight per thread: written as load-then-store per piece the compiler waits out one L2 round
        // trip per piece (8 in a row for D = 128: ~6 us of a 14 us tile)
        constexpr int GU = 8;
        const int nq = D >> 2, total = TR * nq;
        for (int base = tid; base < total; base += 256 * GU) {
          f32x4 v[GU];
          int off[GU];
#pragma unroll
          for (int u = 0; u < GU; ++u) {
            const int i = base + u * 256;
            const int ic = i < total ? i : total - 1;
            const int row = ic / nq, q = ic - row * nq;
            off[u] = i < total ? row * CS + 4 * q : -1;
            v[u] = *(const f32x4*)(a.points + (size_t)pbase[row] * D + 4 * q);
          }
#pragma unroll
          for (int u = 0; u < GU; ++u)
            if (off[u] >= 0) *(f32x4*)(strip + off[u]) = v[u];
        }
      } else {
        for (int i = tid; i < TR * D; i += 256) {
          const int row = i / D, c = i - row * D;
          strip[row * CS + c] = a.points[(size_t)pbase[row] * D + c];
        }
      }
    }
    if (tid < TR) {
      float* dst = strip + tid * CS + D;
      dst[0] = px0 - cx; dst[1] = px1 - cy; dst[2] = px2 - cz;
      for (int c = D + 3; c < cin0; ++c) strip[tid * CS + c] = 0.f;
    }
    __syncthreads();
    // ---- the layers
    for (int l = 0; l < a.nlayers; ++l) {
      const int nks = a.cin[l] >> 3, nb = a.cout[l] >> 5;
      if (l + 1 < a.nlayers) {
        if (nb >= 4) {                          // a wave: blocks {wv, wv + 4, ...}, every 32-row half (one weight fragment feeds them all)
          const int cnt = nb > wv ? (nb - wv + 3) >> 2 : 0;
          switch (cnt) {
            case 1: st_hidden<1, NHT>(strip, CS, a.w[l], bias_s + b_off[l], nks, wv, 4, 0, lane); break;
            case 2: st_hidden<2, NHT>(strip, CS, a.w[l], bias_s + b_off[l], nks, wv, 4, 0, lane); break;
            case 3: if constexpr (WIDE) { st_hidden<3, NHT>(strip, CS, a.w[l], bias_s + b_off[l], nks, wv, 4, 0, lane); break; }
            case 4: if constexpr (WIDE) { st_hidden<4, NHT>(strip, CS, a.w[l], bias_s + b_off[l], nks, wv, 4, 0, lane); break; }
            default: st_hidden_idle(); break;
          }
        } else {                                // 32 / 64 / 96 channels: a wave takes ONE half of the rows and the blocks {wv >> 1, (wv >> 1) + 2}
          const int p = wv >> 1, cnt = nb > p ? (nb - p + 1) >> 1 : 0;
          switch (cnt) {
            case 1: st_hidden<1, NHT / 2>(strip, CS, a.w[l], bias_s + b_off[l], nks, p, 2, (TR / 2) * (wv & 1), lane); break;
            case 2: st_hidden<2, NHT / 2>(strip, CS, a.w[l], bias_s + b_off[l], nks, p, 2, (TR / 2) * (wv & 1), lane); break;
            default: st_hidden_idle(); break;
          }
        }
      } else {
        const int cnt = nb > wv ? (nb - wv + 3) >> 2 : 0;
        for (int i0 = 0; i0 < cnt; i0 += 2) {
          if (cnt - i0 >= 2) st_last<2, KP, NHT>(a, strip, a.w[l], bias_s + b_off[l], nks, wv + 4 * i0, lane, kt, rmax, goff);
          else st_last<1, KP, NHT>(a, strip, a.w[l], bias_s + b_off[l], nks, wv + 4 * i0, lane, kt, rmax, goff);
        }
      }
    }
    if (!more) break;
    slot = nslot; kt = nkt;
  }
}

// rows of [features | xyz | zero pad]: the input matrix of the group-all layer's GEMM chain (sample_and_group_all, pointnet2.py:132-149:
// grouped_xyz is xyz itself, not centred)
__global__ void sa_concat_kernel(const float* xyz, const float* points, long rows, int D, int ld, float* out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * ld) return;
  const long r = i / ld; const int c = (int)(i - r * ld);
  float v = 0.f;
  if (c < D) v = points[r * D + c];
  else if (c < D + 3) v = xyz[r * 3 + (c - D)];
  out[i] = v;
}

template <int KP, bool WIDE, int TR>
int launch_st(const STArgs& a, size_t lds, long nslots, int dev, hipStream_t s) {
  auto kern = sa_tile_kernel<KP, WIDE, TR>;
  static bool attr_set[CG_MAX_DEVICES] = {};
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set[dev] = true;
  }
  const int n_cu = cg_device_cu_count(dev);
  if (n_cu <= 0) return CG_ERR_UNSUPPORTED;
  int per_cu = (int)((160 * 1024) / lds);
  const int reg_cap = WIDE ? 1 : (TR == 128 ? 2 : 3);      // workgroups of 4 waves a CU's registers hold
  if (per_cu > reg_cap) per_cu = reg_cap;
  static const char* pc_env = getenv("CATGRASP_AMD_SAT_PER_CU");        // dev knob: resident workgroups per CU
  if (pc_env && atoi(pc_env) > 0 && atoi(pc_env) < per_cu) per_cu = atoi(pc_env);
  if (per_cu < 1) per_cu = 1;
  long grid = nslots < (long)n_cu * per_cu ? nslots : (long)n_cu * per_cu;        // persistent: a workgroup walks slots blockIdx, + grid, ...
  STArgs b = a;
  static const char* pr_env = getenv("CATGRASP_AMD_SAT_PRIO");           // dev knob: 1 = unequal wave priorities per resident workgroup
  b.prio_div = (pr_env && atoi(pr_env) == 1) ? n_cu : 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, b);
  return cg_hip_status(hipGetLastError());
}

template <bool WIDE>
int launch_st_kp(const STArgs& a, size_t lds, long nslots, int dev, hipStream_t s) {
  switch (a.KP) {
    case 8: return launch_st<8, WIDE, 64>(a, lds, nslots, dev, s);
    case 16: return launch_st<16, WIDE, 64>(a, lds, nslots, dev, s);
    case 32: return launch_st<32, WIDE, 64>(a, lds, nslots, dev, s);
  }
  return launch_st<64, WIDE, 64>(a, lds, nslots, dev, s);
}
int launch_st_kp128(const STArgs& a, size_t lds, long nslots, int dev, hipStream_t s) {
  switch (a.KP) {
    case 8: return launch_st<8, false, 128>(a, lds, nslots, dev, s);
    case 16: return launch_st<16, false, 128>(a, lds, nslots, dev, s);
    case 32: return launch_st<32, false, 128>(a, lds, nslots, dev, s);
    case 64: return launch_st<64, false, 128>(a, lds, nslots, dev, s);
  }
  return launch_st<128, false, 128>(a, lds, nslots, dev, s);
}

}  // namespace

extern "C" int cg_sa_tile_mlp_max(const float* xyz, const float* points, const float* new_xyz, const long long* idx, int B, int N, int S,
                                  int K, int D, int n_layers, const int* h_cin, const int* h_cout, const float* const* h_w_packed,
                                  const float* const* h_bias, float* out, long out_bs, long out_ss, long out_cs, int append_xyz,
                                  int* err_flag, void* stream) {
  if (B < 0 || N <= 0 || S < 0 || K <= 0 || D < 0 || n_layers < 1 || n_layers > ST_MAX_LAYERS) return CG_ERR_ARG;
  if (!h_cin || !h_cout || !h_w_packed || !h_bias) return CG_ERR_ARG;
  if ((long)B * S == 0) return CG_OK;
  if (!xyz || !out || (D > 0 && !points)) return CG_ERR_ARG;
  if (!idx && (S != 1 || K != N)) return CG_ERR_ARG;             // no index list: the group-all layer (one group of all N points)
  if (append_xyz != 0 && (append_xyz < 3 || append_xyz > 16 || !new_xyz)) return CG_ERR_ARG;
  if ((long)B * S >= 0x7fffffffL / 8 || (long)B * N >= 0x7fffffffL) return CG_ERR_UNSUPPORTED;
  STArgs a{};
  a.xyz = xyz; a.points = D > 0 ? points : nullptr; a.new_xyz = new_xyz; a.idx = idx;
  a.B = B; a.N = N; a.S = S; a.K = K; a.D = D; a.nlayers = n_layers;
  a.out = out; a.out_bs = out_bs; a.out_ss = out_ss; a.out_cs = out_cs; a.err_flag = err_flag; a.append_n = append_xyz;
  int cstore = 0, hidden_max = 0;
  for (int l = 0; l < n_layers; ++l) {
    if (!h_w_packed[l] || !h_bias[l]) return CG_ERR_ARG;
    const int want = l == 0 ? ((3 + D + 7) & ~7) : h_cout[l - 1];       // layer 0: [features | xyz] zero padded to a multiple of 8 by the host
    if (h_cin[l] != want) return CG_ERR_ARG;
    if (h_cout[l] <= 0 || (h_cout[l] % 32) != 0) return CG_ERR_UNSUPPORTED;
    if (((uintptr_t)h_w_packed[l] & 15) != 0) return CG_ERR_ARG;
    a.cin[l] = h_cin[l]; a.cout[l] = h_cout[l]; a.w[l] = h_w_packed[l]; a.b[l] = h_bias[l];
    if (h_cin[l] > cstore) cstore = h_cin[l];
    if (l + 1 < n_layers && h_cout[l] > hidden_max) hidden_max = h_cout[l];
  }
  if (hidden_max > 512) return CG_ERR_UNSUPPORTED;               // a hidden layer's accumulators must fit one wave's registers (4 blocks x 2 halves)
  a.CS = cstore + 4;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= CG_MAX_DEVICES) return CG_ERR_UNSUPPORTED;
  const int n_cu = cg_device_cu_count(dev);
  if (n_cu <= 0) return CG_ERR_UNSUPPORTED;
  const long G = (long)B * S;
  size_t bias_floats = 0;
  for (int l = 0; l < n_layers; ++l) bias_floats += (size_t)a.cout[l];
  const size_t tail = (size_t)a.cout[n_layers - 1] * 4 + 16 * sizeof(long) + bias_floats * 4;      // running maxima, output offsets, biases
  // 128-row tiles when the launch still fills the chip with them (>= 2 tiles per CU: one round of two resident workgroups), the strip
  // leaves room for two workgroups per CU and no hidden layer needs the wide instance; 64-row tiles otherwise
  const int kp128 = K <= 8 ? 8 : K <= 16 ? 16 : K <= 32 ? 32 : K <= 64 ? 64 : 128;
  const long slots128 = kp128 == 128 ? G : (G + 128 / kp128 - 1) / (128 / kp128);
  const size_t lds128 = (size_t)128 * a.CS * 4 + tail + 128 * sizeof(int);
  static const char* tr_env = getenv("CATGRASP_AMD_SAT_TILE_ROWS");      // dev knob: 64 / 128
  bool use128 = hidden_max <= 256 && lds128 <= 80 * 1024 && slots128 * (kp128 == 128 ? (K + 127) / 128 : 1) >= 2L * n_cu;
  if (tr_env) use128 = atoi(tr_env) == 128 && hidden_max <= 256 && lds128 <= 158 * 1024;
  if (use128) {
    a.KP = kp128; a.RT = kp128 == 128 ? (K + 127) / 128 : 1;
    return launch_st_kp128(a, lds128, slots128, dev, (hipStream_t)stream);
  }
  a.KP = K <= 8 ? 8 : K <= 16 ? 16 : K <= 32 ? 32 : 64;
  a.RT = a.KP == 64 ? (K + 63) / 64 : 1;
  const size_t lds = (size_t)64 * a.CS * 4 + tail + 64 * sizeof(int);
  if (lds > 158 * 1024) return CG_ERR_UNSUPPORTED;
  const int npt = 64 / a.KP;
  const long nslots = a.KP == 64 ? G : (G + npt - 1) / npt;
  if (hidden_max > 256) return launch_st_kp<true>(a, lds, nslots, dev, (hipStream_t)stream);
  return launch_st_kp<false>(a, lds, nslots, dev, (hipStream_t)stream);
}

extern "C" int cg_sa_concat_input(const float* xyz, const float* points, long rows, int D, int ld, float* out, void* stream) {
  if (rows < 0 || D < 0 || ld < D + 3) return CG_ERR_ARG;
  if (rows == 0) return CG_OK;
  if (!xyz || !out || (D > 0 && !points)) return CG_ERR_ARG;
  const long n = rows * ld;
  hipLaunchKernelGGL(sa_concat_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, xyz, points, rows, D, ld, out);
  return cg_hip_status(hipGetLastError());
}
