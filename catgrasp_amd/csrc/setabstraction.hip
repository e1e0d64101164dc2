// Fused PointNet++ set-abstraction layer: group -> shared per-neighbour MLP -> max over the neighbourhood
// (BASELINE.json north_star: "grouped per-neighbourhood MLP reductions ... LDS-staged neighbourhoods and wavefront shuffle
// reductions, MFMA only for the dense per-point MLP GEMMs").
//
// Replaces the op sequence that consumes sample_and_group's output (pointnet2.py:101-129):
//   new_points (B,S,K,3+D) = cat(xyz[idx] - new_xyz, points[idx])            pointnet2.py:116-123
//   -> permute to (B,3+D,K,S) -> [Conv2d(1x1) -> BatchNorm2d -> ReLU] x L -> max over K -> (B,C_L,S)
// without ever materialising the grouped tensor (K x the input in HBM) or any activation.
//
// Two kernels.  sa_reg_kernel (all layer widths <= 128): the ACTIVATIONS NEVER LEAVE REGISTERS.  Hidden layers run "transposed" --
// weights are the MFMA A operand (rows = output channels), the neighbours' activations the B operand (columns = neighbours) -- so a
// layer's accumulator registers ARE the next layer's operand fragments: lane (neighbour n, half h) holds, in register 4q+j of channel
// block nb, channel 32nb + 8q + 4h + j, which is exactly the element the next layer's MFMA j of k-step 4nb+q wants from that lane
// (v_mfma_f32_32x32x2_f32 takes A[i = lane&31][k = lane>>5] and B[k = lane>>5][j = lane&31] from the same lane positions).  Bias
// = the accumulators' initial value (read from LDS straight into them), ReLU = one v_max per register, no LDS round trip, no
// wave barrier between layers.  The LAST layer flips back (activations = A, weights = B), so its rows are the neighbours and the
// max over a neighbourhood is a per-lane reduction over accumulator registers + one lane^32 exchange; its bias + ReLU commute with
// the max (same bias for every row, monotone) and are applied once per output.  LDS holds only the weights (53 KB for 9-64-64-128).
// sa_group_mlp_max_kernel (a layer wider than 128): the round-2/3 strip kernel below, activations staged in a wave-private LDS strip.
//
// Common to both: PERSISTENT workgroups with the layer weights resident in LDS.
//   * A workgroup stages the folded, fragment-packed weights + biases of every layer into LDS ONCE (53 KB for 9-64-64-128) and then
//     loops over neighbourhoods; the B operand of every MFMA is a conflict-free ds_read_b128 instead of an L2 round trip per
//     k-step per wave (round 2: ~700 cycles exposed per 8..16 MFMAs).  Layers that do not fit next to the strips stay in L2.
//   * One wavefront owns one 32-row MFMA tile = PACK neighbourhoods of 32/PACK rows (K <= 8: 4, K <= 16: 2, else 1 and K > 32 takes
//     several row tiles with a running max; short neighbourhoods are padded by repeating neighbour 0 -- the max is idempotent).
//     The accumulator registers of a lane split by neighbourhood (rows 8*(r>>2) + (r&3) + 4*(lane>>5)), so the segmented max
//     is still a per-lane reduction + one lane^32 exchange.
//   * The gather of the NEXT tile (index -> point -> centre, two dependent L2 accesses) is issued while the current tile is in the
//     matrix pipe: indices two tiles ahead, coordinates/features one tile ahead, both held in registers.
//   * Every layer computes ALL its output channel blocks before storing anything, so it overwrites its own input strip; fragments
//     of the next k-step are requested before the current k-step's MFMAs.  No workgroup barrier after the staging.
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"

namespace {

constexpr int SA_MAX_LAYERS = 4;
constexpr int C0 = 16;          // first-layer input channels (3 + D) padded to 16

struct SAArgs {
  const float* xyz; const float* points; const float* new_xyz; const long long* idx;
  int B, N, S, K, D;
  int nlayers; int cin[SA_MAX_LAYERS]; int cout[SA_MAX_LAYERS];
  const float* w[SA_MAX_LAYERS]; const float* b[SA_MAX_LAYERS];
  int w_off[SA_MAX_LAYERS];       // float offset of the layer's packed weights inside the LDS weight region, or -1: read from L2
  int b_off[SA_MAX_LAYERS];       // float offset of the layer's bias (always staged)
  int wb_floats;                  // size of the LDS weight + bias region
  float* out; long out_bs, out_ss, out_cs;       // out[b * out_bs + s * out_ss + c * out_cs]
  int* err_flag;
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// One layer for the wave's 32-row tile.  w: fragment-packed weights Wp[nb][ks][lane][4] (LDS or global), bias: LDS.
// Last layer: the ReLU'd accumulators go straight into the running maxima run[nb][p] of the tile's PACK neighbourhoods.
template <int NNB, int PACK>
__device__ __forceinline__ void sa_layer(float* strip, int CS, const float* w, const float* bias, int nks, bool last, int lane,
                                         float (*run)[PACK]) {
  const int l31 = lane & 31, lhi = lane >> 5;
  f32x16 c[NNB];
#pragma unroll
  for (int nb = 0; nb < NNB; ++nb) {
    const float bv = bias[nb * 32 + l31];
#pragma unroll
    for (int r = 0; r < 16; ++r) c[nb][r] = bv;
  }
  const float* arow = strip + l31 * CS + lhi * 4;
  const f32x4* wp = (const f32x4*)w + lane;
  if constexpr (NNB <= 4) {
    // software pipeline, one k-step deep: the fragments of k-step ks+1 are requested before the MFMAs of k-step ks
    f32x4 av = *(const f32x4*)arow;
    f32x4 bv[NNB];
#pragma unroll
    for (int nb = 0; nb < NNB; ++nb) bv[nb] = wp[(size_t)(nb * nks) * 64];
    for (int ks = 0; ks < nks; ++ks) {
      const int kn = ks + 1 < nks ? ks + 1 : ks;
      f32x4 an = *(const f32x4*)(arow + kn * 8);
      f32x4 bn[NNB];
#pragma unroll
      for (int nb = 0; nb < NNB; ++nb) bn[nb] = wp[(size_t)(nb * nks + kn) * 64];
      __builtin_amdgcn_sched_barrier(0);                  // the requests go out BEFORE the MFMA block, whatever the scheduler prefers
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < NNB; ++nb) c[nb] = mfma32(av[j], bv[nb][j], c[nb]);
      // Pin the prefetch to THIS iteration: without a use here LLVM sinks the loads to the top of the next iteration (right in front
      // of the MFMAs that need them), i.e. it undoes the software pipeline.  The empty asm is the use; by now the data has landed.
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("" : "+v"(an));
#pragma unroll
      for (int nb = 0; nb < NNB; ++nb) asm volatile("" : "+v"(bn[nb]));
      av = an;
#pragma unroll
      for (int nb = 0; nb < NNB; ++nb) bv[nb] = bn[nb];
    }
  } else {
    // 5..8 channel blocks: 80..128 accumulator registers leave no room for a second fragment set (it would spill); 20..32 MFMAs per
    // k-step and the SIMD's other wave cover the fragment latency instead
    for (int ks = 0; ks < nks; ++ks) {
      const f32x4 av = *(const f32x4*)(arow + ks * 8);
      f32x4 bv[NNB];
#pragma unroll
      for (int nb = 0; nb < NNB; ++nb) bv[nb] = wp[(size_t)(nb * nks + ks) * 64];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int nb = 0; nb < NNB; ++nb) c[nb] = mfma32(av[j], bv[nb][j], c[nb]);
    }
  }
  wave_sync();                       // every lane's fragment reads of the strip are done before it is overwritten
#pragma unroll
  for (int nb = 0; nb < NNB; ++nb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) c[nb][r] = fmaxf(c[nb][r], 0.f);
    if (!last) {
#pragma unroll
      for (int r = 0; r < 16; ++r) strip[acc_row(r, lane) * CS + nb * 32 + l31] = c[nb][r];
    } else {
      constexpr int RP = 16 / PACK;     // accumulator registers per neighbourhood: r in [p*RP, (p+1)*RP) <-> rows [p*32/PACK, ...)
#pragma unroll
      for (int p = 0; p < PACK; ++p) {
        float m = c[nb][p * RP];
#pragma unroll
        for (int r = 1; r < RP; ++r) m = fmaxf(m, c[nb][p * RP + r]);
        m = fmaxf(m, __shfl_xor(m, 32));
        run[nb][p] = fmaxf(run[nb][p], m);
      }
    }
  }
  wave_sync();
}

template <int PACK, bool WIDE>
__device__ __forceinline__ void sa_layer_any(int nnb, float* strip, int CS, const float* w, const float* bias, int nks, bool last, int lane,
                                             float (*run)[PACK]) {
  if (nnb == 1) sa_layer<1, PACK>(strip, CS, w, bias, nks, last, lane, run);
  else if (nnb == 2) sa_layer<2, PACK>(strip, CS, w, bias, nks, last, lane, run);
  else if (nnb == 3) sa_layer<3, PACK>(strip, CS, w, bias, nks, last, lane, run);
  else if (nnb == 4) sa_layer<4, PACK>(strip, CS, w, bias, nks, last, lane, run);
  else if constexpr (WIDE) {
    if (nnb == 5) sa_layer<5, PACK>(strip, CS, w, bias, nks, last, lane, run);
    else if (nnb == 6) sa_layer<6, PACK>(strip, CS, w, bias, nks, last, lane, run);
    else if (nnb == 7) sa_layer<7, PACK>(strip, CS, w, bias, nks, last, lane, run);
    else sa_layer<8, PACK>(strip, CS, w, bias, nks, last, lane, run);
  }
}

// CS: row stride (floats) of the wave-private activation strip = widest STORED activation (layer inputs; the last layer's output
// goes straight from the accumulators into the max) + 4 -> conflict-free 16-byte fragment reads.
// MAXNB: widest layer / 32 -- 4 (all widths <= 128) or 8.  PACK: neighbourhoods per 32-row tile.
template <int MAXNB, int PACK>
__global__ __launch_bounds__(512) void sa_group_mlp_max_kernel(SAArgs a, int CS, int WAVES) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  // ---- stage the weights (16-byte copies, all threads) and biases once per workgroup ----
  for (int l = 0; l < a.nlayers; ++l) {
    if (a.w_off[l] >= 0) {
      const int n4 = a.cin[l] * a.cout[l] / 4;
      const f32x4* src = (const f32x4*)a.w[l];
      f32x4* dst = (f32x4*)(smem + a.w_off[l]);
      for (int i = threadIdx.x; i < n4; i += blockDim.x) dst[i] = src[i];
    }
    for (int i = threadIdx.x; i < a.cout[l]; i += blockDim.x) smem[a.b_off[l] + i] = a.b[l][i];
  }
  __syncthreads();
  float* strip = smem + a.wb_floats + (size_t)wv * 32 * CS;

  constexpr int RPN = 32 / PACK;                         // rows per neighbourhood
  const int G = a.B * a.S;                               // < 2^31 (checked by the launcher)
  const int nslots = (G + PACK - 1) / PACK;              // a slot = the PACK neighbourhoods of one tile
  const int RT = PACK == 1 ? (a.K + 31) / 32 : 1;        // row tiles per slot
  const int stride = gridDim.x * WAVES;
  const int c_last = a.cout[a.nlayers - 1];
  struct Tile { int slot; int kt; };                     // the wave's tiles in order: (slot0, 0..RT-1), (slot0 + stride, 0..RT-1), ...
  auto next = [&](Tile t) { return t.kt + 1 < RT ? Tile{t.slot, t.kt + 1} : Tile{t.slot + stride, 0}; };
  Tile cur{(int)blockIdx.x * WAVES + wv, 0};
  if (cur.slot >= nslots) return;

  // lane's row r = l31 belongs to neighbourhood p = r / RPN of the tile's slot
  auto nbhd_of = [&](Tile t) -> int {
    const int g = t.slot * PACK + l31 / RPN;
    return g < G ? g : G - 1;                            // tail slot: duplicate the last neighbourhood (its result is not stored)
  };
  auto load_id = [&](Tile t) -> long long {
    const int kk = PACK == 1 ? t.kt * 32 + l31 : l31 % RPN;
    return a.idx[(size_t)nbhd_of(t) * a.K + (kk < a.K ? kk : 0)];
  };
  // lane (row r, half h) gathers channels [8h, 8h+8) of its row: centred xyz ++ features, zero padded to 16
  auto gather = [&](Tile t, long long id, float* v) {
    const int g = nbhd_of(t);
    if (id < 0 || id >= a.N) { if (a.err_flag) *a.err_flag = 1; id = 0; }        // index_points raises on such an index
    const int b = g / a.S;
    const float* px = a.xyz + ((size_t)b * a.N + id) * 3;
    const float* pf = a.points ? a.points + ((size_t)b * a.N + id) * a.D : nullptr;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ch = 8 * lhi + j;
      float x = 0.f;
      if (ch < 3) x = px[ch] - a.new_xyz[(size_t)g * 3 + ch];
      else if (ch - 3 < a.D) x = pf[ch - 3];
      v[j] = x;
    }
  };

  float run[MAXNB][PACK];
  float v[8];
  gather(cur, load_id(cur), v);
  Tile nxt = next(cur);
  long long id_next = nxt.slot < nslots ? load_id(nxt) : 0;
  for (;;) {
    if (cur.kt == 0) {
#pragma unroll
      for (int q = 0; q < MAXNB; ++q)
#pragma unroll
        for (int p = 0; p < PACK; ++p) run[q][p] = -INFINITY;
    }
    *(f32x4*)(strip + l31 * CS + 8 * lhi) = f32x4{v[0], v[1], v[2], v[3]};
    *(f32x4*)(strip + l31 * CS + 8 * lhi + 4) = f32x4{v[4], v[5], v[6], v[7]};
    wave_sync();
    // next tile's rows (their indices arrived during the previous tile) and the indices of the tile after: in flight during the layers
    const bool more = nxt.slot < nslots;
    if (more) {
      gather(nxt, id_next, v);
      const Tile after = next(nxt);
      if (after.slot < nslots) id_next = load_id(after);
    }
    for (int l = 0; l < a.nlayers; ++l) {
      const bool last = l == a.nlayers - 1;
      // two call sites so that each sees ONE address space: ds_read_b128 for LDS-resident weights, global_load for the others
      if (a.w_off[l] >= 0) sa_layer_any<PACK, MAXNB == 8>(a.cout[l] / 32, strip, CS, smem + a.w_off[l], smem + a.b_off[l], a.cin[l] / 8, last, lane, run);
      else sa_layer_any<PACK, MAXNB == 8>(a.cout[l] / 32, strip, CS, a.w[l], smem + a.b_off[l], a.cin[l] / 8, last, lane, run);
    }
    if (cur.kt == RT - 1 && lane < 32) {
#pragma unroll
      for (int p = 0; p < PACK; ++p) {
        const int g = cur.slot * PACK + p;
        if (g < G) {
          const int b = g / a.S, s = g - b * a.S;
#pragma unroll
          for (int q = 0; q < MAXNB; ++q)
            if (q * 32 < c_last) a.out[b * a.out_bs + s * a.out_ss + (q * 32 + lane) * a.out_cs] = run[q][p];
        }
      }
    }
    if (!more) break;
    cur = nxt; nxt = next(cur);
  }
}


// ================================================================ register-resident kernel (all widths <= 128)
// wp: fragment-packed weights Wp[nb][ks][lane][4] (+lane applied by the caller), element j of lane l = W[32nb + (l&31)][8ks + 4(l>>5) + j]
template <int NOUT>
__device__ __forceinline__ void sa_frags(const f32x4* wp, int nks, int ks, f32x4 (&f)[NOUT]) {
#pragma unroll
  for (int nb = 0; nb < NOUT; ++nb) f[nb] = wp[(size_t)(nb * nks + ks) * 64];
}

// accumulators of a hidden layer start as its bias: rows of D^T are channels 32nb + 8q + 4h + (r&3)
template <int NOUT>
__device__ __forceinline__ void sa_bias_init(const float* bias, int lhi, f32x16 (&out)[NOUT]) {
#pragma unroll
  for (int nb = 0; nb < NOUT; ++nb)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 b4 = *(const f32x4*)(bias + nb * 32 + q * 8 + lhi * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) out[nb][4 * q + j] = b4[j];
    }
}

template <int NOUT>
__device__ __forceinline__ void sa_relu(f32x16 (&a)[NOUT]) {
#pragma unroll
  for (int nb = 0; nb < NOUT; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) a[nb][r] = fmaxf(a[nb][r], 0.f);
}

// One layer over its input blocks, fully unrolled, as an explicit one-k-step-ahead software pipeline: the weight fragments of k-step
// ks+1 are requested, THEN the 4 x NOUT MFMAs of k-step ks issue (scheduling barriers keep that order and bound the live fragment
// sets to two -- left alone, the scheduler hoists every fragment load of the layer to its top and spills).
// HIDDEN: weights = A operand, activations = B (D^T orientation);  !HIDDEN (last layer): activations = A, weights = B.
template <int NIN, int NOUT, bool HIDDEN>
__device__ __forceinline__ void sa_layer_regs(const f32x16 (&in)[NIN], const f32x4* wp, f32x16 (&out)[NOUT]) {
  constexpr int NKS = NIN * 4;
  f32x4 wf[2][NOUT];
  sa_frags<NOUT>(wp, NKS, 0, wf[0]);
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    if (ks + 1 < NKS) sa_frags<NOUT>(wp, NKS, ks + 1, wf[(ks + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int nb = 0; nb < NOUT; ++nb) {
        const float act = in[ks >> 2][4 * (ks & 3) + j], wgt = wf[ks & 1][nb][j];
        out[nb] = HIDDEN ? mfma32(wgt, act, out[nb]) : mfma32(act, wgt, out[nb]);
      }
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <int NIN, int NOUT>
__device__ __forceinline__ void sa_hidden(const f32x16 (&in)[NIN], const float* w, const float* bias, int lane, f32x16 (&out)[NOUT]) {
  sa_bias_init<NOUT>(bias, lane >> 5, out);
  sa_layer_regs<NIN, NOUT, true>(in, (const f32x4*)w + lane, out);
  sa_relu<NOUT>(out);
}
// first layer as a hidden layer: the gathered rows x0 (2 k-steps: channels 4h + j and 8 + 4h + j) are the B operand
// `c0` = 3 + D real input channels (wave-uniform): MFMA j of k-step ks carries channels 8ks + j and 8ks + 4 + j, so with c0 = 9 the
// MFMAs (1, 1..3) multiply the zero padding only and are skipped (5 instead of 8 per output block: 6 of a 9-64-64-128 tile's 208)
template <int NOUT>
__device__ __forceinline__ void sa_hidden0(const float (&x0)[8], const float* w, const float* bias, int lane, int c0, f32x16 (&out)[NOUT]) {
  const f32x4* wp = (const f32x4*)w + lane;
  sa_bias_init<NOUT>(bias, lane >> 5, out);
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    if (8 * ks >= c0) break;
    f32x4 wf[NOUT];
    sa_frags<NOUT>(wp, 2, ks, wf);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (8 * ks + j >= c0) break;
#pragma unroll
      for (int nb = 0; nb < NOUT; ++nb) out[nb] = mfma32(wf[nb][j], x0[4 * ks + j], out[nb]);
    }
  }
  sa_relu<NOUT>(out);
}

// last layer: activations = A operand (rows = neighbours), weights = B; accumulate from 0, fold the rows of each neighbourhood into run
template <int NOUT, int PACK>
__device__ __forceinline__ void sa_last_fold(f32x16 (&c)[NOUT], float (*run)[PACK]) {
  constexpr int RP = 16 / PACK;       // accumulator registers per neighbourhood: r in [p*RP, (p+1)*RP) <-> rows [p*32/PACK, ...)
#pragma unroll
  for (int nb = 0; nb < NOUT; ++nb)
#pragma unroll
    for (int p = 0; p < PACK; ++p) {
      float m = c[nb][p * RP];
#pragma unroll
      for (int r = 1; r < RP; ++r) m = fmaxf(m, c[nb][p * RP + r]);
      m = fmaxf(m, __shfl_xor(m, 32));
      run[nb][p] = fmaxf(run[nb][p], m);
    }
}
template <int NIN, int NOUT, int PACK>
__device__ __forceinline__ void sa_last(const f32x16 (&in)[NIN], const float* w, int lane, float (*run)[PACK]) {
  f32x16 c[NOUT];
#pragma unroll
  for (int nb = 0; nb < NOUT; ++nb) c[nb] = f32x16{0};
  sa_layer_regs<NIN, NOUT, false>(in, (const f32x4*)w + lane, c);
  sa_last_fold<NOUT, PACK>(c, run);
}
template <int NOUT, int PACK>
__device__ __forceinline__ void sa_last0(const float (&x0)[8], const float* w, int lane, int c0, float (*run)[PACK]) {
  const f32x4* wp = (const f32x4*)w + lane;
  f32x16 c[NOUT];
#pragma unroll
  for (int nb = 0; nb < NOUT; ++nb) c[nb] = f32x16{0};
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    if (8 * ks >= c0) break;
    f32x4 wf[NOUT];
    sa_frags<NOUT>(wp, 2, ks, wf);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (8 * ks + j >= c0) break;
#pragma unroll
      for (int nb = 0; nb < NOUT; ++nb) c[nb] = mfma32(x0[4 * ks + j], wf[nb][j], c[nb]);
    }
  }
  sa_last_fold<NOUT, PACK>(c, run);
}

// The layer widths are template parameters (N_l = cout_l / 32, 0 = layer absent; up to 3 layers of width 32 / 64 / 128): the register
// arrays then have exactly the extents a network needs (9-64-64-128: 32 + 32 + 64 accumulator registers), which decides how many waves
// a SIMD holds.  NT = threads per workgroup, from the same estimate (sa_reg_threads).
constexpr int sa_reg_threads(int n0, int n1, int n2) {
  const int a = n0 + (n1 ? n1 : 0), b = n2 ? n1 + n2 : 0;
#ifndef SA_REG_PAD
#define SA_REG_PAD 72
#endif
  const int est = 16 * (a > b ? a : b) + SA_REG_PAD;    // live accumulator blocks of the widest layer pair + fragments, gather, addresses
  return est <= 124 ? 1024 : est <= 140 ? 768 : 512;
}

// Weights + biases of every layer -> LDS, once per workgroup.  ALL loads are issued before the first LDS store: written as a loop
// per layer the compiler waits for every load before its store (global_load; s_waitcnt vmcnt(0); ds_write in a rolled loop), ten
// dependent L2 round trips for 9-64-64-128 at 512 threads -- several microseconds at the head of every launch.  The layer sizes are
// template parameters, so a workgroup of >= 256 threads holds its whole share in registers (<= 13 x 16 B per thread).
template <int N0, int N1, int N2>
__device__ __forceinline__ void sa_stage(const SAArgs& a, float* smem) {
  constexpr int L = N2 ? 3 : N1 ? 2 : 1;
  constexpr int n4[3] = {C0 * N0 * 32 / 4, N0 * 32 * N1 * 32 / 4, N1 * 32 * N2 * 32 / 4};
  const int tid = threadIdx.x, bd = blockDim.x;
  if (bd >= 256) {
    constexpr int U0 = (n4[0] + 255) / 256, U1 = L > 1 ? (n4[1] + 255) / 256 : 1, U2 = L > 2 ? (n4[2] + 255) / 256 : 1;
    f32x4 r0[U0], r1[U1], r2[U2];
    float bq[3] = {0.f, 0.f, 0.f};
    const f32x4* s0 = (const f32x4*)a.w[0]; const f32x4* s1 = (const f32x4*)a.w[L > 1 ? 1 : 0]; const f32x4* s2 = (const f32x4*)a.w[L > 2 ? 2 : 0];
#pragma unroll
    for (int l = 0; l < L; ++l) if (tid < a.cout[l]) bq[l] = a.b[l][tid];       // widths <= 128 < 256 <= bd
#pragma unroll
    for (int u = 0; u < U0; ++u) if (tid + u * bd < n4[0]) r0[u] = s0[tid + u * bd];
    if constexpr (L > 1) {
#pragma unroll
      for (int u = 0; u < U1; ++u) if (tid + u * bd < n4[1]) r1[u] = s1[tid + u * bd];
    }
    if constexpr (L > 2) {
#pragma unroll
      for (int u = 0; u < U2; ++u) if (tid + u * bd < n4[2]) r2[u] = s2[tid + u * bd];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int l = 0; l < L; ++l) if (tid < a.cout[l]) smem[a.b_off[l] + tid] = bq[l];
#pragma unroll
    for (int u = 0; u < U0; ++u) if (tid + u * bd < n4[0]) ((f32x4*)(smem + a.w_off[0]))[tid + u * bd] = r0[u];
    if constexpr (L > 1) {
#pragma unroll
      for (int u = 0; u < U1; ++u) if (tid + u * bd < n4[1]) ((f32x4*)(smem + a.w_off[1]))[tid + u * bd] = r1[u];
    }
    if constexpr (L > 2) {
#pragma unroll
      for (int u = 0; u < U2; ++u) if (tid + u * bd < n4[2]) ((f32x4*)(smem + a.w_off[2]))[tid + u * bd] = r2[u];
    }
  } else {
#pragma unroll
    for (int l = 0; l < L; ++l) {
      const f32x4* src = (const f32x4*)a.w[l];
      f32x4* dst = (f32x4*)(smem + a.w_off[l]);
      for (int i = tid; i < n4[l]; i += bd) dst[i] = src[i];
      for (int i = tid; i < a.cout[l]; i += bd) smem[a.b_off[l] + i] = a.b[l][i];
    }
  }
}

template <int N0, int N1, int N2, int PACK, int NT>
__global__ __launch_bounds__(NT) void sa_reg_kernel(SAArgs a) {
  constexpr int L = N2 ? 3 : N1 ? 2 : 1;
  constexpr int NL = N2 ? N2 : N1 ? N1 : N0;             // channel blocks of the last layer
  const int WAVES = blockDim.x >> 6;                     // <= NT / 64: small launches use smaller workgroups to reach every CU
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int c0 = 3 + a.D;
  constexpr int RPN = 32 / PACK;                         // rows per neighbourhood
  const int G = a.B * a.S;                               // < 2^31 (checked by the launcher)
  const int nslots = (G + PACK - 1) / PACK;              // a slot = the PACK neighbourhoods of one tile
  const int RT = PACK == 1 ? (a.K + 31) / 32 : 1;        // row tiles per slot
  const int stride = gridDim.x * WAVES;
  constexpr int c_last = NL * 32;
  struct Tile { int slot; int kt; };
  auto next = [&](Tile t) { return t.kt + 1 < RT ? Tile{t.slot, t.kt + 1} : Tile{t.slot + stride, 0}; };
  Tile cur{(int)blockIdx.x * WAVES + wv, 0};
  const bool idle = cur.slot >= nslots;                  // still helps staging the weights
  if (idle) cur.slot = nslots - 1;
  auto nbhd_of = [&](Tile t) -> int {
    const int g = t.slot * PACK + l31 / RPN;
    return g < G ? g : G - 1;                            // tail slot: duplicate the last neighbourhood (its result is not stored)
  };
  auto load_id = [&](Tile t) -> long long {
    const int kk = PACK == 1 ? t.kt * 32 + l31 : l31 % RPN;
    return a.idx[(size_t)nbhd_of(t) * a.K + (kk < a.K ? kk : 0)];
  };
  // lane (neighbour row r = l31, half h) gathers the operand elements of the two first-layer k-steps: channels 4h + j and 8 + 4h + j
  // of centred xyz ++ features (zero padded to 16).  In two halves so that the loads stay in flight across the layers of the
  // current tile: `issue` only computes (clamped, always valid) addresses and loads; `finish` -- run AFTER the layers -- applies
  // the channel layout with selects.  (A select right behind its load makes the wave wait for L2 at the top of every tile.)
  struct Raw { float f[8]; float p[3]; float c[3]; };
  auto issue = [&](Tile t, long long id, Raw& r) {
    const int g = nbhd_of(t);
    const bool bad = id < 0 || id >= a.N;                                        // index_points raises on such an index
    if (bad && a.err_flag) *a.err_flag = 1;
    id = bad ? 0 : id;
    const int b = g / a.S;
    const float* px = a.xyz + ((size_t)b * a.N + id) * 3;
    const float* pc = a.new_xyz + (size_t)g * 3;
#pragma unroll
    for (int e = 0; e < 3; ++e) { r.p[e] = px[e]; r.c[e] = pc[e]; }
    if (a.D > 0) {                                                               // wave-uniform
      const float* pf = a.points + ((size_t)b * a.N + id) * a.D;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int f = (e >> 2) * 8 + 4 * lhi + (e & 3) - 3;                      // feature index of this element (negative: a coordinate)
        r.f[e] = pf[f < 0 ? 0 : (f < a.D ? f : a.D - 1)];
      }
    }
  };
  auto finish = [&](const Raw& r, float (&v)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int f = (e >> 2) * 8 + 4 * lhi + (e & 3) - 3;
      v[e] = (a.D > 0 && f >= 0 && f < a.D) ? r.f[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) v[e] = lhi ? v[e] : r.p[e] - r.c[e];
  };
  const float* W0 = smem + a.w_off[0]; const float* W1 = smem + a.w_off[1]; const float* W2 = smem + a.w_off[2];
  const float* B0 = smem + a.b_off[0]; const float* B1 = smem + a.b_off[1];

  float run[NL][PACK];
  float x0[8];
  Raw raw;
  // the first tile's gather (index -> point: two dependent L2 / HBM round trips) goes out BEFORE the weights are staged, and the
  // indices of the second tile with it: both are in flight while the workgroup copies its weights
  long long id0 = load_id(cur);
  Tile nxt = next(cur);
  long long id_next = nxt.slot < nslots ? load_id(nxt) : 0;
  issue(cur, id0, raw);
  __builtin_amdgcn_sched_barrier(0);
  sa_stage<N0, N1, N2>(a, smem);
  __syncthreads();
  if (idle) return;
  finish(raw, x0);
  for (;;) {
    if (cur.kt == 0) {
#pragma unroll
      for (int q = 0; q < NL; ++q)
#pragma unroll
        for (int p = 0; p < PACK; ++p) run[q][p] = -INFINITY;
    }
    // next tile's rows (their indices arrived during the previous tile) and the indices of the tile after: in flight during the layers
    // (Unconditional: past the end the loads are repeated on valid addresses and never used -- a conditional load would meet the old
    // value in a phi, and the register copies behind that phi make the wave wait for the loads right here.)
    const bool more = nxt.slot < nslots;
    issue(more ? nxt : cur, more ? id_next : 0, raw);
    Tile after = next(nxt);
    if (after.slot >= nslots) after = cur;
    id_next = load_id(after);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (L == 1) {
      sa_last0<N0, PACK>(x0, W0, lane, c0, run);
    } else {
      f32x16 A[N0];
      sa_hidden0<N0>(x0, W0, B0, lane, c0, A);
      if constexpr (L == 2) {
        sa_last<N0, N1, PACK>(A, W1, lane, run);
      } else {
        f32x16 Bf[N1];
        sa_hidden<N0, N1>(A, W1, B1, lane, Bf);
        sa_last<N1, N2, PACK>(Bf, W2, lane, run);
      }
    }
    if (cur.kt == RT - 1 && lane < 32) {
      const float* bl = smem + a.b_off[L - 1];
#pragma unroll
      for (int p = 0; p < PACK; ++p) {
        const int g = cur.slot * PACK + p;
        if (g < G) {
          const int b = g / a.S, s = g - b * a.S;
#pragma unroll
          for (int q = 0; q < NL; ++q)                     // bias + ReLU after the max: both commute with it (see the header)
            a.out[b * a.out_bs + s * a.out_ss + (q * 32 + lane) * a.out_cs] = fmaxf(run[q][p] + bl[q * 32 + lane], 0.f);
        }
      }
    }
    if (!more) break;
    __builtin_amdgcn_sched_barrier(0);
    finish(raw, x0);                                       // the loads were issued a whole tile ago
    cur = nxt; nxt = next(cur);
  }
}

template <int N0, int N1, int N2, int PACK>
int launch_sa_reg(SAArgs& a, hipStream_t s, int dev) {
  constexpr int NT = sa_reg_threads(N0, N1, N2);
  constexpr int WAVES = NT / 64;                         // most waves per workgroup the register budget allows
  int off = 0;
  for (int l = 0; l < SA_MAX_LAYERS; ++l) { a.b_off[l] = 0; a.w_off[l] = 0; }
  for (int l = 0; l < a.nlayers; ++l) { a.b_off[l] = off; off += a.cout[l]; }
  off = (off + 3) & ~3;
  for (int l = 0; l < a.nlayers; ++l) { a.w_off[l] = off; off += a.cin[l] * a.cout[l]; }
  a.wb_floats = off;
  const size_t lds = (size_t)off * 4;                    // <= (16 + 128 + 128) * 128 floats + biases = 140 KB: always fits
  const int n_cu = cg_device_cu_count(dev);
  if (n_cu <= 0 || dev < 0 || dev >= CG_MAX_DEVICES || lds > 158 * 1024) return CG_ERR_UNSUPPORTED;
  const long nslots = ((long)a.B * a.S + PACK - 1) / PACK;
  // Persistent grid, one workgroup per CU (registers: NT threads fill a CU exactly once).  Waves per workgroup: the most the register
  // budget allows, unless fewer waves balance the slots better over the chip (a wave's slots are one dependent chain each: with
  // 16,384 slots 8 waves x 256 CUs take 8 slots each, 12 waves would take 5 or 6) or the launch is too small to give every CU one.
  int waves = WAVES;
  double best = -1.0;
  for (int w = WAVES; w >= 1; --w) {
    const long rounds = (nslots + (long)n_cu * w - 1) / ((long)n_cu * w);
    const double eff = (double)nslots / ((double)rounds * n_cu * w);
    if (eff > best + 0.03) { best = eff; waves = w; }
  }
  long grid = (nslots + waves - 1) / waves;
  if (grid > n_cu) grid = n_cu;
  auto kern = sa_reg_kernel<N0, N1, N2, PACK, NT>;
  static bool attr_set[CG_MAX_DEVICES] = {};
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * waves), lds, s, a);
  return cg_hip_status(hipGetLastError());
}

// dispatch over the supported signatures: up to three layers, every width 32 / 64 / 128
template <int N0, int N1, int N2>
int launch_sa_reg_pack(SAArgs& a, hipStream_t s, int dev) {
  if (a.K <= 8) return launch_sa_reg<N0, N1, N2, 4>(a, s, dev);
  if (a.K <= 16) return launch_sa_reg<N0, N1, N2, 2>(a, s, dev);
  return launch_sa_reg<N0, N1, N2, 1>(a, s, dev);
}
template <int N0, int N1>
int launch_sa_reg_2(SAArgs& a, int n2, hipStream_t s, int dev) {
  switch (n2) {
    case 0: return launch_sa_reg_pack<N0, N1, 0>(a, s, dev);
    case 1: return launch_sa_reg_pack<N0, N1, 1>(a, s, dev);
    case 2: return launch_sa_reg_pack<N0, N1, 2>(a, s, dev);
    case 4: return launch_sa_reg_pack<N0, N1, 4>(a, s, dev);
  }
  return CG_ERR_UNSUPPORTED;
}
template <int N0>
int launch_sa_reg_1(SAArgs& a, int n1, int n2, hipStream_t s, int dev) {
  switch (n1) {
    case 0: return n2 == 0 ? launch_sa_reg_pack<N0, 0, 0>(a, s, dev) : CG_ERR_UNSUPPORTED;
    case 1: return launch_sa_reg_2<N0, 1>(a, n2, s, dev);
    case 2: return launch_sa_reg_2<N0, 2>(a, n2, s, dev);
    case 4: return launch_sa_reg_2<N0, 4>(a, n2, s, dev);
  }
  return CG_ERR_UNSUPPORTED;
}
// -> CG_ERR_UNSUPPORTED when the network is outside the register kernel's signatures (4 layers, a width of 96, ...): strip kernel
int launch_sa_reg_any(SAArgs& a, hipStream_t s, int dev) {
  if (a.nlayers > 3) return CG_ERR_UNSUPPORTED;
  const int n0 = a.cout[0] / 32, n1 = a.nlayers > 1 ? a.cout[1] / 32 : 0, n2 = a.nlayers > 2 ? a.cout[2] / 32 : 0;
  switch (n0) {
    case 1: return launch_sa_reg_1<1>(a, n1, n2, s, dev);
    case 2: return launch_sa_reg_1<2>(a, n1, n2, s, dev);
    case 4: return launch_sa_reg_1<4>(a, n1, n2, s, dev);
  }
  return CG_ERR_UNSUPPORTED;
}

constexpr size_t SA_LDS_BUDGET = 158 * 1024;      // of the CU's 160 KB

template <int MAXNB, int PACK>
int launch_sa(SAArgs& a, int cs, hipStream_t s, int dev) {
  const size_t per_wave = (size_t)32 * cs * 4;
  // Weight residency plan: biases always; a layer's weights go to LDS if, with them, at least 4 waves' strips still fit.  Layers are
  // taken in order of matrix work (cin*cout = bytes, so simply largest first until the budget is spent).
  int off = 0;
  for (int l = 0; l < a.nlayers; ++l) { a.b_off[l] = off; off += a.cout[l]; a.w_off[l] = -1; }
  off = (off + 3) & ~3;
  bool taken[SA_MAX_LAYERS] = {};
  for (int round = 0; round < a.nlayers; ++round) {
    int best = -1;
    for (int l = 0; l < a.nlayers; ++l)
      if (!taken[l] && (best < 0 || a.cin[l] * a.cout[l] > a.cin[best] * a.cout[best])) best = l;
    taken[best] = true;
    const size_t need = (size_t)(off + a.cin[best] * a.cout[best]) * 4 + 4 * per_wave;
    if (need <= SA_LDS_BUDGET) { a.w_off[best] = off; off += a.cin[best] * a.cout[best]; }
  }
  a.wb_floats = off;
  const size_t wb = (size_t)off * 4;
  if (wb + per_wave > SA_LDS_BUDGET) return CG_ERR_UNSUPPORTED;
  int waves = (int)((SA_LDS_BUDGET - wb) / per_wave);
  if (waves > 8) waves = 8;
  const int n_cu = cg_device_cu_count(dev);
  if (n_cu <= 0) return CG_ERR_UNSUPPORTED;
  const long nslots = ((long)a.B * a.S + PACK - 1) / PACK;
  // small launches: fewer waves per workgroup so that every CU gets one (a neighbourhood is one dependent chain: spread first)
  if (nslots < (long)n_cu * waves) { waves = (int)((nslots + n_cu - 1) / n_cu); if (waves < 1) waves = 1; }
  const size_t lds = wb + per_wave * waves;
  int wg_per_cu = (int)(SA_LDS_BUDGET / lds);
  if (wg_per_cu * waves > 8) wg_per_cu = 8 / waves > 0 ? 8 / waves : 1;        // registers: two waves per SIMD
  if (wg_per_cu < 1) wg_per_cu = 1;
  long grid = (nslots + waves - 1) / waves;
  if (grid > (long)n_cu * wg_per_cu) grid = (long)n_cu * wg_per_cu;              // persistent: the waves loop over the slots
  auto kern = sa_group_mlp_max_kernel<MAXNB, PACK>;
  static bool attr_set[CG_MAX_DEVICES] = {};
  if (dev < 0 || dev >= CG_MAX_DEVICES) return CG_ERR_UNSUPPORTED;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * waves), lds, s, a, cs, waves);
  return cg_hip_status(hipGetLastError());
}

template <int MAXNB>
int launch_sa_pack(SAArgs& a, int cs, hipStream_t s, int dev) {
  if (a.K <= 8) return launch_sa<MAXNB, 4>(a, cs, s, dev);
  if (a.K <= 16) return launch_sa<MAXNB, 2>(a, cs, s, dev);
  return launch_sa<MAXNB, 1>(a, cs, s, dev);
}

}  // namespace

extern "C" int cg_sa_group_mlp_max_strided(const float* xyz, const float* points, const float* new_xyz, const long long* idx, int B, int N,
                                           int S, int K, int D, int n_layers, const int* h_cin, const int* h_cout,
                                           const float* const* h_w_packed, const float* const* h_bias, float* out, long out_bs, long out_ss,
                                           long out_cs, int* err_flag, void* stream) {
  if (B < 0 || N <= 0 || S < 0 || K <= 0 || D < 0 || n_layers < 1 || n_layers > SA_MAX_LAYERS) return CG_ERR_ARG;
  if (!h_cin || !h_cout || !h_w_packed || !h_bias) return CG_ERR_ARG;
  if ((long)B * S == 0) return CG_OK;
  if (!xyz || !new_xyz || !idx || !out || (D > 0 && !points)) return CG_ERR_ARG;
  if (3 + D > C0 || (long)B * S >= 0x7fffffffL / 4) return CG_ERR_UNSUPPORTED;
  SAArgs a{};
  a.xyz = xyz; a.points = D > 0 ? points : nullptr; a.new_xyz = new_xyz; a.idx = idx;
  a.B = B; a.N = N; a.S = S; a.K = K; a.D = D; a.nlayers = n_layers; a.out = out; a.err_flag = err_flag;
  a.out_bs = out_bs; a.out_ss = out_ss; a.out_cs = out_cs;
  int cmax = 0, cstore = C0;
  for (int l = 0; l < n_layers; ++l) {
    if (!h_w_packed[l] || !h_bias[l]) return CG_ERR_ARG;
    if (h_cin[l] != (l == 0 ? C0 : h_cout[l - 1])) return CG_ERR_ARG;          // layer 0: K padded to 16 by the host
    if (h_cout[l] <= 0 || (h_cout[l] % 32) != 0) return CG_ERR_UNSUPPORTED;
    if (((uintptr_t)h_w_packed[l] & 15) != 0) return CG_ERR_ARG;
    a.cin[l] = h_cin[l]; a.cout[l] = h_cout[l]; a.w[l] = h_w_packed[l]; a.b[l] = h_bias[l];
    if (h_cout[l] > cmax) cmax = h_cout[l];
    if (l + 1 < n_layers && h_cout[l] > cstore) cstore = h_cout[l];
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return CG_ERR_UNSUPPORTED;
  if (cmax > 256) return CG_ERR_UNSUPPORTED;
  if (cmax <= 128) {
    const int st = launch_sa_reg_any(a, (hipStream_t)stream, dev);
    if (st != CG_ERR_UNSUPPORTED) return st;             // outside the register kernel's signatures (4 layers, a width of 96): strip kernel
  }
  if (cmax <= 128) return launch_sa_pack<4>(a, cstore + 4, (hipStream_t)stream, dev);
  return launch_sa_pack<8>(a, cstore + 4, (hipStream_t)stream, dev);
}

// (B, C_last, S) output: the layout of torch.max(new_points, 2)[0]
extern "C" int cg_sa_group_mlp_max(const float* xyz, const float* points, const float* new_xyz, const long long* idx, int B, int N, int S,
                                   int K, int D, int n_layers, const int* h_cin, const int* h_cout, const float* const* h_w_packed,
                                   const float* const* h_bias, float* out, int* err_flag, void* stream) {
  if (n_layers < 1 || n_layers > SA_MAX_LAYERS || !h_cout) return CG_ERR_ARG;
  const long c_last = h_cout[n_layers - 1];
  return cg_sa_group_mlp_max_strided(xyz, points, new_xyz, idx, B, N, S, K, D, n_layers, h_cin, h_cout, h_w_packed, h_bias, out, c_last * S, 1, S,
                                     err_flag, stream);
}
