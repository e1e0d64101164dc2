// Split-precision variant of the fused per-point MLP chain + max-pool (see pointmlp.hip for the op sequence it
// replaces: pointnet2.py:172-176, :210-214, :243-266).
//
// Every K>=64 contraction is evaluated as three bf16 MFMAs with f32 accumulation ("bf16x3"):
//     x = x_hi + x_lo,  w = w_hi + w_lo  (bf16 each, round-to-nearest-even; lo = bf16(x - x_hi))
//     x.w ~= x_hi.w_hi + x_lo.w_hi + x_hi.w_lo          (dropped term x_lo.w_lo <= 2^-16 |x.w|)
// on v_mfma_f32_32x32x16_bf16 (16x the f32-MFMA rate, so 16/3 = 5.3x per contraction).  Measured end-to-end
// error on the grasp-Q logits is ~1e-5 (tests/test_pointnet_gpu.py), inside the 1e-4 parity bar.
//
// Layout: one workgroup = 8 waves owns one sample (or a slice of its point tiles).  A tile of 256 points is
// carried through 6->64 (f32 VALU) -> [64->64] -> 64->128 -> 128->1024 inside LDS (148 KB, one workgroup per CU;
// a 128-point / 4-wave geometry with two workgroups per CU is also instantiated).
//  * Front layers are WAVE-PRIVATE: wave w takes rows [32w, 32w+32) of the tile through the whole chain with no
//    workgroup barrier.  Its f32 scratch ([32][68] floats twice) aliases exactly its own 32 rows of the two
//    bf16 images of the 128-wide activation (32 rows x 272 B == 32 x 68 floats), which it overwrites last.
//  * The 128-wide activation lives in LDS split into bf16 hi / lo images ([256][136] each; row stride 272 B makes
//    the ds_read_b128 fragment reads conflict-free).
//  * In the 128->1024 layer wave w owns channel blocks [4w,4w+4) and ALL 8 row tiles, so each packed weight
//    fragment is fetched from L2 exactly once per workgroup tile (10.7 B/clk/CU at full MFMA rate), and the max
//    over points is a per-lane reduction over accumulator registers + one lane^32 swap.
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SH = 136;        // bf16 elements per row of the h2 hi / lo images
constexpr int S64 = 68;        // floats per row of the f32 scratch tiles
constexpr int XS = 8;
// Two geometries: RT = 8 row tiles (256 points, 8 waves, 148 KB LDS, one workgroup per CU; the default) and RT = 4
// (128 points, 4 waves, 78 KB, two workgroups per CU; measured 7 % slower: twice the weight traffic).  One wave per 32-row tile.
template <int RT> struct Geo {
  static constexpr int TP = 32 * RT;
  static constexpr int NT = 64 * RT;
  static constexpr int NBW = 32 / RT;       // 32-channel blocks of the 1024-wide layer owned by each wave
  static constexpr size_t LDS_BYTES = (size_t)2 * TP * SH * 2 + 1024 * 4 + (size_t)TP * XS * 4;
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
  float* out; float* pointfeat;
};

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 h0 = (__bf16)v0[e]; hi[e] = h0; lo[e] = (__bf16)(v0[e] - (float)h0);
    const __bf16 h1 = (__bf16)v1[e]; hi[4 + e] = h1; lo[4 + e] = (__bf16)(v1[e] - (float)h1);
  }
}

// packed split weights: Wp[nb][kc][2 (hi,lo)][lane][8] bf16
__device__ __forceinline__ void load_b(const unsigned short* wp, int nb, int kc, int nkc, int lane, bf16x8& bhi, bf16x8& blo) {
  const bf16x8* p = (const bf16x8*)wp + ((size_t)(nb * nkc + kc) * 2) * 64 + lane;
  bhi = p[0]; blo = p[64];
}

// LDS accesses of one wave execute in order; this only stops the COMPILER from reordering the aliased
// (float scratch vs bf16 image) accesses across a phase boundary.
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// An opaque zero: added to the (tile-invariant) front-layer weight pointers inside the tile loop so the compiler
// does not hoist 48 KB of weight-fragment loads out of the loop and spill them.
__device__ __forceinline__ int opaque_zero() { int z; asm volatile("s_mov_b32 %0, 0" : "=s"(z)); return z; }

template <int MID, int RT>
__global__ __launch_bounds__(64 * RT, 2) void pointmlp_max_bf16x3_kernel(ArgsB a) {
  constexpr int TP = Geo<RT>::TP, NT = Geo<RT>::NT, NBW = Geo<RT>::NBW;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __bf16* h2hi = (__bf16*)smem_raw;
  __bf16* h2lo = h2hi + TP * SH;
  float* rmax = (float*)(h2lo + TP * SH);
  float* xs = rmax + 1024;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int b = blockIdx.x / a.nsplit;
  const int split = blockIdx.x - b * a.nsplit;
  const int ntiles = (a.N + TP - 1) / TP;
  const int t_begin = (int)(((long)ntiles * split) / a.nsplit);
  const int t_end = (int)(((long)ntiles * (split + 1)) / a.nsplit);

  // wave-private views: f32 scratch aliasing this wave's 32 rows of the hi / lo images, and its staged points
  float* hA = (float*)(h2hi + w * 32 * SH);
  float* hB = (float*)(h2lo + w * 32 * SH);
  float* xw = xs + w * 32 * XS;

  for (int i = tid; i < 1024; i += NT) rmax[i] = -INFINITY;

  float w1r[6], b1r;     // first layer: lane = output channel
#pragma unroll
  for (int j = 0; j < 6; ++j) w1r[j] = a.w1[lane * 6 + j];
  b1r = a.b1[lane];
  float t3r[9];
  if (a.t3) {
#pragma unroll
    for (int j = 0; j < 9; ++j) t3r[j] = a.t3[b * 9 + j];
  }
  const float* xb = a.x + (size_t)b * a.N * 6;

  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();   // the previous tile's L3 reads of the h2 images (and the rmax init) are complete
#ifdef ABL_NO_FRONT
    if (tile == t_begin)
#endif
    {
    // ================= front layers, wave-private: rows [32w, 32w+32) =================
    const int oz = opaque_zero();
    const unsigned short* wm_t = a.wm + oz;
    const unsigned short* w2_t = a.w2 + oz;
    const float* t64_t = a.t64 + oz;
    if (lane < 32) {
      int p = tile * TP + w * 32 + lane;
      if (p >= a.N) p = a.N - 1;          // replicate the last point: max-pool is idempotent
      const f32x2* src = (const f32x2*)(xb + (size_t)p * 6);
      f32x2 v0 = src[0], v1 = src[1], v2 = src[2];
      float px = v0[0], py = v0[1], pz = v1[0];
      if (a.t3) {
        const float qx = px * t3r[0] + py * t3r[3] + pz * t3r[6];
        const float qy = px * t3r[1] + py * t3r[4] + pz * t3r[7];
        const float qz = px * t3r[2] + py * t3r[5] + pz * t3r[8];
        px = qx; py = qy; pz = qz;
      }
      *(f32x4*)(xw + lane * XS) = f32x4{px, py, pz, v1[1]};
      *(f32x4*)(xw + lane * XS + 4) = f32x4{v2[0], v2[1], 0.f, 0.f};
    }
    wave_lds_fence();
    {  // L0: 6 -> 64 on the VALU, lane = channel, loop over the wave's 32 points (broadcast LDS reads)
      float* dst = (MID == 0) ? hB : hA;
#pragma unroll 8
      for (int p = 0; p < 32; ++p) {
        const f32x4 q0 = *(const f32x4*)(xw + p * XS);
        const f32x2 q1 = *(const f32x2*)(xw + p * XS + 4);
        float v = b1r;
        v = fmaf(w1r[0], q0[0], v); v = fmaf(w1r[1], q0[1], v); v = fmaf(w1r[2], q0[2], v);
        v = fmaf(w1r[3], q0[3], v); v = fmaf(w1r[4], q1[0], v); v = fmaf(w1r[5], q1[1], v);
        dst[p * S64 + lane] = fmaxf(v, 0.f);
      }
    }
    wave_lds_fence();
    if (MID != 0) {  // mid: 64 -> 64 (shared conv+BN+ReLU, or the per-sample 64x64 feature transform)
      f32x16 c0 = {0}, c1 = {0};
      const float* arow = hA + l31 * S64 + lhi * 8;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) {
        bf16x8 ahi, alo, b0h, b0l, b1h, b1l;
        split8(*(const f32x4*)(arow + kc * 16), *(const f32x4*)(arow + kc * 16 + 4), ahi, alo);
        if (MID == 1) {
          load_b(wm_t, 0, kc, 4, lane, b0h, b0l);
          load_b(wm_t, 1, kc, 4, lane, b1h, b1l);
        } else {
          // t64 is stored TRANSPOSED (Tt[n][k] = T[k][n]): a lane's 8 consecutive k are two 16-byte loads
          const float* tp = t64_t + (size_t)b * 4096 + l31 * 64 + kc * 16 + lhi * 8;
          split8(*(const f32x4*)tp, *(const f32x4*)(tp + 4), b0h, b0l);
          tp += 32 * 64;
          split8(*(const f32x4*)tp, *(const f32x4*)(tp + 4), b1h, b1l);
        }
        c0 = mfma_bf16(alo, b0h, c0); c1 = mfma_bf16(alo, b1h, c1);
        c0 = mfma_bf16(ahi, b0l, c0); c1 = mfma_bf16(ahi, b1l, c1);
        c0 = mfma_bf16(ahi, b0h, c0); c1 = mfma_bf16(ahi, b1h, c1);
      }
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int col = nb * 32 + l31;
        const float bias = (MID == 1) ? a.bm[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = acc_row(r, lane);
          float v = (nb ? c1[r] : c0[r]) + bias;
          if (MID == 1) v = fmaxf(v, 0.f);
          hB[row * S64 + col] = v;
          if (MID == 2 && a.pointfeat) {
            const int p = tile * TP + w * 32 + row;
            if (p < a.N) a.pointfeat[((size_t)b * a.N + p) * 64 + col] = v;
          }
        }
      }
      wave_lds_fence();
    }
    {  // L2: 64 -> 128.  All A fragments are pulled into registers first: the result overwrites the scratch rows.
      bf16x8 ah[4], al[4];
      const float* arow = hB + l31 * S64 + lhi * 8;
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) split8(*(const f32x4*)(arow + kc * 16), *(const f32x4*)(arow + kc * 16 + 4), ah[kc], al[kc]);
      wave_lds_fence();
#pragma unroll
      for (int np = 0; np < 2; ++np) {       // two channel blocks at a time
        f32x16 c0 = {0}, c1 = {0};
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          bf16x8 b0h, b0l, b1h, b1l;
          load_b(w2_t, np * 2, kc, 4, lane, b0h, b0l);
          load_b(w2_t, np * 2 + 1, kc, 4, lane, b1h, b1l);
          c0 = mfma_bf16(al[kc], b0h, c0); c1 = mfma_bf16(al[kc], b1h, c1);
          c0 = mfma_bf16(ah[kc], b0l, c0); c1 = mfma_bf16(ah[kc], b1l, c1);
          c0 = mfma_bf16(ah[kc], b0h, c0); c1 = mfma_bf16(ah[kc], b1h, c1);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int col = (np * 2 + h) * 32 + l31;
          const float bias = a.b2[col];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = w * 32 + acc_row(r, lane);
            const float v = fmaxf((h ? c1[r] : c0[r]) + bias, 0.f);
            const __bf16 hv = (__bf16)v;
            h2hi[row * SH + col] = hv;
            h2lo[row * SH + col] = (__bf16)(v - (float)hv);
          }
        }
      }
    }
    }
    __syncthreads();
#ifndef ABL_NO_L3
    // ================= L3: 128 -> 1024 + running max.  wave w owns channel blocks [4w, 4w+4) =================
    // Two-stage software pipeline per 16-deep k chunk: while the 12 MFMAs of one half of the row tiles run, the LDS
    // reads of the other half (and the L2 weight fetch of the next chunk) are in flight.
    {
      constexpr int G = 2;                 // row tiles per pipeline stage
      constexpr int NST = RT / G;          // stages per k chunk
      const __bf16* ahi_base = h2hi + l31 * SH + lhi * 8;
      const __bf16* alo_base = h2lo + l31 * SH + lhi * 8;
      for (int q = 0; q < NBW; ++q) {
        const int nb = w * NBW + q;
        f32x16 c[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) c[rt] = f32x16{0};
        bf16x8 bh[2], bl[2], sh[2][G], sl[2][G];
        load_b(a.w3, nb, 0, 8, lane, bh[0], bl[0]);
#pragma unroll
        for (int r = 0; r < G; ++r) {
          sh[0][r] = *(const bf16x8*)(ahi_base + r * 32 * SH);
          sl[0][r] = *(const bf16x8*)(alo_base + r * 32 * SH);
        }
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
          const int cur = kc & 1, nxt = cur ^ 1;
          if (kc < 7) load_b(a.w3, nb, kc + 1, 8, lane, bh[nxt], bl[nxt]);
#pragma unroll
          for (int st = 0; st < NST; ++st) {
            const int sc = st & 1, sn = sc ^ 1;
            // prefetch the next stage's A fragments (next row-tile group, or the first group of the next k chunk)
            const int nst = (st + 1) % NST, nkc = (st + 1 == NST) ? kc + 1 : kc;
            if (nkc < 8) {
#pragma unroll
              for (int r = 0; r < G; ++r) {
                sh[sn][r] = *(const bf16x8*)(ahi_base + (nst * G + r) * 32 * SH + nkc * 16);
                sl[sn][r] = *(const bf16x8*)(alo_base + (nst * G + r) * 32 * SH + nkc * 16);
              }
            }
#pragma unroll
            for (int r = 0; r < G; ++r) c[st * G + r] = mfma_bf16(sl[sc][r], bh[cur], c[st * G + r]);
#pragma unroll
            for (int r = 0; r < G; ++r) c[st * G + r] = mfma_bf16(sh[sc][r], bl[cur], c[st * G + r]);
#pragma unroll
            for (int r = 0; r < G; ++r) c[st * G + r] = mfma_bf16(sh[sc][r], bh[cur], c[st * G + r]);
            // interleave: M M D M D M D M D M  (6 MFMAs of this stage, 4 LDS fragment reads of the next)
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          }
        }
        float m = max16(c[0]);
#pragma unroll
        for (int rt = 1; rt < RT; ++rt) m = fmaxf(m, max16(c[rt]));
        m = fmaxf(m, __shfl_xor(m, 32));
        if (lane < 32) {
          const int ch = nb * 32 + lane;
          rmax[ch] = fmaxf(rmax[ch], m);
        }
      }
    }
#endif
  }
  __syncthreads();
  if (t_end > t_begin) {
    for (int ch = tid; ch < 1024; ch += NT) {
      float v = rmax[ch] + a.b3[ch];
      if (a.relu3) v = fmaxf(v, 0.f);
      if (a.nsplit == 1) a.out[(size_t)b * 1024 + ch] = v;
      else atomic_max_f32(a.out + (size_t)b * 1024 + ch, v);
    }
  }
}

__global__ void fill_kernel_b(float* p, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

template <int MID, int RT>
int launch(const ArgsB& a, hipStream_t s) {
  constexpr int NT = Geo<RT>::NT;
  constexpr size_t LDS_BYTES = Geo<RT>::LDS_BYTES;
  auto kern = pointmlp_max_bf16x3_kernel<MID, RT>;
  static bool attr_set = false;     // per instantiation
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * a.nsplit)), dim3(NT), LDS_BYTES, s, a);
  return cg_hip_status(hipGetLastError());
}

}  // namespace

extern "C" int cg_pointmlp_max_bf16x3(const float* x, int B, int N, const float* t3, const float* w1, const float* b1,
                                      int mid_mode, const unsigned short* wm_split, const float* bm, const float* t64,
                                      const unsigned short* w2_split, const float* b2, const unsigned short* w3_split,
                                      const float* b3, int relu3, int nsplit, int tile_points, float* out, float* pointfeat,
                                      void* stream) {
  if (B < 0 || N <= 0 || mid_mode < 0 || mid_mode > 2 || (tile_points != 128 && tile_points != 256)) return CG_ERR_ARG;
  if (B == 0) return CG_OK;
  if (!x || !w1 || !b1 || !w2_split || !b2 || !w3_split || !b3 || !out) return CG_ERR_ARG;
  if (mid_mode == 1 && (!wm_split || !bm)) return CG_ERR_ARG;
  if (mid_mode == 2 && !t64) return CG_ERR_ARG;
  if (pointfeat && mid_mode != 2) return CG_ERR_ARG;
  const int ntiles = (N + tile_points - 1) / tile_points;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > ntiles) nsplit = ntiles;
  hipStream_t s = (hipStream_t)stream;
  if (nsplit > 1) {
    const size_t n = (size_t)B * 1024;
    hipLaunchKernelGGL(fill_kernel_b, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, n, -INFINITY);
  }
  ArgsB a{x, B, N, t3, w1, b1, wm_split, bm, t64, w2_split, b2, w3_split, b3, relu3, nsplit, out, pointfeat};
  if (tile_points == 256) {
    if (mid_mode == 0) return launch<0, 8>(a, s);
    if (mid_mode == 1) return launch<1, 8>(a, s);
    return launch<2, 8>(a, s);
  }
  if (mid_mode == 0) return launch<0, 4>(a, s);
  if (mid_mode == 1) return launch<1, 4>(a, s);
  return launch<2, 4>(a, s);
}
