// Split-precision variant of the fused per-point MLP chain + max-pool (see pointmlp.hip for the op sequence it
// replaces: pointnet2.py:172-176, :210-214, :243-266).
//
// Every K>=64 contraction is evaluated as three bf16 MFMAs with f32 accumulation ("bf16x3"):
//     x = x_hi + x_lo,  w = w_hi + w_lo  (bf16 each, round-to-nearest-even; lo = bf16(x - x_hi))
//     x.w ~= x_hi.w_hi + x_lo.w_hi + x_hi.w_lo          (dropped term x_lo.w_lo <= 2^-16 |x.w|)
// on v_mfma_f32_32x32x16_bf16 (16x the f32-MFMA rate, so 16/3 = 5.3x per contraction).  Measured end-to-end
// error on the grasp-Q logits is ~1e-5 (tests/test_pointnet_gpu.py), inside the 1e-4 parity bar; the exact-f32
// kernel in pointmlp.hip remains the default.
//
// Layout: one workgroup = 8 waves owns one sample (or a slice of its point tiles).  A tile of TP = 32*RT points
// is carried through 6->64 (f32 VALU) -> [64->64] -> 64->128 -> 128->1024 inside LDS.  The 128-wide activation
// lives in LDS already split into bf16 hi / lo images ([TP][136] each: row stride 272 B = conflict-free
// ds_read_b128); the front layers run on 64-point sub-tiles whose f32 scratch aliases the last 64 rows of those
// images (64 rows x 272 B == 64 x 68 floats).  In the 128->1024 layer wave w owns channel blocks [4w,4w+4) and
// ALL RT row tiles, so each packed weight fragment is fetched from L2 exactly once per workgroup tile.
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SH = 136;   // bf16 elements per row of the h2 hi / lo images
constexpr int S64 = 68;   // floats per row of the f32 scratch tiles
constexpr int XS = 8;
constexpr int NT = 512;   // threads per workgroup (8 waves)

struct ArgsB {
  const float* x; int B; int N;
  const float* t3;
  const float* w1; const float* b1;
  const unsigned short* wm; const float* bm;     // split-packed 64->64 (MID==1)
  const float* t64;                              // (B,64,64) f32 (MID==2)
  const unsigned short* w2; const float* b2;     // split-packed 64->128
  const unsigned short* w3; const float* b3;     // split-packed 128->1024
  int relu3; int nsplit;
  float* out; float* pointfeat;
};

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x16 mfma3(bf16x8 ahi, bf16x8 alo, bf16x8 bhi, bf16x8 blo, f32x16 c) {
  c = mfma_bf16(alo, bhi, c);
  c = mfma_bf16(ahi, blo, c);
  c = mfma_bf16(ahi, bhi, c);
  return c;
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

template <int MID, int RT>
__global__ __launch_bounds__(NT) void pointmlp_max_bf16x3_kernel(ArgsB a) {
  constexpr int TP = 32 * RT;
  constexpr int NSUB = TP / 64;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __bf16* h2hi = (__bf16*)smem_raw;
  __bf16* h2lo = h2hi + TP * SH;
  float* rmax = (float*)(h2lo + TP * SH);
  float* xs = rmax + 1024;
  float* hA = (float*)(h2hi + (TP - 64) * SH);   // f32 [64][68] scratch aliasing the last 64 rows of the hi image
  float* hB = (float*)(h2lo + (TP - 64) * SH);   // ... of the lo image

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int b = blockIdx.x / a.nsplit;
  const int split = blockIdx.x - b * a.nsplit;
  const int ntiles = (a.N + TP - 1) / TP;
  const int t_begin = (int)(((long)ntiles * split) / a.nsplit);
  const int t_end = (int)(((long)ntiles * (split + 1)) / a.nsplit);

  for (int i = tid; i < 1024; i += NT) rmax[i] = -INFINITY;

  float w1r[6], b1r;
  {
    const int ch = tid & 63;
#pragma unroll
    for (int j = 0; j < 6; ++j) w1r[j] = a.w1[ch * 6 + j];
    b1r = a.b1[ch];
  }
  float t3r[9];
  if (a.t3) {
#pragma unroll
    for (int j = 0; j < 9; ++j) t3r[j] = a.t3[b * 9 + j];
  }
  const float* xb = a.x + (size_t)b * a.N * 6;

  for (int tile = t_begin; tile < t_end; ++tile) {
    // ================= front layers on 64-point sub-tiles =================
    for (int sub = 0; sub < NSUB; ++sub) {
      __syncthreads();   // previous users of hA/hB/xs (and, for sub 0, the previous tile's L3 reads) are done
      if (tid < 64) {
        int p = tile * TP + sub * 64 + tid;
        if (p >= a.N) p = a.N - 1;
        const f32x2* src = (const f32x2*)(xb + (size_t)p * 6);
        f32x2 v0 = src[0], v1 = src[1], v2 = src[2];
        float px = v0[0], py = v0[1], pz = v1[0];
        if (a.t3) {
          const float qx = px * t3r[0] + py * t3r[3] + pz * t3r[6];
          const float qy = px * t3r[1] + py * t3r[4] + pz * t3r[7];
          const float qz = px * t3r[2] + py * t3r[5] + pz * t3r[8];
          px = qx; py = qy; pz = qz;
        }
        *(f32x4*)(xs + tid * XS) = f32x4{px, py, pz, v1[1]};
        *(f32x4*)(xs + tid * XS + 4) = f32x4{v2[0], v2[1], 0.f, 0.f};
      }
      __syncthreads();
      {  // L0: 6 -> 64, f32 VALU.  thread = (channel, 8-point group)
        float* dst = (MID == 0) ? hB : hA;
        const int ch = tid & 63;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int p = w * 8 + i;
          const f32x4 q0 = *(const f32x4*)(xs + p * XS);
          const f32x2 q1 = *(const f32x2*)(xs + p * XS + 4);
          float v = b1r;
          v = fmaf(w1r[0], q0[0], v); v = fmaf(w1r[1], q0[1], v); v = fmaf(w1r[2], q0[2], v);
          v = fmaf(w1r[3], q0[3], v); v = fmaf(w1r[4], q1[0], v); v = fmaf(w1r[5], q1[1], v);
          dst[p * S64 + ch] = fmaxf(v, 0.f);
        }
      }
      __syncthreads();
      if (MID != 0) {  // mid: 64 -> 64 on waves 0..3 (one 32x32 output tile each)
        if (w < 4) {
          const int rt = w >> 1, nb = w & 1;
          f32x16 c = {0};
          const float* arow = hA + (rt * 32 + l31) * S64 + lhi * 8;
#pragma unroll
          for (int kc = 0; kc < 4; ++kc) {
            bf16x8 ahi, alo, bhi, blo;
            split8(*(const f32x4*)(arow + kc * 16), *(const f32x4*)(arow + kc * 16 + 4), ahi, alo);
            if (MID == 1) {
              load_b(a.wm, nb, kc, 4, lane, bhi, blo);
            } else {
              const float* tp = a.t64 + (size_t)b * 4096 + (kc * 16 + lhi * 8) * 64 + nb * 32 + l31;
              f32x4 u0 = {tp[0], tp[64], tp[128], tp[192]};
              f32x4 u1 = {tp[256], tp[320], tp[384], tp[448]};
              split8(u0, u1, bhi, blo);
            }
            c = mfma3(ahi, alo, bhi, blo, c);
          }
          const int col = nb * 32 + l31;
          const float bias = (MID == 1) ? a.bm[col] : 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rt * 32 + acc_row(r, lane);
            float v = c[r] + bias;
            if (MID == 1) v = fmaxf(v, 0.f);
            hB[row * S64 + col] = v;
            if (MID == 2 && a.pointfeat) {
              const int p = tile * TP + sub * 64 + row;
              if (p < a.N) a.pointfeat[((size_t)b * a.N + p) * 64 + col] = v;
            }
          }
        }
        __syncthreads();
      }
      {  // L2: 64 -> 128, one 32x32 output tile per wave, result split into the bf16 hi/lo images
        const int rt = w >> 2, nb = w & 3;
        f32x16 c = {0};
        const float* arow = hB + (rt * 32 + l31) * S64 + lhi * 8;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          bf16x8 ahi, alo, bhi, blo;
          split8(*(const f32x4*)(arow + kc * 16), *(const f32x4*)(arow + kc * 16 + 4), ahi, alo);
          load_b(a.w2, nb, kc, 4, lane, bhi, blo);
          c = mfma3(ahi, alo, bhi, blo, c);
        }
        if (sub == NSUB - 1) __syncthreads();   // the last sub-tile's rows alias the scratch everyone just read
        const int col = nb * 32 + l31;
        const float bias = a.b2[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = sub * 64 + rt * 32 + acc_row(r, lane);
          const float v = fmaxf(c[r] + bias, 0.f);
          const __bf16 h = (__bf16)v;
          h2hi[row * SH + col] = h;
          h2lo[row * SH + col] = (__bf16)(v - (float)h);
        }
      }
    }
    __syncthreads();
    // ================= L3: 128 -> 1024 + running max.  wave w owns channel blocks [4w, 4w+4) =================
    {
      const __bf16* ahi_base = h2hi + l31 * SH + lhi * 8;
      const __bf16* alo_base = h2lo + l31 * SH + lhi * 8;
      for (int q = 0; q < 4; ++q) {
        const int nb = w * 4 + q;
        f32x16 c[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) c[rt] = f32x16{0};
#pragma unroll
        for (int kc = 0; kc < 8; ++kc) {
          bf16x8 bhi, blo;
          load_b(a.w3, nb, kc, 8, lane, bhi, blo);
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) {
            const bf16x8 ahi = *(const bf16x8*)(ahi_base + rt * 32 * SH + kc * 16);
            const bf16x8 alo = *(const bf16x8*)(alo_base + rt * 32 * SH + kc * 16);
            c[rt] = mfma3(ahi, alo, bhi, blo, c[rt]);
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
  constexpr int TP = 32 * RT;
  const size_t lds = (size_t)2 * TP * SH * 2 + 1024 * 4 + 64 * XS * 4;
  auto kern = pointmlp_max_bf16x3_kernel<MID, RT>;
  static bool attr_set = false;     // per instantiation
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.B * a.nsplit)), dim3(NT), lds, s, a);
  return cg_hip_status(hipGetLastError());
}

template <int RT>
int dispatch_mid(int mid_mode, const ArgsB& a, hipStream_t s) {
  if (mid_mode == 0) return launch<0, RT>(a, s);
  if (mid_mode == 1) return launch<1, RT>(a, s);
  return launch<2, RT>(a, s);
}

}  // namespace

extern "C" int cg_pointmlp_max_bf16x3(const float* x, int B, int N, const float* t3, const float* w1, const float* b1,
                                      int mid_mode, const unsigned short* wm_split, const float* bm, const float* t64,
                                      const unsigned short* w2_split, const float* b2, const unsigned short* w3_split,
                                      const float* b3, int relu3, int nsplit, int tile_points, float* out, float* pointfeat,
                                      void* stream) {
  if (B < 0 || N <= 0 || mid_mode < 0 || mid_mode > 2) return CG_ERR_ARG;
  if (tile_points != 128 && tile_points != 256) return CG_ERR_ARG;
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
  if (tile_points == 128) return dispatch_mid<4>(mid_mode, a, s);
  return dispatch_mid<8>(mid_mode, a, s);
}
