// Split-precision variant of gemm.hip:  Y = act(X . W^T + bias [+ row_bias[row / rows_per_group]]) [+ I_k]
// with every product block evaluated as three 16-bit MFMAs with f32 accumulation (see pointmlp_split.hip):
//     x.w ~= x_lo.w_hi + x_hi.w_lo + x_hi.w_hi      on v_mfma_f32_32x32x16_{f16,bf16}.
// 'f16x3' (IEEE-half pieces, the engine's default): every wide dense layer -- the FC tails (pointnet2.py:178-185, :216-223,
// :295-298) and the per-point segmentation head (:324-328); the 9- and 10-wide output layers stay on the exact-f32 kernel.
// 'bf16x3': the segmentation head only (with bf16 pieces the FC tails would cost 4e-5 of the 1e-4 error budget for 3 % of the step).
// X (f32) is split while it is staged: 128 rows x 64 columns per workgroup as two 16-bit images (144-byte rows: conflict-free
// ds_read_b128 fragment reads).  W is split and packed on the host (folding.pack_b_split):
// Wp[nb][kc][2 (hi,lo)][lane][8], element e of lane l = W[nb*32 + (l&31)][kc*16 + (l>>5)*8 + e].
#include "cg_split.hpp"
#include "../../include/catgrasp_amd.h"

namespace {

constexpr int BM = 128;      // rows per workgroup (4 row tiles per wave: each weight fragment feeds 12 MFMAs)
constexpr int BK = 64;       // K chunk staged in LDS
constexpr int SR = BK + 8;   // 16-bit elements per LDS row

struct GemmArgsB {
  const float* x; int M; int K; int ldx;
  const unsigned short* wp; int N; int nblocks;
  const float* bias;
  const float* row_bias; int rows_per_group; int ld_rb;
  int relu; int eye_k;
  float* y; int ldy;
  int* status;     // optional device int (F16 only): |= CG_HALF_OVERFLOW / CG_HALF_UNDERFLOW (cg_split.hpp)
};

template <bool F16>
__global__ __launch_bounds__(256) void gemm_bias_act_split_kernel(GemmArgsB a) {
  __shared__ __attribute__((aligned(16))) unsigned short xh[BM * SR];
  __shared__ __attribute__((aligned(16))) unsigned short xl[BM * SR];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int row0 = blockIdx.x * BM;
  const int nb = blockIdx.y * 4 + w;            // this wave's 32-channel block
  const bool active = nb < a.nblocks;
  const int nkc_total = a.K / 16;               // K is a multiple of 16 (checked by the launcher)
  f32x16 c[4];
  float amax = 0.f;
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) c[rt] = f32x16{0};
  for (int k0 = 0; k0 < a.K; k0 += BK) {
    const int kc = min(BK, a.K - k0);
    __syncthreads();
    for (int i = tid; i < BM * (BK / 4); i += 256) {
      const int r = i / (BK / 4), cq = i - r * (BK / 4);
      if (cq * 4 < kc) {
        int row = row0 + r; if (row >= a.M) row = a.M - 1;
        const f32x4 v = *(const f32x4*)(a.x + (size_t)row * a.ldx + k0 + cq * 4);
        unsigned h0, l0, h1, l1;
        split2<F16>(v[0], v[1], h0, l0, amax); split2<F16>(v[2], v[3], h1, l1, amax);
        *(u32x2*)(xh + r * SR + cq * 4) = u32x2{h0, h1};
        *(u32x2*)(xl + r * SR + cq * 4) = u32x2{l0, l1};
      }
    }
    __syncthreads();
    if (active) {
      const frag* bp = (const frag*)a.wp + ((size_t)(nb * nkc_total + k0 / 16) * 2) * 64 + lane;
      const unsigned short* ah0 = xh + l31 * SR + lhi * 8;
      const unsigned short* al0 = xl + l31 * SR + lhi * 8;
      const int ns = kc / 16;
#pragma unroll 2
      for (int s = 0; s < ns; ++s) {
        const frag bh = bp[s * 128], bl = bp[s * 128 + 64];
        frag ah[4], al[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
          ah[rt] = *(const frag*)(ah0 + rt * 32 * SR + s * 16);
          al[rt] = *(const frag*)(al0 + rt * 32 * SR + s * 16);
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) c[rt] = mfma_x<F16>(al[rt], bh, c[rt]);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) c[rt] = mfma_x<F16>(ah[rt], bl, c[rt]);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) c[rt] = mfma_x<F16>(ah[rt], bh, c[rt]);
      }
    }
  }
  if constexpr (F16) {      // amax covers this thread's share of the workgroup's X tile over the whole K
    // Overflow: any value anywhere.  Underflow: decided over the WHOLE 128-row tile the workgroup staged (all four waves), not over
    // one wave's strided share of it -- a share that happens to hold only dead post-ReLU entries must not flag a healthy tile.
    __shared__ int seen_large;
    if (tid == 0) seen_large = 0;
    __syncthreads();
    const bool over = __builtin_amdgcn_ballot_w64(!(amax < HALF_MAX)) != 0;
    const bool large = __builtin_amdgcn_ballot_w64(amax >= HALF_LOW) != 0;
    if (lane == 0 && large) atomicOr(&seen_large, 1);
    __syncthreads();
    if (lane == 0 && a.status && blockIdx.y == 0) {      // the column blocks (blockIdx.y) stage the same X tile: report once
      const int flags = (over ? CG_HALF_OVERFLOW : 0) | ((w == 0 && !seen_large) ? CG_HALF_UNDERFLOW : 0);
      if (flags) atomicOr(a.status, flags);
    }
  }
  if (!active) return;
  const int col = nb * 32 + l31;
  if (col >= a.N) return;
  float bias = a.bias ? a.bias[col] : 0.f;
  if (a.eye_k > 0 && (col % (a.eye_k + 1)) == 0) bias += 1.f;   // flattened identity: col = i*k + i
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + rt * 32 + acc_row(r, lane);
      if (row < a.M) {
        float v = c[rt][r] + bias;
        if (a.row_bias) v += a.row_bias[(size_t)(row / a.rows_per_group) * a.ld_rb + col];
        if (a.relu) v = fmaxf(v, 0.f);
        a.y[(size_t)row * a.ldy + col] = v;
      }
    }
  }
}

}  // namespace

template <bool F16>
static int gemm_bias_act_split(const float* x, int M, int K, int ldx, const unsigned short* w_split, int N,
                                const float* bias, const float* row_bias, int rows_per_group, int ld_rb,
                                int relu, int eye_k, float* y, int ldy, int* status, void* stream) {
  if (!x || !w_split || !y) return CG_ERR_ARG;
  if (M < 0 || N <= 0 || K <= 0 || (K % 16) != 0 || (ldx % 4) != 0 || ldx < K || ldy < N) return CG_ERR_ARG;
  if (((uintptr_t)x & 15) != 0) return CG_ERR_ARG;
  if (row_bias && (rows_per_group <= 0 || ld_rb < N)) return CG_ERR_ARG;
  if (M == 0) return CG_OK;
  GemmArgsB a{x, M, K, ldx, w_split, N, (N + 31) / 32, bias, row_bias, rows_per_group, ld_rb, relu, eye_k, y, ldy, F16 ? status : nullptr};
  dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((a.nblocks + 3) / 4)), block(256);
  hipLaunchKernelGGL(gemm_bias_act_split_kernel<F16>, grid, block, 0, (hipStream_t)stream, a);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_gemm_bias_act_bf16x3(const float* x, int M, int K, int ldx, const unsigned short* w_split, int N,
                                       const float* bias, const float* row_bias, int rows_per_group, int ld_rb,
                                       int relu, int eye_k, float* y, int ldy, void* stream) {
  return gemm_bias_act_split<false>(x, M, K, ldx, w_split, N, bias, row_bias, rows_per_group, ld_rb, relu, eye_k, y, ldy, nullptr, stream);
}

extern "C" int cg_gemm_bias_act_f16x3(const float* x, int M, int K, int ldx, const unsigned short* w_split, int N,
                                       const float* bias, const float* row_bias, int rows_per_group, int ld_rb,
                                       int relu, int eye_k, float* y, int ldy, int* status, void* stream) {
  return gemm_bias_act_split<true>(x, M, K, ldx, w_split, N, bias, row_bias, rows_per_group, ld_rb, relu, eye_k, y, ldy, status, stream);
}
