// Row-batched dense layer  Y = act(X . W^T + bias [+ row_bias[row / rows_per_group]]) [+ I_k]
// on v_mfma_f32_32x32x2_f32 (exact f32).
//
// Replaces the reference's Linear -> BatchNorm1d -> ReLU tails of STN3d / STNkd / PointNetCls
// (pointnet2.py:178-185, :216-223, :295-298; one row per candidate) and the Conv1d(k=1) -> BN -> ReLU
// segmentation head of PointNetSeg (pointnet2.py:324-328; one row per point).  BatchNorm is folded
// into W / bias on the host.  W is pre-packed into MFMA B-fragment order (see pack_b in
// catgrasp_amd/folding.py): Wp[nb][ks][lane][4] = W[nb*32 + (lane&31)][ks*8 + (lane>>5)*4 + j].
#include <stdlib.h>
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"

namespace {

constexpr int BM = 64;       // rows per workgroup
constexpr int BK = 64;       // K chunk staged in LDS
constexpr int SA = BK + 4;   // LDS row stride (floats)

struct GemmArgs {
  const float* x; int M; int K; int ldx;
  const float* wp; int N; int nblocks;   // nblocks = ceil(N/32) (packed, zero padded)
  const float* bias;                      // (N) or null
  const float* row_bias; int rows_per_group; int ld_rb;  // optional per-group bias (groups = row / rows_per_group)
  int relu; int eye_k;                    // eye_k>0: add identity of a flattened k x k matrix
  float* y; int ldy;
  int gmax_rows;                          // > 0: instead of storing Y, fold relu(Y) into y[row / gmax_rows][col] with an atomic max (y pre-zeroed)
};

// Epilogue of the group-all layer's last GEMM (sample_and_group_all + shared MLP + max over all points, pointnet2.py:132-149): the 32 x 32
// accumulator tile `c` (rows row_base ..., column col) is not stored; relu(c + bias) >= 0 is folded into y[group][col] with an integer
// atomic max on the float bits (y pre-zeroed: the order of non-negative floats is the order of their bit patterns).  A tile inside one
// group reduces in registers first (one atomic per column); a tile across a group boundary falls back to one atomic per element.
// fmaxf drops a NaN operand where torch.relu / torch.max keep it: a NaN activation is folded in as the quiet-NaN pattern 0x7fc00000,
// which as an integer lies above every finite float, so it wins the max and the pooled feature is NaN as in the reference's ops.
__device__ __forceinline__ int relu_bits_keep_nan(float v) { return v != v ? 0x7fc00000 : __float_as_int(fmaxf(v, 0.f)); }

__device__ __forceinline__ void gemm_fold_groupmax(const GemmArgs& a, const f32x16& c, int row_base, int col, float bias, int lane) {
  const int last = min(row_base + 31, a.M - 1);
  if (row_base >= a.M) return;
  const int g0 = row_base / a.gmax_rows;
  if (last / a.gmax_rows == g0 && row_base + 31 < a.M) {
    float m = max16(c);
    bool nan = false;
#pragma unroll
    for (int r = 0; r < 16; ++r) nan |= c[r] != c[r];
    m = nan ? __int_as_float(0x7fc00000) : m;
    const float o = __shfl_xor(m, 32);
    m = (o != o) ? o : (m != m ? m : fmaxf(m, o));
    if (lane < 32) atomicMax((int*)(a.y + (size_t)g0 * a.ldy + col), relu_bits_keep_nan(m + bias));
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row_base + acc_row(r, lane);
    if (row < a.M) atomicMax((int*)(a.y + (size_t)(row / a.gmax_rows) * a.ldy + col), relu_bits_keep_nan(c[r] + bias));
  }
}

__global__ __launch_bounds__(256) void gemm_bias_act_kernel(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float xs[BM * SA];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int row0 = blockIdx.x * BM;
  const int nb = blockIdx.y * 4 + w;            // this wave's 32-channel block
  const bool active = nb < a.nblocks;
  const int ksteps_total = a.K / 8;             // K is a multiple of 8 (checked on the host)
  f32x16 c0 = {0}, c1 = {0};
  for (int k0 = 0; k0 < a.K; k0 += BK) {
    const int kc = min(BK, a.K - k0);
    __syncthreads();
    // stage X[row0:row0+64, k0:k0+kc] (rows clamped; columns are multiples of 4)
    for (int i = tid; i < BM * (BK / 4); i += 256) {
      const int r = i / (BK / 4), cq = i - r * (BK / 4);
      if (cq * 4 < kc) {
        int row = row0 + r; if (row >= a.M) row = a.M - 1;
        f32x4 v = *(const f32x4*)(a.x + (size_t)row * a.ldx + k0 + cq * 4);
        *(f32x4*)(xs + r * SA + cq * 4) = v;
      }
    }
    __syncthreads();
    if (active) {
      const f32x4* bp = (const f32x4*)a.wp + ((size_t)nb * ksteps_total + k0 / 8) * 64 + lane;
      const float* ar0 = xs + l31 * SA + lhi * 4;
      const float* ar1 = ar0 + 32 * SA;
      const int nks = kc / 8;
#pragma unroll 4
      for (int ks = 0; ks < nks; ++ks) {
        f32x4 bv = bp[ks * 64];
        f32x4 a0 = *(const f32x4*)(ar0 + ks * 8);
        f32x4 a1 = *(const f32x4*)(ar1 + ks * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) { c0 = mfma32(a0[j], bv[j], c0); c1 = mfma32(a1[j], bv[j], c1); }
      }
    }
  }
  if (!active) return;
  const int col = nb * 32 + l31;
  if (col >= a.N) return;
  float bias = a.bias ? a.bias[col] : 0.f;
  if (a.eye_k > 0 && (col % (a.eye_k + 1)) == 0) bias += 1.f;   // flattened identity: col = i*k + i
  if (a.gmax_rows > 0) { gemm_fold_groupmax(a, c0, row0, col, bias, lane); gemm_fold_groupmax(a, c1, row0 + 32, col, bias, lane); return; }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row0 + half * 32 + acc_row(r, lane);
      if (row < a.M) {
        float v = (half ? c1[r] : c0[r]) + bias;
        if (a.row_bias) v += a.row_bias[(size_t)(row / a.rows_per_group) * a.ld_rb + col];
        if (a.relu) v = fmaxf(v, 0.f);
        a.y[(size_t)row * a.ldy + col] = v;
      }
    }
  }
}

// The same layer for FEW output tiles (the FC tails of a predict_batch call of 1 .. ~1,000 poses: the reference scores a few hundred per
// object, predicter.py:67-94).  The tile kernel above gives a 64 x 128 output tile to a workgroup: at M <= 64 and N = 512 that is 4
// workgroups on 4 of 256 CUs, each streaming 128 KB of weights through one MFMA chain per wave -- 42-53 us per launch, nine launches
// per call, i.e. 0.4 ms of a 1 ms call whatever the number of poses.  Here ONE WAVEFRONT owns a 32 x 32 output tile and walks the
// whole K itself, operands straight from global memory (the x rows are L1 / L2 resident, the weights are read once per row tile):
// ceil(M/32) x ceil(N/32) independent wavefronts.  Per output element the sequence of MFMAs and their operands is the one the tile
// kernel issues, so a row's result does not depend on which kernel -- i.e. on how many rows -- it was computed with.
__global__ __launch_bounds__(256) void gemm_bias_act_small_kernel(GemmArgs a) {
  const int lane = threadIdx.x & 63;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int tiles_m = (a.M + 31) / 32;
  if (tile >= tiles_m * a.nblocks) return;
  const int tm = tile / a.nblocks, nb = tile - tm * a.nblocks;     // neighbouring wavefronts share their x rows, not their weights
  const int ksteps = a.K / 8;
  int xrow = tm * 32 + l31; if (xrow >= a.M) xrow = a.M - 1;
  const float* xr = a.x + (size_t)xrow * a.ldx + lhi * 4;
  const f32x4* bp = (const f32x4*)a.wp + (size_t)nb * ksteps * 64 + lane;
  f32x16 c = {0};
  // The chain of MFMAs is fed from global memory: U k-steps of operands (2 x U 16-byte loads per lane) are requested before the previous
  // U are multiplied, so a wavefront has 2 x U loads in flight instead of waiting out one L2 round trip per k-step (the rolled loop the
  // compiler made of the plain form: 232 ns per k-step, 30 us for K = 1024).
  constexpr int U = 8;
  f32x4 bv[2][U], av[2][U];
  const int nfull = ksteps / U;
  if (nfull > 0) {
#pragma unroll
    for (int u = 0; u < U; ++u) { bv[0][u] = bp[(size_t)u * 64]; av[0][u] = *(const f32x4*)(xr + u * 8); }
  }
  for (int blk = 0; blk < nfull; blk += 2) {                // two halves per trip so that the buffers are indexed statically
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int cur = blk + half;
      if (cur < nfull) {
        if (cur + 1 < nfull) {
#pragma unroll
          for (int u = 0; u < U; ++u) {
            bv[half ^ 1][u] = bp[(size_t)((cur + 1) * U + u) * 64];
            av[half ^ 1][u] = *(const f32x4*)(xr + ((cur + 1) * U + u) * 8);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) c = mfma32(av[half][u][j], bv[half][u][j], c);
      }
    }
  }
  for (int ks = nfull * U; ks < ksteps; ++ks) {
    const f32x4 b1 = bp[(size_t)ks * 64];
    const f32x4 a1 = *(const f32x4*)(xr + ks * 8);
#pragma unroll
    for (int j = 0; j < 4; ++j) c = mfma32(a1[j], b1[j], c);
  }
  const int col = nb * 32 + l31;
  if (col >= a.N) return;
  float bias = a.bias ? a.bias[col] : 0.f;
  if (a.eye_k > 0 && (col % (a.eye_k + 1)) == 0) bias += 1.f;
  if (a.gmax_rows > 0) { gemm_fold_groupmax(a, c, tm * 32, col, bias, lane); return; }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = tm * 32 + acc_row(r, lane);
    if (row < a.M) {
      float v = c[r] + bias;
      if (a.row_bias) v += a.row_bias[(size_t)(row / a.rows_per_group) * a.ld_rb + col];
      if (a.relu) v = fmaxf(v, 0.f);
      a.y[(size_t)row * a.ldy + col] = v;
    }
  }
}

constexpr long SMALL_TILES = 2048;  // 32 x 32 output tiles up to which the wavefront-per-tile kernel is used (measured: profiles/r4_gemm_small.txt)

int launch_gemm(GemmArgs& a, void* stream) {
  static const long small_tiles = getenv("CATGRASP_AMD_GEMM_SMALL_TILES") ? atol(getenv("CATGRASP_AMD_GEMM_SMALL_TILES")) : SMALL_TILES;   // dev knob
  const long tiles = (long)((a.M + 31) / 32) * a.nblocks;
  if (tiles <= small_tiles) {
    hipLaunchKernelGGL(gemm_bias_act_small_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
    return cg_hip_status(hipGetLastError());
  }
  dim3 grid((unsigned)((a.M + BM - 1) / BM), (unsigned)((a.nblocks + 3) / 4)), block(256);
  hipLaunchKernelGGL(gemm_bias_act_kernel, grid, block, 0, (hipStream_t)stream, a);
  return cg_hip_status(hipGetLastError());
}

}  // namespace


extern "C" int cg_gemm_bias_act(const float* x, int M, int K, int ldx, const float* w_packed, int N,
                                const float* bias, const float* row_bias, int rows_per_group, int ld_rb,
                                int relu, int eye_k, float* y, int ldy, void* stream) {
  if (!x || !w_packed || !y) return CG_ERR_ARG;
  if (M < 0 || N <= 0 || K <= 0 || (K % 8) != 0 || (ldx % 4) != 0 || ldx < K || ldy < N) return CG_ERR_ARG;
  if (((uintptr_t)x & 15) != 0) return CG_ERR_ARG;
  if (row_bias && (rows_per_group <= 0 || ld_rb < N)) return CG_ERR_ARG;
  if (M == 0) return CG_OK;
  GemmArgs a{x, M, K, ldx, w_packed, N, (N + 31) / 32, bias, row_bias, rows_per_group, ld_rb, relu, eye_k, y, ldy, 0};
  return launch_gemm(a, stream);
}

// out[g][n] = max over the rows_per_group rows of group g of relu(X . W^T + bias): the last layer of the group-all set-abstraction level
// with its max over the points folded into the epilogue (the (M, N) activation is never written).  out (M / rows_per_group, N).
extern "C" int cg_gemm_bias_relu_groupmax(const float* x, int M, int K, int ldx, const float* w_packed, int N, const float* bias,
                                          int rows_per_group, float* out, void* stream) {
  if (!x || !w_packed || !out) return CG_ERR_ARG;
  if (M < 0 || N <= 0 || K <= 0 || (K % 8) != 0 || (ldx % 4) != 0 || ldx < K || rows_per_group <= 0 || (M % rows_per_group) != 0) return CG_ERR_ARG;
  if (((uintptr_t)x & 15) != 0) return CG_ERR_ARG;
  if (M == 0) return CG_OK;
  hipError_t e = hipMemsetAsync(out, 0, (size_t)(M / rows_per_group) * N * sizeof(float), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  GemmArgs a{x, M, K, ldx, w_packed, N, (N + 31) / 32, bias, nullptr, 1, 0, 1, 0, out, N, rows_per_group};
  return launch_gemm(a, stream);
}
