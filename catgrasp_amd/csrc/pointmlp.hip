// Fused shared per-point MLP chain + max-pool over points ("PointNet set encoder" pass).
//
// Replaces, per network pass, the reference op sequence
//   Conv1d(k=1) -> BatchNorm1d -> ReLU  (x2 or x3)  ->  Conv1d(128->1024) -> BN [-> ReLU] -> max(dim=N)
// of STN3d.forward (pointnet2.py:172-176), STNkd.forward (:210-214) and
// PointNetEncoder.forward (:243-266), including the learned input transform `bmm(x, trans)` (:248)
// and feature transform `bmm(x, trans_feat)` (:257).  BatchNorm (eval) is folded into the conv
// weights on the host (catgrasp_amd/folding.py), so every layer here is  y = act(W' x + b').
//
// One workgroup (4 waves) owns one sample (or one slice of a sample's point tiles).  A tile of
// TP=64 points is carried through the whole chain inside LDS; only the 1024-wide max ever
// leaves the CU.  All contractions with K>=64 run on v_mfma_f32_32x32x2_f32 (exact f32):
// points are the MFMA M dimension (A operand, ds_read_b128 from LDS), channels the N dimension
// (B operand, pre-packed weight fragments streamed from L2 with global_load_dwordx4), so the
// max over points is a per-lane reduction over the 16 accumulator registers + one lane^32 swap.
#include <stdlib.h>
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"
#include "l3_f32_asm.inc"

namespace {

constexpr int TP = 64;      // points per tile
constexpr int S64 = 68;     // LDS row stride (floats) of 64-wide activations (16B aligned, conflict-free b128)
constexpr int S128 = 132;   // LDS row stride of the 128-wide activation
constexpr int XS = 8;       // LDS row stride of the staged input points

struct Args {
  const float* x; int B; int N;
  const float* t3;                  // (B,9) or null
  const float* w1; const float* b1; // (64,6) row-major, (64)
  const float* wm; const float* bm; // packed 64->64 (MID==1)
  const float* t64;                 // (B,64,64) (MID==2), TRANSPOSED: t64[b][n][k] = T_b[k][n];  h' = h . T
  const float* w2; const float* b2; // packed 64->128
  const float* w3; const float* b3; // packed 128->1024
  int relu3;
  int nsplit;                       // workgroups per sample (point tiles are divided between them)
  int n_main; int tail_split;       // samples [n_main, B) use tail_split workgroups each (tail balancing, see the launcher)
  float* out;                       // (B,1024); pre-filled with -inf when nsplit>1
  float* pointfeat;                 // optional (B,N,64): output of the mid layer (MID==2 only)
};

// LDS layout (floats): [h2 / hA region: TP*S128] [hB: TP*S64] [xs: TP*XS] [rmax: 1024]
constexpr int WM_FLOATS = 2 * 8 * 64 * 4;          // the 64 -> 64 layer's B fragments [nb 2][ks 8][lane 64][4]: 16 KB
constexpr int LDS_FLOATS = TP * S128 + TP * S64 + TP * XS + 1024 + WM_FLOATS;

// CS: workgroups that share one (sample, tile slice) and divide the 1024 output channels of the 128 -> 1024 layer between them (each
// repeats the cheap front layers).  CS = 1 is the throughput kernel; CS = 2 / 4 / 8 serve calls of a few poses (predicter.py:67-94 scores
// a few hundred per object, a live caller one at a time): one pose is 32 tiles, i.e. 32 workgroups that each stream the whole 0.5 MB
// weight image through one MFMA chain per wave (38 us per pass) -- split 8 ways it is 256 workgroups with one 32-channel block per wave.
// Per output element the sequence of MFMAs is the one the CS = 1 stream issues (k ascending from a zero accumulator), so a
// candidate's bits do not depend on how many candidates it was scored with.
template <int MID, int CS>
__global__ __launch_bounds__(256) void pointmlp_max_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* h2 = smem;                 // also hosts hA (first TP*S64 floats) while h2 is not live
  float* hA = smem;
  float* hB = smem + TP * S128;
  float* xs = hB + TP * S64;
  float* rmax = xs + TP * XS;
  f32x4* wms = (f32x4*)(rmax + 1024);      // MID != 0: tile-invariant mid-layer weight fragments, staged once per workgroup

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = tid >> 6;
  int b, split, nsp;
  const int bid = CS == 1 ? (int)blockIdx.x : (int)blockIdx.x / CS;
  const int cs = CS == 1 ? 0 : (int)blockIdx.x % CS;           // this workgroup's share of the last layer's channels
  if (bid < a.n_main * a.nsplit) {
    nsp = a.nsplit; b = bid / nsp; split = bid - b * nsp;
  } else {
    const int r = bid - a.n_main * a.nsplit;
    nsp = a.tail_split; b = a.n_main + r / nsp; split = r - (r / nsp) * nsp;
  }
  const int ntiles = (a.N + TP - 1) / TP;
  const int t_begin = (int)(((long)ntiles * split) / nsp);
  const int t_end = (int)(((long)ntiles * (split + 1)) / nsp);

  for (int i = tid; i < 1024; i += 256) rmax[i] = -INFINITY;

  // per-thread first-layer weights: channel = tid&63
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
  const int l31 = lane & 31;
  const int lhi = lane >> 5;
  // Tile-invariant operands of the front layers, fetched ONCE per workgroup instead of once per tile with their L2 latency exposed
  // between two barriers: the 64 -> 128 weight fragments of this wave's channel block live in registers, the 64 -> 64 fragments
  // (shared weights, or this sample's feature transform) in LDS.
  f32x4 w2r[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) w2r[ks] = ((const f32x4*)a.w2)[(w * 8 + ks) * 64 + lane];
  if (MID == 1) {
    for (int i = tid; i < 2 * 8 * 64; i += 256) wms[i] = ((const f32x4*)a.wm)[i];
  }
  if (MID == 2) {      // t64 is stored TRANSPOSED (Tt[n][k] = T[k][n]): a lane's 4 consecutive k are one 16-byte load
    for (int i = tid; i < 2 * 8 * 64; i += 256) {
      const int ln = i & 63, ks = (i >> 6) & 7, nb = i >> 9;
      wms[i] = *(const f32x4*)(a.t64 + (size_t)b * 4096 + (nb * 32 + (ln & 31)) * 64 + ks * 8 + (ln >> 5) * 4);
    }
  }
  // the next tile's points travel from HBM during the current tile's 128 -> 1024 stream
  f32x2 xn0 = {0.f, 0.f}, xn1 = xn0, xn2 = xn0;
  auto fetch_points = [&](int tile) {
    if (tid < TP && tile < t_end) {
      int p = tile * TP + tid;
      if (p >= a.N) p = a.N - 1;        // replicate the last point: max-pool is idempotent
      const f32x2* src = (const f32x2*)(xb + (size_t)p * 6);
      xn0 = src[0]; xn1 = src[1]; xn2 = src[2];
    }
  };
  fetch_points(t_begin);

  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();   // previous tile's L3 reads of h2 / our rmax init are complete
    // ---- stage input points (apply the 3x3 input transform to xyz, leave normals) ----
    if (tid < TP) {
      const f32x2 v0 = xn0, v1 = xn1, v2 = xn2;
      float px = v0[0], py = v0[1], pz = v1[0];
      if (a.t3) {
        float qx = px * t3r[0] + py * t3r[3] + pz * t3r[6];
        float qy = px * t3r[1] + py * t3r[4] + pz * t3r[7];
        float qz = px * t3r[2] + py * t3r[5] + pz * t3r[8];
        px = qx; py = qy; pz = qz;
      }
      f32x4 o0 = {px, py, pz, v1[1]};
      f32x4 o1 = {v2[0], v2[1], 0.f, 0.f};
      *(f32x4*)(xs + tid * XS) = o0;
      *(f32x4*)(xs + tid * XS + 4) = o1;
    }
    fetch_points(tile + 1);
    __syncthreads();
    // ---- L0: 6 -> 64 on VALU.  thread = (channel, 16-point group) ----
    {
      float* dst = (MID == 0) ? hB : hA;
      const int ch = tid & 63;
#pragma unroll 4
      for (int i = 0; i < 16; ++i) {
        const int p = w * 16 + i;
        f32x4 q0 = *(const f32x4*)(xs + p * XS);
        f32x2 q1 = *(const f32x2*)(xs + p * XS + 4);
        float v = b1r;
        v = fmaf(w1r[0], q0[0], v); v = fmaf(w1r[1], q0[1], v); v = fmaf(w1r[2], q0[2], v);
        v = fmaf(w1r[3], q0[3], v); v = fmaf(w1r[4], q1[0], v); v = fmaf(w1r[5], q1[1], v);
        dst[p * S64 + ch] = fmaxf(v, 0.f);
      }
    }
    __syncthreads();
    // ---- mid: 64 -> 64 (shared conv+BN+ReLU, or per-sample feature transform) ----
    if (MID != 0) {
      const int rt = w >> 1, nb = w & 1;
      f32x16 c = {0};
      const float* arow = hA + (rt * 32 + l31) * S64 + lhi * 4;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        f32x4 av = *(const f32x4*)(arow + ks * 8);
        const f32x4 bv = wms[(nb * 8 + ks) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) c = mfma32(av[j], bv[j], c);
      }
      const int col = nb * 32 + l31;
      const float bias = (MID == 1) ? a.bm[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rt * 32 + acc_row(r, lane);
        float v = c[r] + bias;
        if (MID == 1) v = fmaxf(v, 0.f);
        hB[row * S64 + col] = v;
        if (MID == 2 && a.pointfeat && cs == 0) {
          const int p = tile * TP + row;
          if (p < a.N) a.pointfeat[((size_t)b * a.N + p) * 64 + col] = v;
        }
      }
      __syncthreads();
    }
    // ---- L2: 64 -> 128.  wave w owns channel block w for both row tiles ----
    {
      f32x16 c0 = {0}, c1 = {0};
      const float* ar0 = hB + l31 * S64 + lhi * 4;
      const float* ar1 = ar0 + 32 * S64;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const f32x4 bv = w2r[ks];
        f32x4 a0 = *(const f32x4*)(ar0 + ks * 8);
        f32x4 a1 = *(const f32x4*)(ar1 + ks * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) { c0 = mfma32(a0[j], bv[j], c0); c1 = mfma32(a1[j], bv[j], c1); }
      }
      // hA (aliasing h2) may still be read by other waves' mid layer only before the barrier above,
      // and hB reads of this layer do not alias h2 -> safe to write h2 now when MID!=0; for MID==0
      // nothing else lives in the h2 region.
      const int col = w * 32 + l31;
      const float bias = a.b2[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = acc_row(r, lane);
        h2[row * S128 + col] = fmaxf(c0[r] + bias, 0.f);
        h2[(row + 32) * S128 + col] = fmaxf(c1[r] + bias, 0.f);
      }
    }
    __syncthreads();
    // ---- L3: 128 -> 1024 + running max over points.  wave w owns channel blocks [8w, 8w+8), two at a time x both row tiles.
    // The 1024-MFMA stream of the tile is hand-scheduled assembly (gen_l3_f32_asm.py -> l3_f32_asm.inc): operands prefetched one
    // whole k-step ahead into a double buffer in accumulation registers; it returns the per-lane maxima of the 8 blocks.
    if constexpr (CS > 1) {
      constexpr int NBW = 8 / CS;                          // 32-channel blocks per wave
      const float* ar0 = h2 + l31 * S128 + lhi * 4;
      const float* ar1 = ar0 + 32 * S128;
#pragma unroll
      for (int p = 0; p < NBW; ++p) {
        const int nb = cs * (32 / CS) + w * NBW + p;
        f32x4 bv[16];                                      // the block's 16 weight fragments: all requested before the first product
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) bv[ks] = ((const f32x4*)a.w3)[(size_t)(nb * 16 + ks) * 64 + lane];
        f32x16 c0 = {0}, c1 = {0};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const f32x4 a0 = *(const f32x4*)(ar0 + ks * 8);
          const f32x4 a1 = *(const f32x4*)(ar1 + ks * 8);
#pragma unroll
          for (int j = 0; j < 4; ++j) { c0 = mfma32(a0[j], bv[ks][j], c0); c1 = mfma32(a1[j], bv[ks][j], c1); }
        }
        float m = fmaxf(max16(c0), max16(c1));
        m = fmaxf(m, __shfl_xor(m, 32));
        if (lane < 32) rmax[nb * 32 + lane] = fmaxf(rmax[nb * 32 + lane], m);
      }
    } else {
      const unsigned ar = (unsigned)(uintptr_t)(h2 + l31 * S128 + lhi * 4);      // generic -> LDS byte address = low 32 bits
      const unsigned voff = (unsigned)((w * 8 * 16) * 64 + lane) * 16u;          // this wave's first weight fragment
      float m00, m01, m10, m11, m20, m21, m30, m31, t0, t1;
      unsigned vo0, vo1;
      asm volatile(CG_L3_F32_ASM
                   : [m00] "=&v"(m00), [m01] "=&v"(m01), [m10] "=&v"(m10), [m11] "=&v"(m11), [m20] "=&v"(m20), [m21] "=&v"(m21),
                     [m30] "=&v"(m30), [m31] "=&v"(m31), [t0] "=&v"(t0), [t1] "=&v"(t1), [vo0] "=&v"(vo0), [vo1] "=&v"(vo1)
                   : [voff] "v"(voff), [ar] "v"(ar), [wb] "s"(a.w3)
                   : "memory", CG_L3_F32_CLOBBERS);
      const float mm[4][2] = {{m00, m01}, {m10, m11}, {m20, m21}, {m30, m31}};
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float m0 = mm[p][0], m1 = mm[p][1];
        m0 = fmaxf(m0, __shfl_xor(m0, 32));
        m1 = fmaxf(m1, __shfl_xor(m1, 32));
        if (lane < 32) {
          const int ch0 = (w * 8 + p * 2) * 32 + lane;
          rmax[ch0] = fmaxf(rmax[ch0], m0);
          rmax[ch0 + 32] = fmaxf(rmax[ch0 + 32], m1);
        }
      }
    }
  }
  __syncthreads();
  if (t_end > t_begin) {
    for (int ch = tid; ch < 1024; ch += 256) {
      if (CS > 1 && ch / (1024 / CS) != cs) continue;
      float v = rmax[ch] + a.b3[ch];
      if (a.relu3) v = fmaxf(v, 0.f);
      if (nsp == 1) a.out[(size_t)b * 1024 + ch] = v;
      else atomic_max_f32(a.out + (size_t)b * 1024 + ch, v);
    }
  }
}

__global__ void fill_kernel(float* p, size_t n, float v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace

extern "C" int cg_pointmlp_max(const float* x, int B, int N, const float* t3, const float* w1, const float* b1,
                               int mid_mode, const float* wm_packed, const float* bm, const float* t64,
                               const float* w2_packed, const float* b2, const float* w3_packed, const float* b3,
                               int relu3, int nsplit, float* out, float* pointfeat, void* stream) {
  if (!x || !w1 || !b1 || !w2_packed || !b2 || !w3_packed || !b3 || !out) return CG_ERR_ARG;
  if (B < 0 || N <= 0 || mid_mode < 0 || mid_mode > 2) return CG_ERR_ARG;
  if (mid_mode == 1 && (!wm_packed || !bm)) return CG_ERR_ARG;
  if (mid_mode == 2 && !t64) return CG_ERR_ARG;
  if (pointfeat && mid_mode != 2) return CG_ERR_ARG;
  if (B == 0) return CG_OK;
  const int ntiles = (N + TP - 1) / TP;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > ntiles) nsplit = ntiles;
  hipStream_t s = (hipStream_t)stream;
  // Tail balancing (as in pointmlp_split.hip): with one workgroup per sample and B >= the number of resident workgroups
  // (2 per CU), the samples of the last, partially filled scheduling round are split 8 ways so that round is short.
  int n_main = B, tail_split = 1;
  if (nsplit == 1 && ntiles >= 8) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return CG_ERR_UNSUPPORTED;
    const int slots = 2 * cg_device_cu_count(dev);          // resident workgroups (57 KB of LDS each): two per CU
    if (slots <= 0) return CG_ERR_UNSUPPORTED;
    if (B >= slots && (B % slots) != 0) { n_main = B - B % slots; tail_split = 8; }
  }
  if (nsplit > 1 || tail_split > 1) {
    const int first = (nsplit > 1) ? 0 : n_main;
    const size_t n = (size_t)(B - first) * 1024;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out + (size_t)first * 1024, n, -INFINITY);
  }
  Args a{x, B, N, t3, w1, b1, wm_packed, bm, t64, w2_packed, b2, w3_packed, b3, relu3, nsplit, n_main, tail_split, out, pointfeat};
  const size_t lds = LDS_FLOATS * sizeof(float);          // 73,728 B: above the 64 KB default limit -> per-device function attribute
  int dev_l = 0;
  if (hipGetDevice(&dev_l) != hipSuccess || dev_l < 0 || dev_l >= CG_MAX_DEVICES) return CG_ERR_UNSUPPORTED;
  // few workgroups (a call of a few poses): divide the last layer's channels over 2 / 4 / 8 workgroups per (sample, slice)
  const long wgs = (long)n_main * nsplit + (long)(B - n_main) * tail_split;
  int csi = tail_split > 1 ? 0 : wgs <= 64 ? 3 : wgs <= 128 ? 2 : wgs <= 256 ? 1 : 0;
  static const char* cs_env = getenv("CATGRASP_AMD_POINTMLP_CSPLIT");     // dev knob: 1 / 2 / 4 / 8
  if (cs_env) { const int v = atoi(cs_env); csi = tail_split > 1 ? 0 : v == 8 ? 3 : v == 4 ? 2 : v == 2 ? 1 : 0; }
  static bool attr_set[3][4][CG_MAX_DEVICES] = {};
  typedef void (*kern_t)(Args);
  static const kern_t kerns[3][4] = {
      {pointmlp_max_kernel<0, 1>, pointmlp_max_kernel<0, 2>, pointmlp_max_kernel<0, 4>, pointmlp_max_kernel<0, 8>},
      {pointmlp_max_kernel<1, 1>, pointmlp_max_kernel<1, 2>, pointmlp_max_kernel<1, 4>, pointmlp_max_kernel<1, 8>},
      {pointmlp_max_kernel<2, 1>, pointmlp_max_kernel<2, 2>, pointmlp_max_kernel<2, 4>, pointmlp_max_kernel<2, 8>}};
  const kern_t kern = kerns[mid_mode][csi];
  if (!attr_set[mid_mode][csi][dev_l]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set[mid_mode][csi][dev_l] = true;
  }
  dim3 grid((unsigned)(wgs << csi)), block(256);
  hipLaunchKernelGGL(kern, grid, block, lds, s, a);
  return cg_hip_status(hipGetLastError());
}
