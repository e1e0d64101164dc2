// Fused PointNet++ set-abstraction layer: group -> shared per-neighbour MLP -> max over the neighbourhood
// (BASELINE.json north_star: "grouped per-neighbourhood MLP reductions ... LDS-staged neighbourhoods and wavefront shuffle
// reductions, MFMA only for the dense per-point MLP GEMMs").
//
// Replaces the op sequence that consumes sample_and_group's output (pointnet2.py:101-129):
//   new_points (B,S,K,3+D) = cat(xyz[idx] - new_xyz, points[idx])            pointnet2.py:116-123
//   -> permute to (B,3+D,K,S) -> [Conv2d(1x1) -> BatchNorm2d -> ReLU] x L -> max over K -> (B,C_L,S)
// without ever materialising the grouped tensor (K x the input in HBM) or any activation.
//
// One wavefront owns one neighbourhood (b, s).  Its K neighbours are the 32 rows of an MFMA tile (K > 32: several row tiles,
// K < 32: rows padded by repeating neighbour 0 -- the max is idempotent): the gathered, centred coordinates + features are
// staged in a wave-private LDS strip, every layer is  act(W' x + b')  with BatchNorm folded on the host, evaluated as
// v_mfma_f32_32x32x2_f32 (exact f32) with the neighbours as the M dimension and the channels as N; a layer computes ALL its output
// channel blocks before storing anything, so it overwrites its own input strip (no ping-pong pair), and the last layer's max over
// neighbours is a per-lane reduction over the accumulator registers + one lane^32 exchange -- no workgroup barrier anywhere in the
// kernel.  The strip is as wide as the widest STORED activation, so e.g. the 64-64-128 layer needs 8.7 KB per wave: 8 waves per
// workgroup, two workgroups per CU.
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
  float* out; int* err_flag;
};

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// One layer for the wave's 32-row tile: ALL NNB output channel blocks at once, so an activation fragment is read from LDS once per
// k-step and feeds 4 x NNB MFMAs on NNB independent accumulators, and the NNB weight fragments of a k-step are requested
// together.  Because every accumulator is complete before anything is stored, the layer writes its output over its own input: one
// strip per wave instead of a ping-pong pair.  Last layer: the ReLU'd accumulators go straight into the running max.
template <int NNB>
__device__ __forceinline__ void sa_layer(float* strip, int CS, const float* __restrict__ w, const float* __restrict__ bias, int nks, bool last,
                                         int lane, float* run) {
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
  wave_sync();                       // every lane's fragment reads of the strip are done before it is overwritten
#pragma unroll
  for (int nb = 0; nb < NNB; ++nb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) c[nb][r] = fmaxf(c[nb][r], 0.f);
    if (!last) {
#pragma unroll
      for (int r = 0; r < 16; ++r) strip[acc_row(r, lane) * CS + nb * 32 + l31] = c[nb][r];
    } else {
      float m = max16(c[nb]);
      m = fmaxf(m, __shfl_xor(m, 32));
      run[nb] = fmaxf(run[nb], m);
    }
  }
  wave_sync();
}

// CS: row stride (floats) of the wave-private activation strip = widest STORED activation (layer inputs; the last layer's output
// goes straight from the accumulators into the max) + 4 -> conflict-free 16-byte fragment reads.  A run-time value, so narrow
// networks get small strips and therefore more resident waves (the chain inside a wave is latency bound: more waves = more overlap).
template <int MAXNB>          // widest layer / 32: 4 (all widths <= 128; fits 128 registers, 4 waves per SIMD) or 8
__global__ __launch_bounds__(512) void sa_group_mlp_max_kernel(SAArgs a, int CS, int WAVES) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  float* strip = smem + (size_t)wv * 32 * CS;
  const long g = (long)blockIdx.x * WAVES + wv;     // neighbourhood index b*S + s
  if (g >= (long)a.B * a.S) return;
  const int b = (int)(g / a.S), s = (int)(g - (long)b * a.S);
  const float cx = a.new_xyz[g * 3 + 0], cy = a.new_xyz[g * 3 + 1], cz = a.new_xyz[g * 3 + 2];
  const long long* idg = a.idx + g * a.K;
  const int c_last = a.cout[a.nlayers - 1];
  float run[MAXNB];
#pragma unroll
  for (int q = 0; q < MAXNB; ++q) run[q] = -INFINITY;

  for (int k0 = 0; k0 < a.K; k0 += 32) {
    // ---- gather + centre: lane (row r, half h) writes channels [8h, 8h+8) of neighbour k0 + r ----
    {
      const int kk = k0 + l31;
      long long id = idg[kk < a.K ? kk : 0];
      if (id < 0 || id >= a.N) { if (a.err_flag) *a.err_flag = 1; id = 0; }      // index_points raises on such an index
      const float* px = a.xyz + ((size_t)b * a.N + id) * 3;
      const float* pf = a.points ? a.points + ((size_t)b * a.N + id) * a.D : nullptr;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ch = 8 * lhi + j;
        float x = 0.f;
        if (ch == 0) x = px[0] - cx; else if (ch == 1) x = px[1] - cy; else if (ch == 2) x = px[2] - cz;
        else if (ch - 3 < a.D) x = pf[ch - 3];
        v[j] = x;
      }
      *(f32x4*)(strip + l31 * CS + 8 * lhi) = f32x4{v[0], v[1], v[2], v[3]};
      *(f32x4*)(strip + l31 * CS + 8 * lhi + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
    wave_sync();
    for (int l = 0; l < a.nlayers; ++l) {
      const int nks = a.cin[l] / 8;
      const bool last = (l == a.nlayers - 1);
      const int nnb = a.cout[l] / 32;        // wave-uniform
      if (nnb == 1) sa_layer<1>(strip, CS, a.w[l], a.b[l], nks, last, lane, run);
      else if (nnb == 2) sa_layer<2>(strip, CS, a.w[l], a.b[l], nks, last, lane, run);
      else if (nnb == 3) sa_layer<3>(strip, CS, a.w[l], a.b[l], nks, last, lane, run);
      else if (nnb == 4) sa_layer<4>(strip, CS, a.w[l], a.b[l], nks, last, lane, run);
      else if constexpr (MAXNB == 8) {
        if (nnb == 5) sa_layer<5>(strip, CS, a.w[l], a.b[l], nks, last, lane, run);
        else if (nnb == 6) sa_layer<6>(strip, CS, a.w[l], a.b[l], nks, last, lane, run);
        else if (nnb == 7) sa_layer<7>(strip, CS, a.w[l], a.b[l], nks, last, lane, run);
        else sa_layer<8>(strip, CS, a.w[l], a.b[l], nks, last, lane, run);
      }
    }
  }
  if (lane < 32) {
#pragma unroll
    for (int q = 0; q < MAXNB; ++q)
      if (q * 32 < c_last) a.out[((size_t)b * c_last + q * 32 + lane) * a.S + s] = run[q];
  }
}

template <int MAXNB>
int launch_sa(const SAArgs& a, int cs, hipStream_t s, int dev) {
  const size_t per_wave = (size_t)32 * cs * 4;
  int waves = (int)((size_t)(78 * 1024) / per_wave);        // <= half a CU's LDS per workgroup: two workgroups co-reside
  if (waves > 8) waves = 8;
  if (waves < 1) return CG_ERR_UNSUPPORTED;
  const size_t lds = per_wave * waves;
  auto kern = sa_group_mlp_max_kernel<MAXNB>;
  static bool attr_set[CG_MAX_DEVICES] = {};
  if (dev < 0 || dev >= CG_MAX_DEVICES) return CG_ERR_UNSUPPORTED;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set[dev] = true;
  }
  const long groups = (long)a.B * a.S;
  hipLaunchKernelGGL(kern, dim3((unsigned)((groups + waves - 1) / waves)), dim3(64 * waves), lds, s, a, cs, waves);
  return cg_hip_status(hipGetLastError());
}

}  // namespace

extern "C" int cg_sa_group_mlp_max(const float* xyz, const float* points, const float* new_xyz, const long long* idx, int B, int N, int S,
                                   int K, int D, int n_layers, const int* h_cin, const int* h_cout, const float* const* h_w_packed,
                                   const float* const* h_bias, float* out, int* err_flag, void* stream) {
  if (B < 0 || N <= 0 || S < 0 || K <= 0 || D < 0 || n_layers < 1 || n_layers > SA_MAX_LAYERS) return CG_ERR_ARG;
  if (!h_cin || !h_cout || !h_w_packed || !h_bias) return CG_ERR_ARG;
  if ((long)B * S == 0) return CG_OK;
  if (!xyz || !new_xyz || !idx || !out || (D > 0 && !points)) return CG_ERR_ARG;
  if (3 + D > C0) return CG_ERR_UNSUPPORTED;
  SAArgs a{};
  a.xyz = xyz; a.points = D > 0 ? points : nullptr; a.new_xyz = new_xyz; a.idx = idx;
  a.B = B; a.N = N; a.S = S; a.K = K; a.D = D; a.nlayers = n_layers; a.out = out; a.err_flag = err_flag;
  int cmax = 0, cstore = C0;
  for (int l = 0; l < n_layers; ++l) {
    if (!h_w_packed[l] || !h_bias[l]) return CG_ERR_ARG;
    if (h_cin[l] != (l == 0 ? C0 : h_cout[l - 1])) return CG_ERR_ARG;          // layer 0: K padded to 16 by the host
    if (h_cout[l] <= 0 || (h_cout[l] % 32) != 0) return CG_ERR_UNSUPPORTED;
    a.cin[l] = h_cin[l]; a.cout[l] = h_cout[l]; a.w[l] = h_w_packed[l]; a.b[l] = h_bias[l];
    if (h_cout[l] > cmax) cmax = h_cout[l];
    if (l + 1 < n_layers && h_cout[l] > cstore) cstore = h_cout[l];
  }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return CG_ERR_UNSUPPORTED;
  if (cmax > 256) return CG_ERR_UNSUPPORTED;
  if (cmax <= 128) return launch_sa<4>(a, cstore + 4, (hipStream_t)stream, dev);
  return launch_sa<8>(a, cstore + 4, (hipStream_t)stream, dev);
}
