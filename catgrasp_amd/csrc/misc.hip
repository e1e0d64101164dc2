// Small HBM-bound kernels around the two networks:
//  * cg_build_grasp_input  : per-candidate GraspDataset.transform on the device (dataset_grasp.py:63-91):
//                            gather resampled points, cloud -> grasp frame, optional (x-mean)/std.
//  * cg_build_nunocs_input : NunocsIsolatedDataset.transform + NormalizeCloud
//                            (dataset_nunocs.py:38-65, augmentations.py:66-75).
//  * cg_softmax_pg         : softmax / argmax / confidence of predicter.py:86-91 and the
//                            p_G reduction of run_grasp_simulation.py:313.
//  * cg_nunocs_decode      : bin argmax decode + z-confidence of predicter.py:144-150.
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"

namespace {

// one thread per PAIR of consecutive output points: an 8-byte id load, two gathers from the L2-resident object cloud, and
// three 16-byte stores per thread, so a wave writes one contiguous 3 KB span with full-width stores.
__device__ __forceinline__ void grasp_point(const float* __restrict__ xyz, const float* __restrict__ nrm, int id,
                                            const float* __restrict__ T, const float* __restrict__ mean,
                                            const float* __restrict__ inv_std, float* v) {
  const float px = xyz[(size_t)id * 3 + 0], py = xyz[(size_t)id * 3 + 1], pz = xyz[(size_t)id * 3 + 2];
  const float nx = nrm[(size_t)id * 3 + 0], ny = nrm[(size_t)id * 3 + 1], nz = nrm[(size_t)id * 3 + 2];
  v[0] = fmaf(T[0], px, fmaf(T[1], py, fmaf(T[2], pz, T[3])));      // rows of [R | t]: x_g = R x + t, n_g = R n
  v[1] = fmaf(T[4], px, fmaf(T[5], py, fmaf(T[6], pz, T[7])));
  v[2] = fmaf(T[8], px, fmaf(T[9], py, fmaf(T[10], pz, T[11])));
  v[3] = fmaf(T[0], nx, fmaf(T[1], ny, T[2] * nz));
  v[4] = fmaf(T[4], nx, fmaf(T[5], ny, T[6] * nz));
  v[5] = fmaf(T[8], nx, fmaf(T[9], ny, T[10] * nz));
  if (mean) {
#pragma unroll
    for (int j = 0; j < 6; ++j) v[j] = (v[j] - mean[j]) * inv_std[j];
  }
}

__global__ __launch_bounds__(256) void build_grasp_input_kernel(
    const float* __restrict__ xyz, const float* __restrict__ nrm, const int* __restrict__ ids,
    const float* __restrict__ pose_inv, const float* __restrict__ mean, const float* __restrict__ inv_std,
    int G, int n_pts, float* __restrict__ out) {
  const long total = (long)G * n_pts;
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i >= total) return;
  float v[12];
  if (i + 1 < total && (n_pts & 1) == 0) {          // the pair lies inside one candidate and is 8-byte aligned
    const int2 id = *(const int2*)(ids + i);
    const float* T = pose_inv + (size_t)(i / n_pts) * 12;
    grasp_point(xyz, nrm, id.x, T, mean, inv_std, v);
    grasp_point(xyz, nrm, id.y, T, mean, inv_std, v + 6);
    f32x4* o = (f32x4*)(out + i * 6);
    o[0] = f32x4{v[0], v[1], v[2], v[3]}; o[1] = f32x4{v[4], v[5], v[6], v[7]}; o[2] = f32x4{v[8], v[9], v[10], v[11]};
    return;
  }
  for (long k = i; k < total && k < i + 2; ++k) {   // odd n_pts: scalar path
    grasp_point(xyz, nrm, ids[k], pose_inv + (size_t)(k / n_pts) * 12, mean, inv_std, v);
    for (int j = 0; j < 6; ++j) out[k * 6 + j] = v[j];
  }
}

// Staged variant (the one that normally runs): a workgroup owns BGI_CPB consecutive candidates.  All resample indices of a
// candidate point into ONE object's slice of the scene cloud (dataset_grasp.py:66-69 resamples the object cloud), so per
// candidate the workgroup finds the id range and -- unless the slice already in LDS covers it, the common case for
// consecutive candidates of one object -- copies that slice (<= BGI_CAP points, xyz + normal = 63 KB) into LDS with
// coalesced 16-byte loads, then gathers from LDS: 12-byte random gathers from L2 move a whole cache line each (~10x the
// useful bytes).  Falls back to L2 gathers when the range does not fit.  Output rows are transposed through a wave-private
// LDS strip so the stores are 16 B per lane at 16-byte stride (row-per-lane 24-byte-stride stores measured 1.25x slower).
// Same fmaf chain as above, so results are bit-identical.
constexpr int BGI_CPB = 8;
constexpr int BGI_CAP = 2688;
constexpr int BGI_NT = 512;         // 8 waves; two workgroups per CU (77 KB LDS each)
__global__ __launch_bounds__(BGI_NT) void build_grasp_input_staged_kernel(
    const float* __restrict__ xyz, const float* __restrict__ nrm, int n_cloud, const int* __restrict__ ids,
    const float* __restrict__ pose_inv, const float* __restrict__ mean, const float* __restrict__ inv_std,
    int G, int n_pts, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float sx[BGI_CAP * 3];
  __shared__ __attribute__((aligned(16))) float sn[BGI_CAP * 3];
  __shared__ __attribute__((aligned(16))) float tw[(BGI_NT / 64) * 64 * 6];
  __shared__ int red[2][BGI_NT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float* strip = tw + wv * (64 * 6);
  int cur_lo = 0, cur_hi = -1;                         // slice currently staged: points [cur_lo, cur_hi]
  const int g_end = min(G, (int)(blockIdx.x + 1) * BGI_CPB);
  for (int g = blockIdx.x * BGI_CPB; g < g_end; ++g) {
    const int* idb = ids + (size_t)g * n_pts;         // n_pts % 64 == 0, rows 16-byte aligned (checked by the launcher)
    int lo = 0x7fffffff, hi = -1;
    for (int i = tid * 4; i < n_pts; i += BGI_NT * 4) {
      const int4 v = *(const int4*)(idb + i);
      lo = min(min(lo, v.x), min(v.y, min(v.z, v.w)));
      hi = max(max(hi, v.x), max(v.y, max(v.z, v.w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); }
    __syncthreads();                                   // previous candidate: strips, slice and red[] are no longer in use
    if (lane == 0) { red[0][wv] = lo; red[1][wv] = hi; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BGI_NT / 64; ++k) { lo = min(lo, red[0][k]); hi = max(hi, red[1][k]); }
    bool staged = lo >= cur_lo && hi <= cur_hi;
    if (!staged && lo >= 0 && hi < n_cloud && hi - (lo & ~3) + 1 <= BGI_CAP) {
      cur_lo = lo & ~3; cur_hi = hi;                   // 4-point granule: the slice starts 16-byte aligned
      const int nfl = (cur_hi - cur_lo + 1) * 3;       // floats to copy from each array
      const float* gx = xyz + (size_t)cur_lo * 3;
      const float* gn = nrm + (size_t)cur_lo * 3;
      for (int i = tid * 4; i < nfl; i += BGI_NT * 4) {
        if (i + 4 <= nfl) {
          *(f32x4*)(sx + i) = *(const f32x4*)(gx + i);
          *(f32x4*)(sn + i) = *(const f32x4*)(gn + i);
        } else {
          for (int k = i; k < nfl; ++k) { sx[k] = gx[k]; sn[k] = gn[k]; }
        }
      }
      staged = true;
      __syncthreads();
    }
    float T[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) T[j] = pose_inv[(size_t)g * 12 + j];
    auto emit = [&](int i0, int id) {
      float v[6];
      if (staged) grasp_point(sx, sn, id - cur_lo, T, mean, inv_std, v);
      else grasp_point(xyz, nrm, id, T, mean, inv_std, v);
      f32x2* sp = (f32x2*)(strip + lane * 6);
      sp[0] = f32x2{v[0], v[1]}; sp[1] = f32x2{v[2], v[3]}; sp[2] = f32x2{v[4], v[5]};
      __builtin_amdgcn_wave_barrier();
      f32x4* o = (f32x4*)(out + ((size_t)g * n_pts + i0) * 6);
      o[lane] = *(const f32x4*)(strip + lane * 4);
      if (lane < 32) o[64 + lane] = *(const f32x4*)(strip + 256 + lane * 4);
      __builtin_amdgcn_wave_barrier();
    };
    // two 64-point strips per round so that two id loads (and their dependent chains) are in flight per wave
    for (int i0 = wv * 64; i0 < n_pts; i0 += 2 * BGI_NT) {
      const int i1 = i0 + BGI_NT;
      const int ida = idb[i0 + lane];
      const int idc = (i1 < n_pts) ? idb[i1 + lane] : 0;
      emit(i0, ida);
      if (i1 < n_pts) emit(i1, idc);
    }
  }
}

// one workgroup per object cloud: gather -> min/max reduce -> normalise.
__global__ __launch_bounds__(1024) void build_nunocs_input_kernel(
    const float* __restrict__ xyz, const float* __restrict__ nrm, const int* __restrict__ ids,
    const float* __restrict__ mean, const float* __restrict__ inv_std, int n_pts, float* __restrict__ out) {
  __shared__ float red[6][16];
  __shared__ float mn[3], inv_scale;
  const int b = blockIdx.x;
  const int* idb = ids + (size_t)b * n_pts;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int i = threadIdx.x; i < n_pts; i += blockDim.x) {
    const int id = idb[i];
#pragma unroll
    for (int j = 0; j < 3; ++j) { float v = xyz[(size_t)id * 3 + j]; lo[j] = fminf(lo[j], v); hi[j] = fmaxf(hi[j], v); }
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    for (int o = 32; o > 0; o >>= 1) { lo[j] = fminf(lo[j], __shfl_xor(lo[j], o)); hi[j] = fmaxf(hi[j], __shfl_xor(hi[j], o)); }
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { for (int j = 0; j < 3; ++j) { red[j][w] = lo[j]; red[3 + j][w] = hi[j]; } }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = blockDim.x >> 6;
    float l[3], h[3];
    for (int j = 0; j < 3; ++j) { l[j] = red[j][0]; h[j] = red[3 + j][0]; for (int k = 1; k < nw; ++k) { l[j] = fminf(l[j], red[j][k]); h[j] = fmaxf(h[j], red[3 + j][k]); } }
    const float scale = fmaxf(fmaxf(h[0] - l[0], h[1] - l[1]), h[2] - l[2]);
    mn[0] = l[0]; mn[1] = l[1]; mn[2] = l[2];
    inv_scale = scale + 1e-15f;   // divisor, as the reference: (xyz - min) / (scale + 1e-15)
  }
  __syncthreads();
  const float div = inv_scale;
  for (int i = threadIdx.x; i < n_pts; i += blockDim.x) {
    const int id = idb[i];
    float v[6];
#pragma unroll
    for (int j = 0; j < 3; ++j) { v[j] = (xyz[(size_t)id * 3 + j] - mn[j]) / div; v[3 + j] = nrm[(size_t)id * 3 + j]; }
    if (mean) {
#pragma unroll
      for (int j = 0; j < 6; ++j) v[j] = (v[j] - mean[j]) * inv_std[j];
    }
    float* o = out + ((size_t)b * n_pts + i) * 6;
#pragma unroll
    for (int j = 0; j < 6; ++j) o[j] = v[j];
  }
}

__global__ __launch_bounds__(256) void softmax_pg_kernel(const float* __restrict__ logits, int B, int C,
                                                         float* __restrict__ probs, int* __restrict__ label,
                                                         float* __restrict__ conf, float* __restrict__ p_g) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* l = logits + (size_t)b * C;
  float m = l[0]; int am = 0;
  for (int k = 1; k < C; ++k) { if (l[k] > m) { m = l[k]; am = k; } }
  float s = 0.f;
  for (int k = 0; k < C; ++k) s += expf(l[k] - m);
  const float inv = 1.f / s;
  float pg = 0.f, best = -1.f; int bl = 0;
  for (int k = 0; k < C; ++k) {
    const float p = expf(l[k] - m) * inv;
    probs[(size_t)b * C + k] = p;
    pg = fmaf(p, (float)k, pg);
    if (p > best) { best = p; bl = k; }       // argmax of the probabilities, first maximum (predicter.py:89)
  }
  (void)am;
  label[b] = bl; conf[b] = best; p_g[b] = pg / (float)C;
}

// A wavefront decodes DEC_ROWS consecutive (point, axis) rows.  A row's nbins logits are contiguous, so they are read
// coalesced; all rows' loads are issued before the first reduction (12 loads = 2.4 KB in flight per wave -- a single row per
// wave keeps too few bytes in flight to cover HBM latency).  First-maximum arg-max (torch.argmax semantics,
// predicter.py:144-148) by a (value, index) butterfly; z rows also reduce sum exp.
constexpr int DEC_ROWS = 6;
constexpr int DEC_MAX_PER_LANE = 2;      // nbins <= 128
__global__ __launch_bounds__(256) void nunocs_decode_kernel(const float* __restrict__ logits, long P, int nbins,
                                                            float* __restrict__ coords, float* __restrict__ conf_z) {
  const long row0 = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * DEC_ROWS;
  const long nrows = P * 3;
  if (row0 >= nrows) return;
  const int lane = threadIdx.x & 63;
  float v[DEC_ROWS][DEC_MAX_PER_LANE];
#pragma unroll
  for (int r = 0; r < DEC_ROWS; ++r) {
    const float* l = logits + (row0 + r) * nbins;
#pragma unroll
    for (int j = 0; j < DEC_MAX_PER_LANE; ++j) {
      const int k = lane + 64 * j;
      v[r][j] = (row0 + r < nrows && k < nbins) ? l[k] : -INFINITY;
    }
  }
#pragma unroll
  for (int r = 0; r < DEC_ROWS; ++r) {
    const long i = row0 + r;
    if (i >= nrows) break;
    float m = v[r][0]; int am = lane;
#pragma unroll
    for (int j = 1; j < DEC_MAX_PER_LANE; ++j) { if (v[r][j] > m) { m = v[r][j]; am = lane + 64 * j; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float om = __shfl_xor(m, o); const int oa = __shfl_xor(am, o);
      if (om > m || (om == m && oa < am)) { m = om; am = oa; }
    }
    if (lane == 0) coords[i] = (float)am * (1.0f / (float)nbins) - 0.5f;
    if ((i % 3) == 2) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < DEC_MAX_PER_LANE; ++j) { if (lane + 64 * j < nbins) s += expf(v[r][j] - m); }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
      if (lane == 0) conf_z[i / 3] = 1.f / s;
    }
  }
}

// nbins % 4 == 0 (config_nunocs.yml: 100): HALF a wavefront per row, one 16-byte load per lane (25 of 32 lanes for 100 bins), two
// adjacent rows = 800 contiguous bytes per wave instruction, DEC4_PAIRS row pairs (3.2 KB) in flight per wave; the (value, index)
// arg-max and the sum of exponentials reduce through DPP row operations + one lane^16 exchange instead of six ds_bpermute
// round trips per value.  Same first-maximum semantics; 40 -> see profiles/ for the 8 x 8192-point decode.
constexpr int DEC4_PAIRS = 4;
template <int CTRL>
__device__ __forceinline__ void dpp_argmax_step(float& m, int& am) {
  const float om = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(m), __float_as_int(m), CTRL, 0xf, 0xf, false));
  const int oa = __builtin_amdgcn_update_dpp(am, am, CTRL, 0xf, 0xf, false);
  const bool take = om > m || (om == m && oa < am);
  m = take ? om : m; am = take ? oa : am;
}
template <int CTRL>
__device__ __forceinline__ float dpp_add_step(float s) {
  return s + __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(s), __float_as_int(s), CTRL, 0xf, 0xf, false));
}
__global__ __launch_bounds__(256) void nunocs_decode_x4_kernel(const float* __restrict__ logits, long P, int nbins,
                                                               float* __restrict__ coords, float* __restrict__ conf_z) {
  const int lane = threadIdx.x & 63, half = lane >> 5, t = lane & 31;
  const long nrows = P * 3;
  const long row0 = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (2 * DEC4_PAIRS);
  if (row0 >= nrows) return;
  const int nq = nbins >> 2;                      // 16-byte pieces per row, <= 32
  f32x4 v[DEC4_PAIRS];
#pragma unroll
  for (int q = 0; q < DEC4_PAIRS; ++q) {
    const long row = row0 + 2 * q + half;
    v[q] = (row < nrows && t < nq) ? *(const f32x4*)(logits + row * nbins + 4 * t) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  }
#pragma unroll
  for (int q = 0; q < DEC4_PAIRS; ++q) {
    const long row = row0 + 2 * q + half;
    float m = v[q][0]; int am = 4 * t;
#pragma unroll
    for (int j = 1; j < 4; ++j) { if (v[q][j] > m) { m = v[q][j]; am = 4 * t + j; } }
    dpp_argmax_step<0xB1>(m, am);                 // quad_perm [1,0,3,2]
    dpp_argmax_step<0x4E>(m, am);                 // quad_perm [2,3,0,1]
    dpp_argmax_step<0x141>(m, am);                // row_half_mirror
    dpp_argmax_step<0x140>(m, am);                // row_mirror: every lane of a 16-lane row holds its row's result
    {                                             // the two 16-lane rows of the half
      const float om = __shfl_xor(m, 16); const int oa = __shfl_xor(am, 16);
      const bool take = om > m || (om == m && oa < am);
      m = take ? om : m; am = take ? oa : am;
    }
    if (t == 0 && row < nrows) coords[row] = (float)am * (1.0f / (float)nbins) - 0.5f;
    if ((row0 + 2 * q) % 3 != 0) {                // one of the pair's two rows is a z row (wave-uniform test): softmax confidence of its arg-max
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) s += (t < nq) ? expf(v[q][j] - m) : 0.f;
      s = dpp_add_step<0xB1>(s); s = dpp_add_step<0x4E>(s); s = dpp_add_step<0x141>(s); s = dpp_add_step<0x140>(s);
      s += __shfl_xor(s, 16);
      if (t == 0 && row < nrows && (row % 3) == 2) conf_z[row / 3] = 1.f / s;
    }
  }
}

}  // namespace

namespace {
// out[g][c] = max over the `rows` consecutive rows of group g of x[.][c] (the max-pool over points of pointnet2.py:176,214 for an
// activation tensor that was materialised, i.e. the standalone STNkd).  One workgroup per (group, 64-channel slab): lanes map to
// channels (coalesced 256-byte row reads), the four waves split the rows and combine through LDS.
__global__ __launch_bounds__(256) void group_max_kernel(const float* __restrict__ x, long rows, int C, float* __restrict__ out) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + lane;
  const float* xg = x + (size_t)blockIdx.x * rows * C;
  float m = -INFINITY;
  if (c < C)
    for (long r = w; r < rows; r += 4) m = fmaxf(m, xg[(size_t)r * C + c]);
  part[w][lane] = m;
  __syncthreads();
  if (w == 0 && c < C) out[(size_t)blockIdx.x * C + c] = fmaxf(fmaxf(part[0][lane], part[1][lane]), fmaxf(part[2][lane], part[3][lane]));
}
}  // namespace

extern "C" int cg_group_max(const float* x, long groups, long rows_per_group, int C, float* out, void* stream) {
  if (groups < 0 || rows_per_group <= 0 || C <= 0) return CG_ERR_ARG;
  if (groups == 0) return CG_OK;
  if (!x || !out || groups > 0x7fffffffL) return CG_ERR_ARG;
  hipLaunchKernelGGL(group_max_kernel, dim3((unsigned)groups, (unsigned)((C + 63) / 64)), dim3(256), 0, (hipStream_t)stream, x, rows_per_group, C, out);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_build_grasp_input(const float* cloud_xyz, const float* cloud_normal, int n_cloud, const int* ids,
                                    const float* pose_inv, const float* mean, const float* inv_std, int G, int n_pts,
                                    float* out, void* stream) {
  if (!cloud_xyz || !cloud_normal || !ids || !pose_inv || !out || n_cloud <= 0 || G < 0 || n_pts <= 0) return CG_ERR_ARG;
  if ((mean == nullptr) != (inv_std == nullptr)) return CG_ERR_ARG;
  if (G == 0) return CG_OK;
  if ((n_pts & 63) == 0 && ((uintptr_t)ids & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)cloud_xyz & 15) == 0 &&
      ((uintptr_t)cloud_normal & 15) == 0) {
    hipLaunchKernelGGL(build_grasp_input_staged_kernel, dim3((unsigned)((G + BGI_CPB - 1) / BGI_CPB)), dim3(BGI_NT), 0,
                       (hipStream_t)stream, cloud_xyz, cloud_normal, n_cloud, ids, pose_inv, mean, inv_std, G, n_pts, out);
    return cg_hip_status(hipGetLastError());
  }
  const long total = (long)G * n_pts;
  hipLaunchKernelGGL(build_grasp_input_kernel, dim3((unsigned)((total + 511) / 512)), dim3(256), 0, (hipStream_t)stream,
                     cloud_xyz, cloud_normal, ids, pose_inv, mean, inv_std, G, n_pts, out);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_build_nunocs_input(const float* cloud_xyz, const float* cloud_normal, int n_cloud, const int* ids,
                                     const float* mean, const float* inv_std, int B, int n_pts, float* out, void* stream) {
  if (!cloud_xyz || !cloud_normal || !ids || !out || n_cloud <= 0 || B < 0 || n_pts <= 0) return CG_ERR_ARG;
  if ((mean == nullptr) != (inv_std == nullptr)) return CG_ERR_ARG;
  if (B == 0) return CG_OK;
  hipLaunchKernelGGL(build_nunocs_input_kernel, dim3((unsigned)B), dim3(1024), 0, (hipStream_t)stream,
                     cloud_xyz, cloud_normal, ids, mean, inv_std, n_pts, out);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_softmax_pg(const float* logits, int B, int C, float* probs, int* label, float* conf, float* p_g,
                             void* stream) {
  if (!logits || !probs || !label || !conf || !p_g || B < 0 || C <= 0) return CG_ERR_ARG;
  if (B == 0) return CG_OK;
  hipLaunchKernelGGL(softmax_pg_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     logits, B, C, probs, label, conf, p_g);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_nunocs_decode(const float* logits, long P, int nbins, float* coords, float* conf_z, void* stream) {
  if (!logits || !coords || !conf_z || P < 0 || nbins <= 0) return CG_ERR_ARG;
  if (nbins > 64 * DEC_MAX_PER_LANE) return CG_ERR_UNSUPPORTED;      // config_nunocs.yml: ce_loss_bins = 100
  if (P == 0) return CG_OK;
  if ((nbins & 3) == 0 && (((uintptr_t)logits) & 15) == 0) {
    const long waves4 = (P * 3 + 2 * DEC4_PAIRS - 1) / (2 * DEC4_PAIRS);
    hipLaunchKernelGGL(nunocs_decode_x4_kernel, dim3((unsigned)((waves4 + 3) / 4)), dim3(256), 0, (hipStream_t)stream, logits, P, nbins, coords,
                       conf_z);
    return cg_hip_status(hipGetLastError());
  }
  const long waves = (P * 3 + DEC_ROWS - 1) / DEC_ROWS;
  hipLaunchKernelGGL(nunocs_decode_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     logits, P, nbins, coords, conf_z);
  return cg_hip_status(hipGetLastError());
}
