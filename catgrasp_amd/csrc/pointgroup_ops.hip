// Row N4: the CUDA ops of PointGroup/lib/pointgroup_ops that sit on the reference's inference path
// (predicter.py:285-304) as gfx950 HIP kernels -- segmented, HBM-bound gather/reduce work:
//   ballquery_batch_p   src/bfs_cluster/bfs_cluster.cu:15-62   per-point radius query inside the point's own batch
//   sec_mean/min/max    src/sec_mean/sec_mean.cu:12-85         segmented reduce over CSR offsets
//   roipool_fp          src/roipool/roipool.cu:12-40           segmented arg-max pool
//   get_iou             src/get_iou/get_iou.cu:12-37           proposal x instance IoU
//   voxelize_fp         src/voxelize/voxelize.cu:10-34         mean/sum pool of point features through a rule book
// One wavefront handles one (segment, 64-channel slab): lanes map to consecutive channels, so every row read is a
// coalesced 256-byte access; per-channel accumulation order is the reference's (i = start..end), so sums are bitwise
// those of a sequential float32 loop.
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"

namespace {

constexpr int BQ_MAX = 1000;     // idx_temp_size of the reference (bfs_cluster.cu:21)

// pass 0: counts[p] = min(#neighbours with d2 < r2 in the point's batch, 1000)
// pass 1: idx[start[p] .. ) = the first `len[p]` neighbour indices in ascending order
__global__ __launch_bounds__(256) void ballquery_batch_p_kernel(const float* __restrict__ xyz, const int* __restrict__ batch_idxs,
                                                                const int* __restrict__ batch_offsets, int n, float radius2, int pass,
                                                                const int* __restrict__ start, const int* __restrict__ len,
                                                                int* __restrict__ counts, int* __restrict__ idx) {
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n) return;
  const float ox = xyz[p * 3], oy = xyz[p * 3 + 1], oz = xyz[p * 3 + 2];
  const int b = batch_idxs[p];
  const int s = batch_offsets[b], e = batch_offsets[b + 1];
  const int limit = pass == 0 ? BQ_MAX : len[p];
  int* out = pass == 1 ? idx + start[p] : nullptr;
  int cnt = 0;
  for (int k0 = s; k0 < e && cnt < limit; k0 += 64) {
    const int k = k0 + lane;
    bool in = false;
    if (k < e) {
      const float dx = ox - xyz[k * 3], dy = oy - xyz[k * 3 + 1], dz = oz - xyz[k * 3 + 2];
      in = (dx * dx + dy * dy) + dz * dz < radius2;
    }
    const unsigned long long m = __ballot(in);
    const int pos = cnt + __builtin_popcountll(m & ((1ull << lane) - 1ull));
    if (pass == 1 && in && pos < limit) out[pos] = k;
    cnt += __builtin_popcountll(m);
  }
  if (pass == 0 && lane == 0) counts[p] = cnt < BQ_MAX ? cnt : BQ_MAX;
}

// mode 0 mean (sum of inp/count), 1 min, 2 max, 3 max + argmax (roipool)
__global__ __launch_bounds__(256) void segment_reduce_kernel(const float* __restrict__ inp, const int* __restrict__ offsets, int nseg, int C,
                                                             int mode, float* __restrict__ out, int* __restrict__ argmax) {
  const int lane = threadIdx.x & 63;
  const int slabs = (C + 63) / 64;
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (long)nseg * slabs) return;
  const int seg = (int)(w / slabs), c = (int)(w % slabs) * 64 + lane;
  if (c >= C) return;
  const int s = offsets[seg], e = offsets[seg + 1];
  if (mode == 0) {
    const float count = (float)(e - s);
    float mean = 0.f;
    for (int i = s; i < e; ++i) mean += inp[(size_t)i * C + c] / count;
    out[(size_t)seg * C + c] = mean;
  } else if (mode == 1) {
    float v = (float)1e50;                                   // the reference initialises a float with 1e50 -> +inf
    for (int i = s; i < e; ++i) { const float x = inp[(size_t)i * C + c]; if (x < v) v = x; }
    out[(size_t)seg * C + c] = v;
  } else {
    float v = (float)-1e50; int am = -1;
    for (int i = s; i < e; ++i) { const float x = inp[(size_t)i * C + c]; if (x > v) { v = x; am = i; } }
    out[(size_t)seg * C + c] = v;
    if (mode == 3) argmax[(size_t)seg * C + c] = am;
  }
}

__global__ __launch_bounds__(256) void get_iou_kernel(const int* __restrict__ proposals_idx, const int* __restrict__ proposals_offset,
                                                      const long long* __restrict__ instance_labels, const int* __restrict__ instance_pointnum,
                                                      int nProposal, int nInstance, float* __restrict__ iou) {
  const int pr = blockIdx.x;
  if (pr >= nProposal) return;
  const int s = proposals_offset[pr], e = proposals_offset[pr + 1];
  for (int inst = threadIdx.x; inst < nInstance; inst += blockDim.x) {
    int inter = 0;
    for (int i = s; i < e; ++i) inter += ((int)instance_labels[proposals_idx[i]] == inst) ? 1 : 0;
    const int tot = (e - s) + instance_pointnum[inst] - inter;
    iou[(size_t)pr * nInstance + inst] = (float)((double)(float)inter / ((double)(float)tot + 1e-5));
  }
}

__global__ __launch_bounds__(256) void voxelize_fp_kernel(const float* __restrict__ feats, const int* __restrict__ rules, int nRows, int maxActive,
                                                          int C, int average, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int slabs = (C + 63) / 64;
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= (long)nRows * slabs) return;
  const int row = (int)(w / slabs), c = (int)(w % slabs) * 64 + lane;
  if (c >= C) return;
  const int* r = rules + (size_t)row * (maxActive + 1);
  const int nActive = r[0];
  const float mult = (average && nActive > 0) ? 1.0f / (float)nActive : 1.0f;
  float acc = out[(size_t)row * C + c];                      // the reference accumulates into the (zero-initialised) output
  for (int i = 1; i <= nActive; ++i) acc += mult * feats[(size_t)r[i] * C + c];
  out[(size_t)row * C + c] = acc;
}

}  // namespace

extern "C" int cg_pg_ballquery_batch_p(const float* xyz, const int* batch_idxs, const int* batch_offsets, int n, float radius, int pass,
                                       const int* start, const int* len, int* counts, int* idx, void* stream) {
  if (n < 0 || (pass != 0 && pass != 1)) return CG_ERR_ARG;
  if (n == 0) return CG_OK;
  if (!xyz || !batch_idxs || !batch_offsets || (pass == 0 && !counts) || (pass == 1 && (!start || !len || !idx))) return CG_ERR_ARG;
  hipLaunchKernelGGL(ballquery_batch_p_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, xyz, batch_idxs, batch_offsets,
                     n, radius * radius, pass, start, len, counts, idx);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_pg_segment_reduce(const float* inp, const int* offsets, int n_segments, int C, int mode, float* out, int* argmax,
                                    void* stream) {
  if (n_segments < 0 || C <= 0 || mode < 0 || mode > 3) return CG_ERR_ARG;
  if (n_segments == 0) return CG_OK;
  if (!inp || !offsets || !out || (mode == 3 && !argmax)) return CG_ERR_ARG;
  const long waves = (long)n_segments * ((C + 63) / 64);
  hipLaunchKernelGGL(segment_reduce_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, inp, offsets, n_segments, C,
                     mode, out, argmax);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_pg_get_iou(const int* proposals_idx, const int* proposals_offset, const long long* instance_labels,
                             const int* instance_pointnum, int nProposal, int nInstance, float* proposals_iou, void* stream) {
  if (nProposal < 0 || nInstance < 0) return CG_ERR_ARG;
  if ((long)nProposal * nInstance == 0) return CG_OK;
  if (!proposals_idx || !proposals_offset || !instance_labels || !instance_pointnum || !proposals_iou) return CG_ERR_ARG;
  hipLaunchKernelGGL(get_iou_kernel, dim3((unsigned)nProposal), dim3(256), 0, (hipStream_t)stream, proposals_idx, proposals_offset,
                     instance_labels, instance_pointnum, nProposal, nInstance, proposals_iou);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_pg_voxelize_fp(const float* feats, const int* rules, int n_rows, int max_active, int C, int average, float* out,
                                 void* stream) {
  if (n_rows < 0 || max_active < 0 || C <= 0) return CG_ERR_ARG;
  if (n_rows == 0) return CG_OK;
  if (!feats || !rules || !out) return CG_ERR_ARG;
  const long waves = (long)n_rows * ((C + 63) / 64);
  hipLaunchKernelGGL(voxelize_fp_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, feats, rules, n_rows, max_active,
                     C, average, out);
  return cg_hip_status(hipGetLastError());
}
