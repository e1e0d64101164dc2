// Row N4: the CUDA ops of PointGroup/lib/pointgroup_ops that sit on the reference's inference path
// (predicter.py:285-304) as gfx950 HIP kernels -- segmented, HBM-bound gather/reduce work:
//   ballquery_batch_p   src/bfs_cluster/bfs_cluster.cu:15-62   per-point radius query inside the point's own batch
//   sec_mean/min/max    src/sec_mean/sec_mean.cu:12-85         segmented reduce over CSR offsets
//   roipool_fp          src/roipool/roipool.cu:12-40           segmented arg-max pool
//   get_iou             src/get_iou/get_iou.cu:12-37           proposal x instance IoU
//   voxelize_fp         src/voxelize/voxelize.cu:10-34         mean/sum pool of point features through a rule book
//   point_recover       src/voxelize/voxelize.cpp:182-192      voxel features back onto their member points (scatter-add)
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


// ---------------------------------------------------------------------------------------------------------------------
// voxelization_idx (src/voxelize/voxelize.cpp:11-151; predicter.py:285) and bfs_cluster (src/bfs_cluster/bfs_cluster.cpp:34-121;
// pointgroup.py:240,245).  The reference runs both on the HOST (std::map insertion order; a sequential queue BFS) and copies
// the tensors over.  Device formulation:
//  * voxelization_idx = sort the packed (batch,x,y,z) keys (stable: ties keep point order), number the runs by their first
//    point index (= the reference's first-appearance order), and fill the two maps -- kernels below + a device sort/scan.
//  * bfs_cluster = connected components of the same-label neighbour graph by min-label propagation with pointer jumping;
//    clusters are numbered by their smallest point index, which is the order the reference's seed loop discovers them in.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void voxel_pack_kernel(const long long* __restrict__ coords, int n, int ncol, long long* __restrict__ keys,
                                                         int* __restrict__ err_flag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long* c = coords + (size_t)i * ncol;
  long long key = 0; bool ok = true;
  if (ncol == 4) { ok = c[0] >= 0 && c[0] < 32768; key = c[0]; }
  for (int j = ncol - 3; j < ncol; ++j) { ok = ok && c[j] >= 0 && c[j] < 65536; key = (key << 16) | (c[j] & 0xffff); }
  if (!ok && err_flag) *err_flag = 1;
  keys[i] = key;
}

__global__ __launch_bounds__(256) void segment_heads_kernel(const long long* __restrict__ sorted_keys, int n, int* __restrict__ head) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  head[j] = (j == 0 || sorted_keys[j] != sorted_keys[j - 1]) ? 1 : 0;
}

// sorted position j -> point perm[j], run seg[j] (starting at seg_start), voxel id vid[seg[j]];  mode 3/4: every member listed,
// mode 1: the first member (outputRows.front(), voxelize.cpp:124-129), mode 2: the last (back(), :130-135), mode 0: the only one.
__global__ __launch_bounds__(256) void voxel_fill_maps_kernel(const long long* __restrict__ perm, const int* __restrict__ seg,
                                                              const int* __restrict__ seg_start, const int* __restrict__ vid, int n,
                                                              int width, int mode, int* __restrict__ input_map, int* __restrict__ output_map) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int sg = seg[j], v = vid[sg], p = (int)perm[j];
  const int pos = j - seg_start[sg];
  const bool last = (j == n - 1) || (seg[j + 1] != sg);
  input_map[p] = v;
  int* row = output_map + (size_t)v * width;
  if (mode == 3 || mode == 4) {
    row[1 + pos] = p;
    if (last) row[0] = pos + 1;
  } else {
    if (pos == 0) row[0] = 1;
    if ((mode == 2) ? last : (pos == 0)) row[1] = p;
  }
}

// Rows are clamped to the n_idx entries that exist and entries outside [0, n) are skipped: a truncated / corrupt CSR list can
// make the result incomplete but never an out-of-bounds access (the host wrapper rejects such input before it gets here).
__global__ __launch_bounds__(256) void cc_propagate_kernel(const int* __restrict__ label, const int* __restrict__ nbr, int n_idx,
                                                           const int* __restrict__ start_len, int n, int* __restrict__ comp,
                                                           int* __restrict__ changed) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int li = label[i];
  const long s0 = start_len[2 * i], e0 = s0 + (long)start_len[2 * i + 1];
  const int s = (int)(s0 < 0 ? 0 : (s0 > n_idx ? n_idx : s0)), e = (int)(e0 < s ? s : (e0 > n_idx ? n_idx : e0));
  int c = comp[i];
  for (int q = s; q < e; ++q) { const int j = nbr[q]; if ((unsigned)j < (unsigned)n && label[j] == li) c = min(c, comp[j]); }
  c = min(c, comp[c]);                                   // pointer jumping
  bool ch = false;
  if (c < comp[i]) { atomicMin(comp + i, c); ch = true; }
  for (int q = s; q < e; ++q) {                          // push to the neighbours too: the relation is used symmetrically
    const int j = nbr[q];
    if ((unsigned)j < (unsigned)n && label[j] == li && comp[j] > c) { atomicMin(comp + j, c); ch = true; }
  }
  if (ch) *changed = 1;
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

// point_recover forward (src/voxelize/voxelize.cpp:182-192 = voxelize_bp_cuda_ with average = false, voxelize.cu:34-48): every voxel row
// adds its feature row to each of its member points.  One lane per (member, channel) pair of a row; atomicAdd like the reference (a
// point listed by several rows receives their sum; in a map made by voxelization_idx every point has exactly one row, so the
// result is a copy and independent of the order of the additions).  A member index outside [0, n_points) sets *err_flag instead of
// writing (the reference writes wherever it points).
__global__ __launch_bounds__(256) void point_recover_kernel(const float* __restrict__ feats, const int* __restrict__ rules, int nRows, int maxActive,
                                                            int C, int nPoints, float* __restrict__ out, int* __restrict__ err) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nRows) return;
  const int* r = rules + (size_t)row * (maxActive + 1);
  int nActive = r[0];
  if (nActive < 0 || nActive > maxActive) { if (lane == 0) *err = 1; nActive = nActive < 0 ? 0 : maxActive; }
  const float* f = feats + (size_t)row * C;
  for (long t = lane; t < (long)nActive * C; t += 64) {
    const int i = (int)(t / C), c = (int)(t - (long)i * C);
    const int p = r[1 + i];
    if (p < 0 || p >= nPoints) { *err = 1; continue; }
    atomicAdd(out + (size_t)p * C + c, f[c]);
  }
}

extern "C" int cg_pg_point_recover(const float* feats, const int* rules, int n_rows, int max_active, int C, int n_points, float* out,
                                   int* err_flag, void* stream) {
  if (n_rows < 0 || max_active < 0 || C <= 0 || n_points < 0) return CG_ERR_ARG;
  if (n_rows == 0) return CG_OK;
  if (!feats || !rules || !out || !err_flag) return CG_ERR_ARG;
  hipLaunchKernelGGL(point_recover_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, feats, rules, n_rows,
                     max_active, C, n_points, out, err_flag);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_pg_voxel_pack_keys(const long long* coords, int n, int ncol, long long* keys, int* err_flag, void* stream) {
  if (n < 0 || (ncol != 3 && ncol != 4)) return CG_ERR_ARG;
  if (n == 0) return CG_OK;
  if (!coords || !keys) return CG_ERR_ARG;
  hipLaunchKernelGGL(voxel_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, coords, n, ncol, keys, err_flag);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_pg_segment_heads(const long long* sorted_keys, int n, int* head, void* stream) {
  if (n < 0) return CG_ERR_ARG;
  if (n == 0) return CG_OK;
  if (!sorted_keys || !head) return CG_ERR_ARG;
  hipLaunchKernelGGL(segment_heads_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sorted_keys, n, head);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_pg_voxel_fill_maps(const long long* perm, const int* seg, const int* seg_start, const int* vid, int n, int width, int mode,
                                     int* input_map, int* output_map, void* stream) {
  if (n < 0 || width < 2 || mode < 0 || mode > 4) return CG_ERR_ARG;
  if (n == 0) return CG_OK;
  if (!perm || !seg || !seg_start || !vid || !input_map || !output_map) return CG_ERR_ARG;
  hipLaunchKernelGGL(voxel_fill_maps_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, perm, seg, seg_start, vid,
                     n, width, mode, input_map, output_map);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_pg_cc_propagate(const int* semantic_label, const int* ball_query_idxs, int n_idx, const int* start_len, int n, int* comp,
                                  int* changed, void* stream) {
  if (n < 0 || n_idx < 0) return CG_ERR_ARG;
  if (n == 0) return CG_OK;
  if (!semantic_label || !start_len || !comp || !changed || (n_idx > 0 && !ball_query_idxs)) return CG_ERR_ARG;
  hipLaunchKernelGGL(cc_propagate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, semantic_label,
                     ball_query_idxs, n_idx, start_len, n, comp, changed);
  return cg_hip_status(hipGetLastError());
}
