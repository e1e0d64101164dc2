// TEST INFRASTRUCTURE ONLY.  Launchers around the REFERENCE's own PointGroup CUDA kernels, compiled for gfx950 with hipcc so that
// the reference implementation itself runs on the MI355X as the oracle of catgrasp_amd/csrc/pointgroup_ops.hip (row N4).  The kernels
// are plain CUDA C (threadIdx / blockIdx / atomicAdd): oracle/build_ref.py copies their text -- the __global__ functions only, by
// line range -- from /root/reference/PointGroup/lib/pointgroup_ops/src/*/*.cu AT BUILD TIME into the git-ignored oracle/_ref/ and
// compiles this file around it; nothing of the reference's text lives in the repository, and nothing under catgrasp_amd/ links it.
// Launch geometry as in the reference's own launchers (bfs_cluster.cu:64-92: DIVUP(n,512) x 512; the others: min(rows,32768) x min(C,32),
// get_iou: x min(nInstance,256)).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include <algorithm>

using Int = int32_t;      // src/datatype/datatype.h:9

#include "_ref/pgk_ballquery.inc"
#include "_ref/pgk_sec_mean.inc"
#include "_ref/pgk_sec_min.inc"
#include "_ref/pgk_sec_max.inc"
#include "_ref/pgk_roipool_fp.inc"
#include "_ref/pgk_get_iou.inc"
#include "_ref/pgk_voxelize_fp.inc"
#include "_ref/pgk_voxelize_bp.inc"

static int status() { return (int)hipGetLastError(); }

// cumsum: device int, pre-zeroed; idx: n*meanActive ints; start_len (n,2) ints
extern "C" int ref_ballquery_batch_p(int n, int meanActive, float radius, const float* xyz, const int* batch_idxs, const int* batch_offsets,
                                     int* idx, int* start_len, int* cumsum, void* stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(ballquery_batch_p_cuda_, dim3((unsigned)(n / 512 + (n % 512 > 0))), dim3(512), 0, (hipStream_t)stream, n, meanActive, radius, xyz,
                     batch_idxs, batch_offsets, idx, start_len, cumsum);
  return status();
}

extern "C" int ref_sec(int which, int nProposal, int C, float* inp, int* offsets, float* out, void* stream) {
  if (nProposal <= 0 || C <= 0) return 0;
  dim3 g((unsigned)std::min(nProposal, 32768)), b((unsigned)std::min(C, 32));
  if (which == 0) hipLaunchKernelGGL(sec_mean_cuda_, g, b, 0, (hipStream_t)stream, nProposal, C, inp, offsets, out);
  else if (which == 1) hipLaunchKernelGGL(sec_min_cuda_, g, b, 0, (hipStream_t)stream, nProposal, C, inp, offsets, out);
  else hipLaunchKernelGGL(sec_max_cuda_, g, b, 0, (hipStream_t)stream, nProposal, C, inp, offsets, out);
  return status();
}

extern "C" int ref_roipool_fp(int nProposal, int C, float* feats, int* proposals_offset, float* output_feats, int* output_maxidx, void* stream) {
  if (nProposal <= 0 || C <= 0) return 0;
  hipLaunchKernelGGL(roipool_fp_cuda_, dim3((unsigned)std::min(nProposal, 32768)), dim3((unsigned)std::min(C, 32)), 0, (hipStream_t)stream, nProposal, C, feats,
                     proposals_offset, output_feats, output_maxidx);
  return status();
}

extern "C" int ref_get_iou(int nInstance, int nProposal, int* proposals_idx, int* proposals_offset, long* instance_labels, int* instance_pointnum,
                           float* proposals_iou, void* stream) {
  if (nProposal <= 0 || nInstance <= 0) return 0;
  hipLaunchKernelGGL(get_iou_cuda_, dim3((unsigned)std::min(nProposal, 32768)), dim3((unsigned)std::min(nInstance, 256)), 0, (hipStream_t)stream, nInstance,
                     nProposal, proposals_idx, proposals_offset, instance_labels, instance_pointnum, proposals_iou);
  return status();
}

// output_feats pre-zeroed (the reference's python wrapper allocates it with zero_(), pointgroup_ops.py:57)
extern "C" int ref_voxelize_fp(int nOutputRows, int maxActive, int nPlanes, float* feats, float* output_feats, int* rules, int average, void* stream) {
  if (nOutputRows <= 0 || nPlanes <= 0) return 0;
  hipLaunchKernelGGL(voxelize_fp_cuda_<float>, dim3((unsigned)std::min(nOutputRows, 32768)), dim3((unsigned)std::min(nPlanes, 32)), 0, (hipStream_t)stream,
                     (Int)nOutputRows, (Int)maxActive, (Int)nPlanes, feats, output_feats, rules, (bool)average);
  return status();
}

// point_recover_fp (voxelize.cpp:182-192): the reference runs its voxelize BACKWARD kernel with (d_output_feats, d_feats) = (voxel features,
// pre-zeroed point features) and average = false
extern "C" int ref_point_recover_fp(int nActive, int maxActive, int nPlanes, float* feats, float* output_feats, int* rules, void* stream) {
  if (nActive <= 0 || nPlanes <= 0) return 0;
  hipLaunchKernelGGL(voxelize_bp_cuda_<float>, dim3((unsigned)std::min(nActive, 32768)), dim3((unsigned)std::min(nPlanes, 32)), 0, (hipStream_t)stream,
                     (Int)nActive, (Int)maxActive, (Int)nPlanes, feats, output_feats, rules, false);
  return status();
}
