// NUNOCS -> camera 9-D similarity RANSAC: the device form of aligning.estimate9DTransform
// (aligning.py:33-119, called from predicter.py:164 with max_iter = 10000, twice per object).
// The reference evaluates the hypotheses one after the other on the CPU (cv2.estimateAffine3D on 4 correspondences
// + full-cloud numpy passes); here one workgroup evaluates one hypothesis, all in float64 like the reference:
//   1. exact affine through the 4 sampled correspondences (what estimateAffine3D returns for 4 points),
//   2. column scales, scale bounds, singular values of the scale-free block in [0.8,1.2], nearest rotation
//      R = U V^T, det > 0                                                   (aligning.py:36-52)
//   3. canonical-frame extent check of the whole target cloud               (aligning.py:57-61)
//   4. inlier count |T src - dst| <= threshold over the whole cloud         (aligning.py:63-68)
// The arg-max over hypotheses (first maximum, aligning.py:112) is taken by the caller.
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"

namespace {

struct RansacArgs {
  const double* src; const double* dst; int N;      // (N,3) each
  const int* ids; int H;                            // (H,4) sampled correspondences
  double thres;
  double min_scale[3], max_scale[3];
  int use_dims; double max_dims[3];
  int* counts;                                      // (H) inlier count or -1 (hypothesis rejected)
  double* transforms;                               // (H,16) row-major 4x4
};

// symmetric 3x3 eigen-decomposition by cyclic Jacobi rotations: S = V diag(e) V^T
__device__ void jacobi3(double S[3][3], double V[3][3], double e[3]) {
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V[i][j] = (i == j);
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(S[0][1]) + fabs(S[0][2]) + fabs(S[1][2]);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (fabs(S[p][q]) < 1e-300) continue;
        const double theta = (S[q][q] - S[p][p]) / (2.0 * S[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) { const double a = S[k][p], b = S[k][q]; S[k][p] = c * a - s * b; S[k][q] = s * a + c * b; }
        for (int k = 0; k < 3; ++k) { const double a = S[p][k], b = S[q][k]; S[p][k] = c * a - s * b; S[q][k] = s * a + c * b; }
        for (int k = 0; k < 3; ++k) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - s * b; V[k][q] = s * a + c * b; }
      }
  }
  for (int i = 0; i < 3; ++i) e[i] = S[i][i];
}

// hypothesis model from 4 correspondences; returns false if rejected.  T: row-major [R diag(s) | t] (3x4), Ti its inverse.
// M: the 4 x 7 system [sx sy sz 1 | dx dy dz] in LDS -- the pivot search indexes its rows dynamically, which as a private array put the
// kernel into scratch memory (240 B private segment)
__device__ bool hypothesis(const RansacArgs& a, const int* id, double T[12], double Ti[12], double (*M)[7]) {
  for (int r = 0; r < 4; ++r) {
    const double* s = a.src + (size_t)id[r] * 3; const double* d = a.dst + (size_t)id[r] * 3;
    M[r][0] = s[0]; M[r][1] = s[1]; M[r][2] = s[2]; M[r][3] = 1.0; M[r][4] = d[0]; M[r][5] = d[1]; M[r][6] = d[2];
  }
  double mag = 0.0;
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 3; ++c) mag = fmax(mag, fabs(M[r][c]));
  // Gauss-Jordan with partial pivoting
  for (int c = 0; c < 4; ++c) {
    int piv = c; double best = fabs(M[c][c]);
    for (int r = c + 1; r < 4; ++r) if (fabs(M[r][c]) > best) { best = fabs(M[r][c]); piv = r; }
    if (!(best > 1e-13 * fmax(mag, 1e-300))) return false;          // coplanar / repeated samples: no affine model
    if (piv != c) for (int k = 0; k < 7; ++k) { const double t = M[c][k]; M[c][k] = M[piv][k]; M[piv][k] = t; }
    const double inv = 1.0 / M[c][c];
    for (int k = 0; k < 7; ++k) M[c][k] *= inv;
    for (int r = 0; r < 4; ++r) if (r != c) { const double f = M[r][c]; if (f != 0.0) for (int k = 0; k < 7; ++k) M[r][k] -= f * M[c][k]; }
  }
  // X = M[:, 4:7] (4x3): dst = [src 1] X  ->  A[i][j] = X[j][i], t[i] = X[3][i]
  double A[3][3], t[3], sc[3];
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) A[i][j] = M[j][4 + i]; t[i] = M[3][4 + i]; }
  for (int j = 0; j < 3; ++j) {
    sc[j] = sqrt(A[0][j] * A[0][j] + A[1][j] * A[1][j] + A[2][j] * A[2][j]);      // np.linalg.norm(transform[:3,:3], axis=0)
    if (sc[j] > a.max_scale[j] || sc[j] < a.min_scale[j] || !(sc[j] > 0.0)) return false;
  }
  double Rn[3][3], S[3][3], V[3][3], e[3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rn[i][j] = A[i][j] / sc[j];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) S[i][j] = Rn[0][i] * Rn[0][j] + Rn[1][i] * Rn[1][j] + Rn[2][i] * Rn[2][j];
  jacobi3(S, V, e);
  double smin = 1e300, smax = 0.0, isg[3];
  for (int i = 0; i < 3; ++i) { const double sg = sqrt(fmax(e[i], 0.0)); smin = fmin(smin, sg); smax = fmax(smax, sg); isg[i] = sg > 0 ? 1.0 / sg : 0.0; }
  if (smin < 0.8 || smax > 1.2) return false;                                     // aligning.py:44-45
  // nearest rotation U V^T = Rn (Rn^T Rn)^(-1/2) = Rn V diag(1/sigma) V^T
  double P[3][3], R[3][3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) P[i][j] = V[i][0] * isg[0] * V[j][0] + V[i][1] * isg[1] * V[j][1] + V[i][2] * isg[2] * V[j][2];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i][j] = Rn[i][0] * P[0][j] + Rn[i][1] * P[1][j] + Rn[i][2] * P[2][j];
  const double det = R[0][0] * (R[1][1] * R[2][2] - R[1][2] * R[2][1]) - R[0][1] * (R[1][0] * R[2][2] - R[1][2] * R[2][0]) +
                     R[0][2] * (R[1][0] * R[2][1] - R[1][1] * R[2][0]);
  if (det < 0.0) return false;                                                    // aligning.py:48-49
  for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T[i * 4 + j] = R[i][j] * sc[j]; T[i * 4 + 3] = t[i]; }
  // inverse: diag(1/s) R^T (x - t)
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) Ti[i * 4 + j] = R[j][i] / sc[i];
    Ti[i * 4 + 3] = -(Ti[i * 4 + 0] * t[0] + Ti[i * 4 + 1] * t[1] + Ti[i * 4 + 2] * t[2]);
  }
  return true;
}

__device__ __forceinline__ double wave_min(double v) { for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o)); return v; }
__device__ __forceinline__ double wave_max(double v) { for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o)); return v; }

__global__ __launch_bounds__(256) void ransac_9d_kernel(RansacArgs a) {
  __shared__ double sT[12], sTi[12], sM[4][7];
  __shared__ int s_ok;
  __shared__ double red[4][6];
  __shared__ int redc[4];
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) {
    double T[12], Ti[12];
    const bool ok = hypothesis(a, a.ids + (size_t)h * 4, T, Ti, sM);
    s_ok = ok;
    if (ok) for (int k = 0; k < 12; ++k) { sT[k] = T[k]; sTi[k] = Ti[k]; }
  }
  __syncthreads();
  bool ok = s_ok != 0;
  if (ok && a.use_dims) {          // extent of the target cloud in the canonical frame (aligning.py:57-61)
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int p = tid; p < a.N; p += 256) {
      const double x = a.dst[p * 3], y = a.dst[p * 3 + 1], z = a.dst[p * 3 + 2];
      for (int i = 0; i < 3; ++i) { const double c = sTi[i * 4] * x + sTi[i * 4 + 1] * y + sTi[i * 4 + 2] * z + sTi[i * 4 + 3]; lo[i] = fmin(lo[i], c); hi[i] = fmax(hi[i], c); }
    }
    for (int i = 0; i < 3; ++i) { lo[i] = wave_min(lo[i]); hi[i] = wave_max(hi[i]); }
    if (lane == 0) for (int i = 0; i < 3; ++i) { red[wv][i] = lo[i]; red[wv][3 + i] = hi[i]; }
    __syncthreads();
    for (int i = 0; i < 3; ++i) {
      const double l = fmin(fmin(red[0][i], red[1][i]), fmin(red[2][i], red[3][i]));
      const double u = fmax(fmax(red[0][3 + i], red[1][3 + i]), fmax(red[2][3 + i], red[3][3 + i]));
      if (u - l > a.max_dims[i]) ok = false;
    }
  }
  int cnt = 0;
  if (ok) {
    for (int p = tid; p < a.N; p += 256) {
      const double x = a.src[p * 3], y = a.src[p * 3 + 1], z = a.src[p * 3 + 2];
      double e2 = 0.0;
      for (int i = 0; i < 3; ++i) { const double c = sT[i * 4] * x + sT[i * 4 + 1] * y + sT[i * 4 + 2] * z + sT[i * 4 + 3] - a.dst[p * 3 + i]; e2 += c * c; }
      cnt += (sqrt(e2) <= a.thres) ? 1 : 0;
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) redc[wv] = cnt;
  }
  __syncthreads();
  if (tid == 0) {
    a.counts[h] = ok ? (redc[0] + redc[1] + redc[2] + redc[3]) : -1;
    double* o = a.transforms + (size_t)h * 16;
    for (int k = 0; k < 12; ++k) o[k] = ok ? sT[k] : 0.0;
    o[12] = 0.0; o[13] = 0.0; o[14] = 0.0; o[15] = ok ? 1.0 : 0.0;
  }
}

__global__ void similarity_inliers_kernel(const double* __restrict__ src, const double* __restrict__ dst, int N, const double* __restrict__ T,
                                          double thres, unsigned char* __restrict__ mask) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const double x = src[p * 3], y = src[p * 3 + 1], z = src[p * 3 + 2];
  double e2 = 0.0;
  for (int i = 0; i < 3; ++i) { const double c = T[i * 4] * x + T[i * 4 + 1] * y + T[i * 4 + 2] * z + T[i * 4 + 3] - dst[p * 3 + i]; e2 += c * c; }
  mask[p] = sqrt(e2) <= thres ? 1 : 0;
}

}  // namespace

extern "C" int cg_ransac_9d(const double* src, const double* dst, int N, const int* ids, int H, double threshold,
                            const double* h_min_scale, const double* h_max_scale, const double* h_max_dimensions,
                            int* counts, double* transforms, void* stream) {
  if (N < 0 || H < 0 || !(threshold >= 0.0) || !h_min_scale || !h_max_scale) return CG_ERR_ARG;
  if (H == 0) return CG_OK;
  if (!src || !dst || !ids || !counts || !transforms || N < 4) return CG_ERR_ARG;
  RansacArgs a;
  a.src = src; a.dst = dst; a.N = N; a.ids = ids; a.H = H; a.thres = threshold;
  for (int i = 0; i < 3; ++i) { a.min_scale[i] = h_min_scale[i]; a.max_scale[i] = h_max_scale[i]; a.max_dims[i] = h_max_dimensions ? h_max_dimensions[i] : 0.0; }
  a.use_dims = h_max_dimensions != nullptr;
  a.counts = counts; a.transforms = transforms;
  hipLaunchKernelGGL(ransac_9d_kernel, dim3((unsigned)H), dim3(256), 0, (hipStream_t)stream, a);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_similarity_inliers(const double* src, const double* dst, int N, const double* transform16, double threshold,
                                     unsigned char* mask, void* stream) {
  if (N < 0) return CG_ERR_ARG;
  if (N == 0) return CG_OK;
  if (!src || !dst || !transform16 || !mask) return CG_ERR_ARG;
  hipLaunchKernelGGL(similarity_inliers_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, N,
                     transform16, threshold, mask);
  return cg_hip_status(hipGetLastError());
}
