/* catgrasp_amd -- C ABI of the MI355X-native grasp-candidate scoring hot path.
 *
 * libcatgrasp_amd.so exports exactly these entry points.  Every pointer is a DEVICE pointer
 * (HBM) unless the parameter name starts with `h_`; sizes are element counts; `stream` is a
 * hipStream_t passed as void*.  All functions are asynchronous on `stream`, never allocate,
 * never synchronise, never throw; they return 0 (CG_OK), a negative CG_ERR_* for argument
 * errors, or a positive hipError_t.
 *
 * Each entry point names the reference interface (wenbowen123/catgrasp @ v1, file:line) it
 * replaces.  INTEGRATION.md shows the reference-side bindings (ctypes) a maintainer adds.
 */
#ifndef CATGRASP_AMD_H
#define CATGRASP_AMD_H
#ifdef __cplusplus
extern "C" {
#endif

#define CG_OK 0
#define CG_ERR_ARG (-1)
#define CG_ERR_UNSUPPORTED (-2)

/* library / build identification (host). */
const char* cg_version(void);

/* ---------------------------------------------------------------------------------------------
 * PointNet networks (pointnet2.py:153-329).  Weights are BatchNorm-folded (eval) and, for every
 * K>=64 layer, packed into MFMA B-fragment order by the host (catgrasp_amd/folding.py):
 *   Wp[nb][ks][lane][j] = W[nb*32 + (lane&31)][ks*8 + (lane>>5)*4 + j],  zero padded to 32 rows.
 * ------------------------------------------------------------------------------------------- */

/* Fused shared per-point MLP chain + max over points: replaces
 *   STN3d.forward conv1..conv3+max        pointnet2.py:172-176   (mid_mode 0, t3 = NULL, relu3 = 1)
 *   STNkd.forward conv1..conv3+max        pointnet2.py:210-214   (mid_mode 1: wm/bm = fstn.conv1, on top of
 *                                          encoder conv1 :243-252 with t3 = learned 3x3 input transform)
 *   PointNetEncoder conv1,bmm,conv2,conv3,max  pointnet2.py:243-266 (mid_mode 2: t64 = learned 64x64
 *                                          feature transform; relu3 = 0)
 * x: (B,N,6) f32.  t3: (B,9) or NULL.  w1: (64,6) row-major, b1: (64).  t64: (B,64,64), h' = h.T.
 * out: (B,1024).  pointfeat (optional, mid_mode 2): (B,N,64) = transformed point features
 * (PointNetEncoder `pointfeat`, pointnet2.py:261).  nsplit: workgroups per sample (>=1); the point
 * tiles of one sample are divided between them and combined with atomic max. */
int cg_pointmlp_max(const float* x, int B, int N, const float* t3, const float* w1, const float* b1,
                    int mid_mode, const float* wm_packed, const float* bm, const float* t64,
                    const float* w2_packed, const float* b2, const float* w3_packed, const float* b3,
                    int relu3, int nsplit, float* out, float* pointfeat, void* stream);

/* Y[M,N] = act(X[M,K] . W^T + bias + row_bias[row / rows_per_group]) (+ flattened identity k x k):
 * replaces Linear->BN->ReLU tails (pointnet2.py:178-185, :216-223, :295-298) and the Conv1d(k=1)
 * segmentation head (pointnet2.py:324-328).  K % 8 == 0, ldx % 4 == 0, x 16-byte aligned. */
int cg_gemm_bias_act(const float* x, int M, int K, int ldx, const float* w_packed, int N,
                     const float* bias, const float* row_bias, int rows_per_group, int ld_rb,
                     int relu, int eye_k, float* y, int ldy, void* stream);

/* softmax / argmax / confidence (predicter.py:86-91) and p_G = sum_k p_k * k / C
 * (run_grasp_simulation.py:313).  logits (B,C) -> probs (B,C), label (B) i32, conf (B), p_g (B). */
int cg_softmax_pg(const float* logits, int B, int C, float* probs, int* label, float* conf, float* p_g,
                  void* stream);

/* NUNOCS bin decode (predicter.py:144-150): logits (P, 3*nbins) -> coords (P,3) = argmax/nbins - 0.5,
 * conf_z (P) = softmax probability of the arg-max z bin. */
int cg_nunocs_decode(const float* logits, long P, int nbins, float* coords, float* conf_z, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Per-sample input transforms (host python loops in the reference, predicter.py:71-74).
 * ------------------------------------------------------------------------------------------- */

/* GraspDataset.transform (dataset_grasp.py:63-91) for G candidates at once.
 * cloud_xyz/cloud_normal: (n_cloud,3) f32 (z>=0.1 filtered, object-centred by the host);
 * ids: (G,n_pts) i32 resample indices into the cloud; pose_inv: (G,12) rows of inv(grasp_pose)[:3,:4]
 * (re-expressed for the centred cloud); mean / inv_std: (6) or both NULL.  out: (G,n_pts,6). */
int cg_build_grasp_input(const float* cloud_xyz, const float* cloud_normal, int n_cloud, const int* ids,
                         const float* pose_inv, const float* mean, const float* inv_std, int G, int n_pts,
                         float* out, void* stream);

/* NunocsIsolatedDataset.transform + NormalizeCloud (dataset_nunocs.py:38-65, augmentations.py:66-75)
 * for B object clouds: ids (B,n_pts) i32 into the shared cloud arrays; out (B,n_pts,6). */
int cg_build_nunocs_input(const float* cloud_xyz, const float* cloud_normal, int n_cloud, const int* ids,
                          const float* mean, const float* inv_std, int B, int n_pts, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
