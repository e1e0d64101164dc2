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
#include <stddef.h>

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
 * x: (B,N,6) f32.  t3: (B,9) or NULL.  w1: (64,6) row-major, b1: (64).  t64: (B,64,64) holding the feature
 * transform TRANSPOSED, t64[b][n][k] = T_b[k][n] (h' = h.T_b); the host permutes fstn.fc3 so the FC kernel emits it so.
 * out: (B,1024).  pointfeat (optional, mid_mode 2): (B,N,64) = transformed point features
 * (PointNetEncoder `pointfeat`, pointnet2.py:261).  nsplit: workgroups per sample (>=1); the point
 * tiles of one sample are divided between them and combined with atomic max. */
int cg_pointmlp_max(const float* x, int B, int N, const float* t3, const float* w1, const float* b1,
                    int mid_mode, const float* wm_packed, const float* bm, const float* t64,
                    const float* w2_packed, const float* b2, const float* w3_packed, const float* b3,
                    int relu3, int nsplit, float* out, float* pointfeat, void* stream);

/* Split-precision ("bf16x3") variant of cg_pointmlp_max: every contraction is three bf16 MFMAs with f32
 * accumulation (x_lo.w_hi + x_hi.w_lo + x_hi.w_hi); logits agree with the exact-f32 path to ~1e-5.
 * Weights are split on the host (folding.pack_b_bf16x3): Wp[nb][kc][2 (hi,lo)][lane][8] bf16 with
 * element e of lane l = W[nb*32 + (l&31)][kc*16 + (l>>5)*8 + e].  tile_points: points per workgroup tile; must be 256
 * (8 waves, one workgroup per CU; other values return CG_ERR_UNSUPPORTED).  w1 / b1 stay plain f32 (split in the
 * kernel).  Other arguments as cg_pointmlp_max. */
int cg_pointmlp_max_bf16x3(const float* x, int B, int N, const float* t3, const float* w1, const float* b1,
                           int mid_mode, const unsigned short* wm_split, const float* bm, const float* t64,
                           const unsigned short* w2_split, const float* b2, const unsigned short* w3_split,
                           const float* b3, int relu3, int nsplit, int tile_points, float* out, float* pointfeat,
                           void* stream);
/* Same with IEEE-half pieces ("f16x3", 11 + 11 significant bits instead of 8 + 8): logits within ~2e-6 of the float64 evaluation,
 * i.e. float32's own distance, at the same three MFMAs per product block.  Weights packed by folding.pack_b_split(w, 'f16').  OPT-IN
 * (CATGRASP_AMD_PRECISION=f16x3): the engine's default arithmetic is the exact-f32 cg_pointmlp_max.  The half pieces have a limited exponent range; `status` (optional device int, owned and
 * zeroed by the caller, one per call or per batch -- there is no process-global state) gets CG_STATUS_HALF_OVERFLOW OR-ed in if
 * a value handed to the split reached 65504 (the result is then meaningless) and CG_STATUS_HALF_UNDERFLOW if a whole layer
 * output of some 32-point tile (cg_gemm_bias_act_f16x3: the whole 128-row X tile of some workgroup) stayed below 2^-6 (its low pieces then sit in the half subnormals: absolute instead of relative
 * error).  On either bit re-run the batch with the bf16x3 (float32 exponent range) or f32 entry point. */
#define CG_STATUS_HALF_OVERFLOW 1
#define CG_STATUS_HALF_UNDERFLOW 2
int cg_pointmlp_max_f16x3(const float* x, int B, int N, const float* t3, const float* w1, const float* b1,
                          int mid_mode, const unsigned short* wm_split, const float* bm, const float* t64,
                           const unsigned short* w2_split, const float* b2, const unsigned short* w3_split,
                           const float* b3, int relu3, int nsplit, int tile_points, float* out, float* pointfeat,
                           int* status, void* stream);
/* "f16fp8x2": cg_pointmlp_max_f16x3 with a cheaper 128 -> 1024 layer (91 % of the pass's matrix work): its main term stays one f16 MFMA
 * per 16 input channels, its two correction terms x_hi.w_lo + x_lo.w_hi run on the block-scaled e4m3 matrix instruction of gfx950
 * (v_mfma_scale_f32_32x32x64_f8f6f4; one power-of-two scale per 32 input channels of a point / of an output channel, applied by the
 * instruction, f32 accumulation into the same registers) -- 128 instead of 192 matrix passes per 32x32x128 product block.  The e4m3
 * rounding of the OTHER factor of each correction term (2^-4 relative, on a term that is 2^-11 of the product) puts the logits within
 * ~5e-5 of the float64 evaluation (parity bar 1e-4).  w3_mx: folding.pack_b_f16fp8x2 (16,640 B per 32 output channels); every
 * other argument, the front layers (f16x3) and `status` as cg_pointmlp_max_f16x3.  EXPERIMENTAL, opt-in
 * (CATGRASP_AMD_PRECISION=f16fp8x2): the mode is specified on the networks' logits / probabilities; on raw encoder features it measures
 * ~1.1e-4 of the feature scale, so the free-standing STN3d / PointNetEncoder modules run f16x3 under it (engine.run_guarded_features). */
int cg_pointmlp_max_f16fp8x2(const float* x, int B, int N, const float* t3, const float* w1, const float* b1,
                             int mid_mode, const unsigned short* wm_split, const float* bm, const float* t64,
                             const unsigned short* w2_split, const float* b2, const void* w3_mx,
                             const float* b3, int relu3, int nsplit, int tile_points, float* out, float* pointfeat,
                             int* status, void* stream);

/* Y[M,N] = act(X[M,K] . W^T + bias + row_bias[row / rows_per_group]) (+ flattened identity k x k):
 * replaces Linear->BN->ReLU tails (pointnet2.py:178-185, :216-223, :295-298) and the Conv1d(k=1)
 * segmentation head (pointnet2.py:324-328).  K % 8 == 0, ldx % 4 == 0, x 16-byte aligned. */
int cg_gemm_bias_act(const float* x, int M, int K, int ldx, const float* w_packed, int N,
                     const float* bias, const float* row_bias, int rows_per_group, int ld_rb,
                     int relu, int eye_k, float* y, int ldy, void* stream);

/* PointNetCls.forward (pointnet2.py:289-299; eval mode, exact-f32 kernels) as ONE call: the three cg_pointmlp_max passes and nine
 * cg_gemm_bias_act layers issued back to back -- the launch chain of a predict_batch call of a few poses without ~10 us of caller-side
 * work per launch.  Same kernels, arguments and order as issuing them one by one: identical results.
 * w: HOST struct of DEVICE pointers to the BatchNorm-folded, packed weights (catgrasp_amd.folding.prepare_cls; names as there).
 * ws: caller-owned device workspace of cg_pointnet_cls_workspace_floats(B) floats, 16-byte aligned.  logits: (B, n_out).
 * *trans_feat_t (optional, HOST pointer to a device pointer): receives the address, inside ws, of the (B,64,64) feature transform,
 * TRANSPOSED like cg_pointmlp_max's t64 (PointNetCls returns it as trans_feat, pointnet2.py:299). */
typedef struct cg_cls_weights {
  const float *stn_w1, *stn_b1, *stn_w2, *stn_b2, *stn_w3, *stn_b3, *stn_fc1, *stn_fc1b, *stn_fc2, *stn_fc2b, *stn_fc3, *stn_fc3b;
  const float *enc_w1, *enc_b1, *fstn_wm, *fstn_bm, *fstn_w2, *fstn_b2, *fstn_w3, *fstn_b3;
  const float *fstn_fc1, *fstn_fc1b, *fstn_fc2, *fstn_fc2b, *fstn_fc3, *fstn_fc3b;
  const float *enc_w2, *enc_b2, *enc_w3, *enc_b3;
  const float *head_fc1, *head_fc1b, *head_fc2, *head_fc2b, *head_fc3, *head_fc3b;
  int n_out;
} cg_cls_weights;
size_t cg_pointnet_cls_workspace_floats(int B);
int cg_pointnet_cls_forward(const float* x, int B, int N, const cg_cls_weights* h_weights, int nsplit, float* workspace, float* logits,
                            float** trans_feat_t, void* stream);

/* Split-precision ("bf16x3") variant of cg_gemm_bias_act for the wide FC tails / segmentation head: every product block
 * is three bf16 MFMAs with f32 accumulation, X is split on the fly, W is split-packed on the host
 * (folding.pack_b_bf16x3, same layout as cg_pointmlp_max_bf16x3).  K % 16 == 0; other arguments as cg_gemm_bias_act. */
int cg_gemm_bias_act_bf16x3(const float* x, int M, int K, int ldx, const unsigned short* w_split, int N,
                            const float* bias, const float* row_bias, int rows_per_group, int ld_rb,
                            int relu, int eye_k, float* y, int ldy, void* stream);
/* IEEE-half variant of the above; `status` as in cg_pointmlp_max_f16x3 (the split operand here is X). */
int cg_gemm_bias_act_f16x3(const float* x, int M, int K, int ldx, const unsigned short* w_split, int N,
                            const float* bias, const float* row_bias, int rows_per_group, int ld_rb,
                            int relu, int eye_k, float* y, int ldy, int* status, void* stream);

/* Max-pool over points of a MATERIALISED activation tensor: x (groups * rows_per_group, C) -> out (groups, C), out[g][c] = max over
 * the rows of group g (torch.max(x, 2)[0] of pointnet2.py:176,214).  The fused passes above never materialise it; this serves the
 * free-standing STNkd module (pointnet2.py:189-223), whose input is an arbitrary k-channel tensor. */
int cg_group_max(const float* x, long groups, long rows_per_group, int C, float* out, void* stream);

/* softmax / argmax / confidence (predicter.py:86-91) and p_G = sum_k p_k * k / C
 * (run_grasp_simulation.py:313).  logits (B,C) -> probs (B,C), label (B) i32, conf (B), p_g (B). */
int cg_softmax_pg(const float* logits, int B, int C, float* probs, int* label, float* conf, float* p_g,
                  void* stream);

/* NUNOCS bin decode (predicter.py:144-150): logits (P, 3*nbins) -> coords (P,3) = argmax/nbins - 0.5,
 * conf_z (P) = softmax probability of the arg-max z bin.  nbins <= 128 (config_nunocs.yml: 100). */
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


/* ---------------------------------------------------------------------------------------------
 * Collision filter: my_cpp (my_cpp/pybind.cpp:11-23).
 * A point cloud registered at `resolution` (CollisionManager::registerPointCloud,
 * my_cpp/collision_manager.cpp:55-79) is the set of occupied octomap depth-16 leaves; it is held in
 * HBM as (n,4) int16 rows (key-32768 per axis, 4th lane unused), unique and sorted.
 * ------------------------------------------------------------------------------------------- */

/* octomap coordToKeyChecked per point: packed[i] = kx<<32 | ky<<16 | kz (keys in [0,65536)) or -1 for a
 * point outside the tree (updateNode ignores it).  The caller de-duplicates (sort/unique). */
int cg_voxel_keys(const float* pts, long n_pts, float resolution, long long* packed, void* stream);
/* packed unique keys -> (n,4) int16 rows. */
int cg_unpack_voxel_keys(const long long* packed, long n, short* keys4, void* stream);

/* CollisionManager::setTransform + isAnyCollision (my_cpp/collision_manager.cpp:81-111) for n_poses
 * independent poses (n_poses,16 row-major 4x4) of one triangle mesh (registerMesh, :15-52) against one
 * voxelised cloud.  out[i] = 1 iff some occupied leaf box intersects some posed triangle. */
int cg_mesh_voxels_collide(const float* vertices, const int* faces, int n_faces, const float* poses, long n_poses,
                           const short* keys, int n_keys, float resolution, unsigned char* out, void* stream);

/* The other pairs of CollisionManager::isAnyCollision (my_cpp/collision_manager.cpp:93-111 loops over EVERY pair of registered
 * objects; the grasp filter itself only forms {mesh, cloud}).  Both write *out = 1 iff the pair collides, 0 otherwise.
 *  cg_mesh_mesh_collide     : two triangle meshes (vertices (nv,3) f32, faces (nf,3) i32), each under its own pose (device, 16 floats
 *                             row-major 4x4: setTransform): some closed triangle of A meets some closed triangle of B
 *                             (FCL: BVHModel<OBBRSSf> against BVHModel<OBBRSSf>, collision_manager.cpp:41-45).
 *  cg_voxels_voxels_collide : two voxelised clouds (keys (n,4) int16 as cg_unpack_voxel_keys writes them, each with its own
 *                             resolution); b_in_a (device, 16 floats) = inv(pose A) . pose B, rigid: some leaf cube of A meets some
 *                             leaf cube of B (fcl::OcTree against fcl::OcTree, collision_manager.cpp:63-70). */
int cg_mesh_mesh_collide(const float* vertices_a, const int* faces_a, int n_faces_a, const float* vertices_b, const int* faces_b,
                         int n_faces_b, const float* pose_a, const float* pose_b, unsigned char* out, void* stream);
int cg_voxels_voxels_collide(const short* keys_a, int n_keys_a, float resolution_a, const short* keys_b, int n_keys_b,
                             float resolution_b, const float* b_in_a, unsigned char* out, void* stream);

/* Optional broad phase for a gripper mesh: a uniform grid in the MESH frame (host descriptor, device arrays).
 * Cell (i,j,k) -> index (i*dims[1] + j)*dims[2] + k; tri_ids[cell_start[c] .. cell_start[c+1]) lists every triangle whose
 * bounding box inflated by 2*(resolution*sqrt(3)/2) + slack overlaps the cell.  Valid for point clouds registered at
 * exactly `resolution` and poses with sigma_min >= 0.5; other calls use the exhaustive path.  Results are identical. */
typedef struct cg_mesh_grid {
  float origin[3]; float cell; int dims[3];
  const int* cell_start;   /* device, prod(dims)+1 */
  const int* tri_ids;      /* device */
  float resolution;
  const float* tri_verts;  /* device: the triangles as (n_faces,12) float32 [v0.xyz v1.xyz v2.xyz 0 0 0], 16-byte aligned -- one
                              dependent load less per narrow-phase test (a grid without it is not used) */
  const unsigned char* coarse_occupancy;  /* device, optional: (ceil(dims/4)) bytes, z fastest: 1 iff some cell of the 4 x 4 x 4 block
                              of cells has a non-empty list -- lets the kernel skip whole blocks of voxels (see open_blocks) */
} cg_mesh_grid;

/* filterGraspPose (my_cpp/common.cpp:156-321; declaration my_cpp/common.h:60) for every
 * (grasp pose i, symmetry transform j) pair, in input order e = i*n_sym + j.
 *  grasp_poses (n_pose,16), symmetry_tfs (n_sym,16): device, row-major float32 4x4, 16-byte aligned (as are poses_out / ee_in_base_out).
 *  h_*: HOST pointers to 16 floats (row-major 4x4): nocs_pose, canonical_to_nocs_transform, cam_in_world,
 *       ee_in_grasp, gripper_in_grasp.
 *  ik_ok: optional (E) u8 produced by the host IK pass (0 => rejected with code 2); NULL => filter_ik=false.
 *  gripper / enclosed mesh: vertices (nv,3) f32, faces (nf,3) i32 (registerMesh inputs, common.cpp:177,181).
 *  open_keys / bg_keys: voxelised gripper_collision_pts / gripper_enclosed_collision_pts (common.cpp:178,182).
 * Outputs: codes (E) i8 {0 keep, 1 approach-dir, 2 IK, 3 open-gripper collision or no nudge found,
 *  4 enclosed-gripper collision}; poses_out (E,16) surviving grasp_in_cam (after column normalisation and
 *  nudge; zero if rejected); nudge (E) i8 accepted nudge index {0:0, 1:+1mm, 2:-1mm, 3:+2mm, 4:-2mm} or -1.
 * If ee_in_base_out != NULL the call only runs the pre-IK stage: it writes ee_in_base = cam_in_world .
 * grasp_in_cam . ee_in_grasp (E,16) (common.cpp:216) and codes {0,1}, for the host-side IK pass. */
int cg_filter_grasp_pose(const float* grasp_poses, int n_pose, const float* symmetry_tfs, int n_sym,
                         const float* h_nocs_pose, const float* h_canonical_to_nocs, const float* h_cam_in_world,
                         const float* h_ee_in_grasp, const float* h_gripper_in_grasp,
                         int filter_approach_dir_face_camera, int adjust_collision_pose,
                         const unsigned char* ik_ok,
                         const float* gripper_vertices, const int* gripper_faces, int n_gripper_faces,
                         const float* enclosed_vertices, const int* enclosed_faces, int n_enclosed_faces,
                         const short* open_keys, int n_open_keys, const short* bg_keys, int n_bg_keys,
                         float resolution, signed char* codes, float* poses_out, signed char* nudge,
                         float* ee_in_base_out, void* stream);
/* Same, with optional broad-phase grids (HOST descriptors, NULL = exhaustive) for the open / enclosed gripper mesh.
 * keep_rejected_pose != 0: poses_out of a REJECTED evaluation holds its composed, column-normalised (un-nudged) grasp_in_cam
 * (common.cpp:191-197) instead of zeros, so a fixed-size batch can be scored without a compaction step.
 * work_stats: optional DEVICE pointer to 3 uint64 counters the grid kernel ADDS to (measurement only): voxel keys read (8 B each),
 * grid cells looked up (8 B each), (voxel, triangle) pairs run through the narrow phase (48 B of triangle each) -- the
 * cache-level byte count bench.py's roofline_filter block is priced on.
 * open_blocks / bg_blocks: optional DEVICE (ceil(n_keys/64), 2, 4) int16 -- lowest and highest key (per axis) of every run of 64
 * consecutive keys of the voxel set.  With keys in a space-filling order (my_cpp.GripperScene sorts them by Morton code) a run is a
 * compact blob, and the grid kernel skips runs whose box lies outside the grid or over empty coarse cells: a pure accelerator, the
 * set of (voxel, triangle) pairs that reach the narrow phase is unchanged.
 * Three launches: pose composition (one thread per evaluation), the grid kernel (one wavefront per live evaluation), and the
 * exhaustive kernel, which only finishes evaluations whose pose a grid does not cover (all of them when there are no grids). */
int cg_filter_grasp_pose_accel(const float* grasp_poses, int n_pose, const float* symmetry_tfs, int n_sym,
                               const float* h_nocs_pose, const float* h_canonical_to_nocs, const float* h_cam_in_world,
                               const float* h_ee_in_grasp, const float* h_gripper_in_grasp,
                               int filter_approach_dir_face_camera, int adjust_collision_pose,
                               const unsigned char* ik_ok,
                               const float* gripper_vertices, const int* gripper_faces, int n_gripper_faces,
                               const float* enclosed_vertices, const int* enclosed_faces, int n_enclosed_faces,
                               const short* open_keys, int n_open_keys, const short* bg_keys, int n_bg_keys,
                               float resolution, signed char* codes, float* poses_out, signed char* nudge,
                               float* ee_in_base_out, const cg_mesh_grid* h_open_grid, const cg_mesh_grid* h_enc_grid,
                               int keep_rejected_pose, unsigned long long* work_stats, const short* open_blocks, const short* bg_blocks,
                               void* stream);

/* Several filterGraspPose calls as ONE launch sequence.  A pick cycle filters every object of a scene twice -- the cone sampler's poses
 * with symmetry_tfs = [I] (dexnet/grasping/grasp_sampler.py:216) and the canonical grasps x the category's symmetries with
 * adjust_collision_pose = True (:345) -- each against that object's own two voxel sets (run_grasp_simulation.py:131-141): a dozen or more
 * calls of a few thousand evaluations, three launches per call, none of which fills 256 CUs.  A SEGMENT is one such call's
 * (grasp_poses x symmetry_tfs) block with its own nocs_pose / canonical_to_nocs, adjust flag and voxel sets; the gripper meshes, their
 * grids, gripper_in_grasp, the resolution and the approach-direction flag are common to the launch (they are in the reference too:
 * one gripper per run).  Evaluations are numbered segment after segment, e = first + i * n_sym + j; codes / poses_out / nudge
 * (and ik_ok) are indexed by that e.  Every evaluation's result is bit-identical to its own cg_filter_grasp_pose_accel call.
 *   cg_filter_segments_prepare (HOST, no device work): validates the table, composes c2c = nocs_pose . canonical_to_nocs with the
 *     float32 operation order of the single-call path, writes `first`; returns the total number of evaluations or a negative cg error.
 *   cg_filter_grasp_pose_multi: h_segments = the prepared HOST table, d_segments = a DEVICE copy of those same n_segments rows (the
 *     caller uploads it however it likes -- a cached upload costs nothing per call -- and keeps it alive until the stream has passed
 *     the call).  No pre-IK stage here (ee_in_base): a caller that filters by IK runs that stage per segment and passes ik_ok. */
typedef struct cg_filter_segment {
  const float* grasp_poses;        /* device (n_pose,16), 16-byte aligned */
  const float* symmetry_tfs;       /* device (n_sym,16), 16-byte aligned */
  int n_pose, n_sym;
  float nocs_pose[16];             /* row-major 4x4 */
  float canonical_to_nocs[16];
  float c2c[16];                   /* written by cg_filter_segments_prepare */
  int adjust_collision_pose;
  int n_open_keys;
  const short* open_keys;          /* device: this segment's voxelised gripper_collision_pts (cg_voxel_keys -> cg_unpack_voxel_keys) */
  const short* open_blocks;        /* device, optional: key box of every run of 64 keys (see cg_filter_grasp_pose_accel) */
  const short* bg_keys;            /* device: voxelised gripper_enclosed_collision_pts */
  const short* bg_blocks;
  int n_bg_keys;
  int reserved;
  long long first;                 /* written by cg_filter_segments_prepare: index of the segment's first evaluation */
} cg_filter_segment;
long cg_filter_segments_prepare(cg_filter_segment* h_segments, int n_segments);
int cg_filter_grasp_pose_multi(const cg_filter_segment* h_segments, const cg_filter_segment* d_segments, int n_segments,
                               const float* h_gripper_in_grasp, int filter_approach_dir_face_camera, const unsigned char* ik_ok,
                               const float* gripper_vertices, const int* gripper_faces, int n_gripper_faces,
                               const float* enclosed_vertices, const int* enclosed_faces, int n_enclosed_faces,
                               float resolution, signed char* codes, float* poses_out, signed char* nudge,
                               const cg_mesh_grid* h_open_grid, const cg_mesh_grid* h_enc_grid, int keep_rejected_pose,
                               unsigned long long* work_stats, void* stream);

/* Device build of cg_mesh_grid (replaces a per-triangle host loop): triangle t is listed in every cell its bounding box,
 * inflated by `inflate`, overlaps (float64 cell arithmetic).  h_origin[3], h_dims[3]: HOST.  Two passes around a host-side
 * exclusive prefix sum: cg_mesh_grid_count adds into counts (prod(dims), pre-zeroed); cg_mesh_grid_fill writes tri_ids
 * (cell_start[prod(dims)] entries) using cursor (prod(dims), pre-zeroed) and then sorts every cell's list ascending. */
int cg_mesh_grid_count(const float* vertices, const int* faces, int n_faces, const double* h_origin, double cell, double inflate,
                       const int* h_dims, int* counts, void* stream);
int cg_mesh_grid_fill(const float* vertices, const int* faces, int n_faces, const double* h_origin, double cell, double inflate,
                      const int* h_dims, const int* cell_start, int* cursor, int* tri_ids, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Host-loop replacements around the two networks (the reference-API entry points were host-capped).
 * ------------------------------------------------------------------------------------------- */

/* The per-candidate resampling draw of GraspDataset.transform (dataset_grasp.py:72-73: np.random.choice(M, n_pts,
 * replace = M < n_pts), once per pose in the loop of predicter.py:71-74) for `count` candidates at once:
 * out (count,n_pts) i32 = base + an n_pts-subset of [0,n_valid) in random order (n_valid >= n_pts: the first n_pts values of a keyed
 * permutation of [0,n_valid) -- a 12-round Feistel bijection of the covering power of four, cycle-walked into the range, round keys
 * from Philox; up to 1,024 points a workgroup per row sorts 48-bit random keys instead) or iid uniform indices (n_valid < n_pts,
 * n_pts % 4 == 0, else CG_ERR_UNSUPPORTED).
 * Counter-based Philox4x32-10 keyed by `seed` and the global row index row_offset + r: reproducible, independent of how a
 * batch is split into calls (a GPU shard draws exactly what the unsharded batch would), but NOT numpy's stream. */
int cg_draw_resample_ids(int n_valid, int n_pts, long count, unsigned long long seed, int base, long row_offset, int* out,
                         void* stream);

/* HOST function (all pointers host memory; no device work): the same per-candidate draw REPLAYED FROM NUMPY'S OWN STREAM, for
 * seeded parity with the reference.  h_mt_key624 / h_mt_pos: the Mersenne-Twister state of numpy's global RandomState as
 * np.random.get_state() returns it (624 uint32 words, position 0..624); updated in place so the caller can np.random.set_state()
 * it back.  Emits, for `count` candidates, exactly what `np.random.choice(np.arange(n_valid), n_pts, replace = n_valid < n_pts)`
 * would have returned call after call (permutation(n_valid)[:n_pts], or randint(0, n_valid, n_pts)), ~3x faster than numpy and
 * outside the GIL.  h_scratch: n_valid ints (replace=False branch).  h_out: (count, n_pts) int32. */
int cg_host_numpy_choice_rows(unsigned int* h_mt_key624, int* h_mt_pos, int n_valid, int n_pts, long count, int* h_scratch, int* h_out);

/* The hypothesis draw of the 9-D RANSAC (aligning.py:89-93: `np.random.choice(len(source), size=4, replace=False)` per iteration,
 * 2 x 10,000 iterations per object, predicter.py:167-170) replayed from numpy's stream: `count` consecutive draws of
 * permutation(n)[:k], 2 <= n <= 65536, 1 <= k <= min(n,16), into h_out (count,k) int32; the generator state advances exactly as
 * numpy's would.  Nothing but the k heads is materialised (csrc/nprng_heads.hip): vectorised generator blocks, a 16-word
 * rejection walk, and the swaps undone for the k tracked positions only.  isa 0 = AVX-512 when the CPU has it, 1 = the scalar twin.  HOST pointers, no device work. */
int cg_host_numpy_choice_heads(unsigned int* h_mt_key624, int* h_mt_pos, int n, int k, long count, int isa, int* h_out);

/* Whole rows of permutation(n)[:n_pts] on the host (2 <= n <= 65536, 1 <= n_pts <= n): the vectorised partner extraction of
 * cg_host_numpy_choice_heads + the swap chain in L1, ~2x faster than cg_host_numpy_choice_rows.  Used for SMALL draws (a predict_batch
 * call of a few poses), where shipping partners to the device would wait ~250 us for one lane's swap chain.  HOST pointers. */
int cg_host_numpy_permutation_rows(unsigned int* h_mt_key624, int* h_mt_pos, int n, int n_pts, long count, int isa, int* h_out);

/* The same draw with the swap chain on the device (n_pts <= n_valid <= 65536, the replace=False branch): the HOST part is only
 * what makes numpy's stream sequential -- the rejection-sampled Fisher-Yates swap partners j(i), i = n_valid-1 .. 1, of `count`
 * consecutive permutation(n_valid) calls, written as u16 at h_partners + r*row_stride + (n_valid-1-i) (row_stride >= n_valid-1,
 * tail zero-filled); the generator state advances exactly as under cg_host_numpy_choice_rows.  HOST pointers. */
int cg_host_numpy_shuffle_partners(unsigned int* h_mt_key624, int* h_mt_pos, int n_valid, long count, long row_stride,
                                   unsigned short* h_partners);
/* ... and the DEVICE part: out (count,n_pts) i32 = base + permutation[:n_pts] of every row, the swap chain a[i] <-> a[j(i)] run
 * one row per lane in LDS.  partners: DEVICE copy of the host array above, 16-byte aligned, row_stride a multiple of 8. */
int cg_apply_shuffle_rows(const unsigned short* partners, long row_stride, int n_valid, int n_pts, long count, int base, int* out,
                          void* stream);

/* inv(grasp_pose) of dataset_grasp.py:69-70 for poses already on the device: poses (n,16) f32 row-major 4x4 with last
 * row 0 0 0 1 -> out (n,12) rows [R | t] with x_grasp = R x_centred + t, where x_centred = x_cam - h_center (HOST,
 * 3 doubles).  float64 arithmetic, rounded once (same contract as the host helper transforms.pose_inverse_rows). */
int cg_pose_inverse_rows(const float* poses, long n_poses, const double* h_center, float* out, void* stream);
/* The same for the caller's float64 poses (predict_batch's grasp_poses list, predicter.py:67: uploaded unconverted, so the inverse
 * is taken of the very numbers np.linalg.inv sees at dataset_grasp.py:69-70).  *bad_flag (optional device int, pre-zeroed) collects
 * bits: 1 a pose holds NaN / Inf; 2 a pose is singular (np.linalg.inv raises LinAlgError) or its inverse does not fit float32;
 * 4 a pose's last row is not 0 0 0 1 (the closed form inverts an affine matrix, np.linalg.inv the full 4x4). */
int cg_pose_inverse_rows_f64(const double* poses, long n_poses, const double* h_center, float* out, int* bad_flag, void* stream);

/* ---------------------------------------------------------------------------------------------
 * PointNet++ grouping primitives (pointnet2.py:14-149).  Index tensors are int64 like the reference's.
 * ------------------------------------------------------------------------------------------- */

/* square_distance (pointnet2.py:14-33): src (B,N,3), dst (B,M,3) -> out (B,N,M) = -2 s.d + |s|^2 + |d|^2. */
int cg_square_distance(const float* src, const float* dst, int B, int N, int M, float* out, void* stream);
/* The same for C-dimensional points, src (B,N,C), dst (B,M,C) (pointnet2.py:14-33 is generic in C; the live callers pass xyz):
 * sums over the channels in index order.  C == 3 forwards to cg_square_distance. */
int cg_square_distance_nd(const float* src, const float* dst, int B, int N, int M, int C, float* out, void* stream);

/* index_points (pointnet2.py:35-51): points (B,N,C), idx (B,S) [S may be S*K flattened] -> out (B,S,C).
 * *err_flag (device int, pre-zeroed) is set to 1 on an out-of-range index (the reference raises IndexError). */
int cg_index_points(const float* points, const long long* idx, int B, int N, int C, long S, float* out, int* err_flag,
                    void* stream);

/* farthest_point_sample (pointnet2.py:54-75) with the `torch.randint` start index (:66) as an explicit
 * input: xyz (B,N,3), start (B) -> out (B,npoint).  dist_scratch: (B,N) floats, required only if N > 24576. */
int cg_farthest_point_sample(const float* xyz, const long long* start, int B, int N, int npoint, float* dist_scratch,
                             long long* out, void* stream);

/* The same sampling, which also writes the sampled points themselves: out_xyz (B,npoint,3) = index_points(xyz, out), the
 * `new_xyz` of sample_and_group (pointnet2.py:110-112) -- the kernel holds each new centre's coordinates anyway, so the
 * set-abstraction layer needs no gather launch behind it.  Same samples as cg_farthest_point_sample. */
int cg_farthest_point_sample_xyz(const float* xyz, const long long* start, int B, int N, int npoint, float* dist_scratch,
                                 long long* out, float* out_xyz, void* stream);

/* query_ball_point (pointnet2.py:78-98): first `nsample` indices (ascending) with d^2 <= radius_sq, padded
 * with the first hit; an empty ball yields N in every slot, as the reference does.  out (B,S,nsample). */
int cg_query_ball_point(const float* xyz, const float* new_xyz, int B, int N, int S, float radius_sq, int nsample,
                        long long* out, void* stream);

/* sample_and_group tail (pointnet2.py:116-123): new_points (B,S,K,3+D) = cat(xyz[idx] - new_xyz, points[idx]);
 * optional grouped_xyz (B,S,K,3).  points may be NULL when D == 0. */
int cg_group_points(const float* xyz, const float* points, const float* new_xyz, const long long* idx, int B, int N, int S,
                    int K, int D, float* new_points, float* grouped_xyz, int* err_flag, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Signed distance field (meshpy Sdf3D, meshpy/meshpy/sdf.py:216-389).  grid: (nx,ny,nz) f32 row-major
 * (x slowest), the `data_[i][j][k]` layout produced by SdfFile._read_3d (meshpy/meshpy/sdf_file.py:59-87).
 * coords are in GRID units, component-major (B,3,N) like the reference's (3,N)/(B,3,N) arrays.
 * ------------------------------------------------------------------------------------------- */

/* mode 0: Sdf3D._signed_distance(coords, fast=False) trilinear (sdf.py:312-343);
 * mode 1: fast=True / _signed_distance_batch nearest voxel, round-half-even (sdf.py:318-321,351-357). */
int cg_sdf_lookup(const float* grid, int nx, int ny, int nz, const float* coords, long B, long N, int mode, float* out,
                  void* stream);
/* Sdf3D.is_any_points_inside (sdf.py:377-389): *flag (device int, pre-zeroed) |= any(sd[round(coords)] < 0) over
 * the points that fall inside the grid. */
int cg_sdf_any_inside(const float* grid, int nx, int ny, int nz, const float* coords, long N, int* flag, void* stream);
/* Per-candidate form (transform_pt_obj_to_grid_batch + is_any_points_inside, sdf.py:362-389): cam_to_grid (E,12)
 * rows [A|t] map camera-frame scene points pts (n_pts,3) into candidate e's gripper grid; out[e] = any inside. */
int cg_sdf_points_inside_batch(const float* grid, int nx, int ny, int nz, const float* cam_to_grid, long E,
                               const float* pts, int n_pts, unsigned char* out, void* stream);


/* augmentGraspPoses (my_cpp/common.cpp:118-153, declaration my_cpp/common.h:58): poses for R in
 * {R0} U {R0.directionVecToRotation(sphere_pt,(1,0,0)).Rx(q*inplane_rot_step)} (nearest rotation of each) and depths
 * d = 0, approach_step, ... : out ((1+n_sphere*n_rot)*n_depth, 16) row-major 4x4, rotation-major then depth.
 * h_R0 (9) and h_selected_point (3) are HOST pointers; sphere_pts (n_sphere,3) device.  n_rot / n_depth are the trip
 * counts of the reference's float loops (`x_rot<180`, `d<hand_depth`), evaluated by the caller. */
int cg_augment_grasp_poses(const float* h_R0, const float* h_selected_point, const float* sphere_pts, int n_sphere,
                           int n_rot, float inplane_rot_step, int n_depth, float approach_step, float init_bite,
                           float* out, void* stream);


/* makeOccupancyGridFromCloudScan (my_cpp/common.cpp:324-431, declaration my_cpp/common.h:61).
 * cg_occupancy_set_bits: occupied leaves (n,4) int16 (key-32768) -> dense bitmap over the key box
 *   [x0,x0+dx) x [y0,y0+dy) x [z0,z0+dz) (bit index ((kx-x0)*dy + ky-y0)*dz + kz-z0; `bits` pre-zeroed).
 * cg_occupancy_grid_rays: for every point of the (nx,ny,nz) lattice origin + i*resolution (x slowest) cast the
 *   octomap ray from the sensor origin (castRay, ignoreUnknownCells=true, max_range) and set keep[i] = 1 iff it hits
 *   an occupied leaf whose centre is not farther than the lattice point; lattice (nx*ny*nz,3) receives the coordinates. */
int cg_occupancy_set_bits(const short* keys4, long n_keys, unsigned int* bits, int x0, int y0, int z0, int dx, int dy,
                          int dz, void* stream);
int cg_occupancy_grid_rays(const unsigned int* bits, int x0, int y0, int z0, int dx, int dy, int dz, float origin_x,
                           float origin_y, float origin_z, float resolution, int nx, int ny, int nz, double max_range,
                           float* lattice, unsigned char* keep, void* stream);


/* ---------------------------------------------------------------------------------------------
 * NUNOCS -> camera 9-D pose RANSAC: aligning.estimate9DTransform (aligning.py:33-119; caller predicter.py:164).
 * ------------------------------------------------------------------------------------------- */

/* Evaluate H hypotheses: src (nocs cloud) / dst (observed cloud) (N,3) float64 device arrays, ids (H,4) i32 sampled
 * correspondences (aligning.py:89-93).  Per hypothesis: exact affine through the 4 pairs (cv2.estimateAffine3D on 4
 * points), scale bounds, singular values in [0.8,1.2], nearest rotation, det>0, optional canonical extent check
 * (h_max_dimensions, NULL to skip), inlier count with `threshold` (aligning.py:36-68).
 * counts (H) = inliers or -1 if rejected; transforms (H,16) row-major 4x4 float64.  h_* are HOST pointers (3 doubles). */
int cg_ransac_9d(const double* src, const double* dst, int N, const int* ids, int H, double threshold,
                 const double* h_min_scale, const double* h_max_scale, const double* h_max_dimensions,
                 int* counts, double* transforms, void* stream);
/* mask[p] = |T src_p - dst_p| <= threshold for one transform (device pointer to 16 doubles). */
int cg_similarity_inliers(const double* src, const double* dst, int N, const double* transform16, double threshold,
                          unsigned char* mask, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Grasp affordance P(T|G) (row N3): run_grasp_simulation.py:50-107 compute_grasp_affordance[_worker] with
 * pybullet_env/env_grasp.py:243-283 get_finger_contact_area.  float64 throughout, like the reference.
 * ------------------------------------------------------------------------------------------- */

/* cam_in_finger (G,12): rows [R|t] of inv(finger_mesh_in_grasp) . inv(grasp_in_cam) per grasp (run_grasp_simulation.py:52).
 * pts / normals (P,3): canonical cloud in the camera frame (voxel-downsampled, :97-99); point_affordance (P): affordance of
 * each point's nearest neighbour in the full canonical cloud (the kd-tree lookup of :63-64, grasp independent).
 * h_finger_extents (n_fingers,4) HOST {xmin,xmax,zmin,zmax} of each finger mesh; h_grip_signs (n_fingers) HOST +1 / -1 for
 * grip_dir (0,+-1,0).  Out: p_t_given_g (G) (NaN = the reference drops the grasp), optional contact_counts (G,n_fingers). */
int cg_grasp_affordance(const double* cam_in_finger, long G, const double* pts, const double* normals, const double* point_affordance,
                        int P, int n_fingers, const double* h_finger_extents, const int* h_grip_signs, double surface_tol,
                        double* p_t_given_g, int* contact_counts, void* stream);
/* idx[q] = index of the nearest ref point (float64, first minimum): the cKDTree.query of run_grasp_simulation.py:63. */
int cg_nearest_neighbor(const double* query, long Q, const double* ref, int R, int* idx, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Cone grasp-candidate generation (row N3): PointConeGraspSampler (dexnet/grasping/grasp_sampler.py:155-298), float64.
 * ------------------------------------------------------------------------------------------- */

/* sample_one_surface_point local frame (grasp_sampler.py:225-266) for K sampled points sample_ids (K) of pts/normals (P,3).
 * mode 0: out_doublings[k] = number of `r_ball *= 2` retries point k needs starting from r0 (:243-247);
 * mode 1: frames (K,9) row-major R0 = [approach | major | minor] using radius r_ball[k]. */
int cg_cone_frames(const double* pts, const double* normals, int P, const int* sample_ids, int K, const double* r_ball, double r0,
                   int mode, int* out_doublings, double* frames, void* stream);
/* pose fan-out (grasp_sampler.py:268-289): per point {R0} U {R0.directionVecToRotation(sphere_pt).Rx(q*rot_step_deg)}, column
 * normalised, depths d = q*approach_step: out (K*(1+S*n_rot)*n_depth, 16) float64 row-major 4x4. */
int cg_cone_poses(const double* pts, const int* sample_ids, const double* frames, int K, const double* sphere_pts, int S, int n_rot,
                  double rot_step_deg, int n_depth, double approach_step, double init_bite, double* out, void* stream);
/* center_ob_between_gripper (grasp_sampler.py:189-198): each pose is shifted along its own y axis to the middle of the
 * object's y-extent in the grasp frame (poses must be rigid).  In place. */
int cg_center_grasps(double* poses, long G, const double* pts, int P, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Row N4: PointGroup/lib/pointgroup_ops CUDA ops on the reference's inference path (predicter.py:285-304), forward only.
 * ------------------------------------------------------------------------------------------- */

/* ballquery_batch_p (src/bfs_cluster/bfs_cluster.cu:15-91): neighbours with d^2 < radius^2 inside the point's own batch
 * (batch_offsets), at most 1000 per point, ascending index.  Two passes: pass 0 writes counts (n); the caller scans them
 * into start/len (and applies the n*meanActive cap); pass 1 fills idx[start[p] .. start[p]+len[p]). */
int cg_pg_ballquery_batch_p(const float* xyz, const int* batch_idxs, const int* batch_offsets, int n, float radius, int pass,
                            const int* start, const int* len, int* counts, int* idx, void* stream);
/* sec_mean / sec_min / sec_max (src/sec_mean/sec_mean.cu) and roipool_fp (src/roipool/roipool.cu:12-40):
 * inp (N,C), offsets (n_segments+1); mode 0 mean, 1 min, 2 max, 3 max + argmax (row index, -1 for an empty segment). */
int cg_pg_segment_reduce(const float* inp, const int* offsets, int n_segments, int C, int mode, float* out, int* argmax,
                         void* stream);
/* get_iou (src/get_iou/get_iou.cu:12-37): proposals (CSR idx/offset) x instances -> (nProposal,nInstance) IoU. */
int cg_pg_get_iou(const int* proposals_idx, const int* proposals_offset, const long long* instance_labels,
                  const int* instance_pointnum, int nProposal, int nInstance, float* proposals_iou, void* stream);
/* voxelization forward (src/voxelize/voxelize.cu:10-34): out (n_rows,C, pre-zeroed) += mean/sum of feats rows listed in
 * rules (n_rows, 1+max_active) = [count, idx...]. */
int cg_pg_voxelize_fp(const float* feats, const int* rules, int n_rows, int max_active, int C, int average, float* out,
                      void* stream);
/* point_recover forward (src/voxelize/voxelize.cpp:182-192 -> voxelize.cu:34-48 with average = false; pointgroup_ops.py:77-99):
 * out (n_points,C, pre-zeroed)[rules[m][1+i]] += feats (n_rows,C)[m] for i < rules[m][0].  *err_flag (device int, pre-zeroed) = 1 if a
 * count exceeds max_active or a member index lies outside [0, n_points): such entries are skipped, never written. */
int cg_pg_point_recover(const float* feats, const int* rules, int n_rows, int max_active, int C, int n_points, float* out,
                        int* err_flag, void* stream);

/* voxelization_idx (PointGroup/lib/pointgroup_ops/src/voxelize/voxelize.cpp:11-151; called at predicter.py:285), device pieces
 * around a stable device sort of the packed keys (the host side, catgrasp_amd/pointgroup_ops.py, owns sort/scan/allocation):
 *  cg_pg_voxel_pack_keys : coords (n, ncol) int64, ncol 4 = [batch,x,y,z] or 3 -> keys (n) = batch<<48 | x<<32 | y<<16 | z;
 *                          *err_flag = 1 if a coordinate is outside [0,65536) or a batch index outside [0,32768).
 *  cg_pg_segment_heads   : head[j] = 1 iff sorted_keys[j] starts a run of equal keys.
 *  cg_pg_voxel_fill_maps : for sorted position j (point perm[j], run seg[j] starting at seg_start[seg[j]], voxel vid[seg[j]]):
 *                          input_map[point] = voxel; output_map (M, width) row = [count, member point ids ...] (mode 3 / 4:
 *                          all members ascending; 1: first; 2: last; 0: the only one), pre-zeroed by the caller. */
int cg_pg_voxel_pack_keys(const long long* coords, int n, int ncol, long long* keys, int* err_flag, void* stream);
int cg_pg_segment_heads(const long long* sorted_keys, int n, int* head, void* stream);
int cg_pg_voxel_fill_maps(const long long* perm, const int* seg, const int* seg_start, const int* vid, int n, int width, int mode,
                          int* input_map, int* output_map, void* stream);

/* bfs_cluster (src/bfs_cluster/bfs_cluster.cpp:34-121; called at PointGroup/model/pointgroup/pointgroup.py:240,245): one sweep
 * of min-label propagation with pointer jumping over the CSR neighbour lists (ball_query_idxs, start_len (n,2)) restricted to
 * equal semantic labels; comp (n) starts as 0..n-1; *changed is set when any entry dropped.  Iterate to the fixed point:
 * comp[i] = smallest point index of i's connected component.  n_idx = number of entries of ball_query_idxs: CSR rows are
 * clamped to it and entries outside [0, n) are skipped, so a truncated list cannot cause an out-of-bounds access (the reference's
 * queue BFS, bfs_cluster.cpp:52-64, reads such lists unchecked). */
int cg_pg_cc_propagate(const int* semantic_label, const int* ball_query_idxs, int n_idx, const int* start_len, int n, int* comp,
                       int* changed, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused PointNet++ set abstraction (north_star: "grouped per-neighbourhood MLP reductions"): the consumer of
 * sample_and_group (pointnet2.py:101-129) -- new_points = cat(xyz[idx] - new_xyz, points[idx]) -> [Conv2d(1x1)+BN+ReLU] x L ->
 * max over the K neighbours -- without materialising the grouped tensor.  xyz (B,N,3), points (B,N,D) or NULL (D = 0),
 * new_xyz (B,S,3), idx (B,S,K) int64 (query_ball_point output) -> out (B, cout[L-1], S).  Layers (HOST arrays of length
 * n_layers <= 4): BatchNorm folded, weights packed like cg_gemm_bias_act's (folding.pack_b), layer 0 with its 3 + D input columns
 * zero-padded to cin[0] = 16; cout multiples of 32, <= 256.  *err_flag (device, pre-zeroed) = 1 on an out-of-range index (the
 * reference's index_points raises).  Exact-f32 MFMA. */
int cg_sa_group_mlp_max(const float* xyz, const float* points, const float* new_xyz, const long long* idx, int B, int N, int S, int K,
                        int D, int n_layers, const int* h_cin, const int* h_cout, const float* const* h_w_packed,
                        const float* const* h_bias, float* out, int* err_flag, void* stream);
/* The same with the output addressed by strides, out[b * out_bs + s * out_ss + c * out_cs]: (B,S,C) rows for the next layer to gather
 * from (out_bs = S C, out_ss = C, out_cs = 1), or a channel slice of a multi-scale layer's concatenated output. */
int cg_sa_group_mlp_max_strided(const float* xyz, const float* points, const float* new_xyz, const long long* idx, int B, int N, int S,
                                int K, int D, int n_layers, const int* h_cin, const int* h_cout, const float* const* h_w_packed,
                                const float* const* h_bias, float* out, long out_bs, long out_ss, long out_cs, int* err_flag,
                                void* stream);

/* The same fused layer for the layers PAST the first one of a PointNet++ stack (csrc/sa_tile.hip): any 3 + D with
 * roundup8(3 + D) <= ~590, hidden widths <= 512, last width any multiple of 32, any K, and -- idx == NULL, new_xyz == NULL, S == 1,
 * K == N -- the single group of sample_and_group_all (pointnet2.py:132-149: every point, not centred).  A 64-row tile per
 * workgroup, activations in one LDS strip, weights streamed from L2.  Differences from cg_sa_group_mlp_max in the arguments:
 * layer 0's packed weights take their input columns in the order [D features | xyz | zero pad] with cin[0] = roundup8(3 + D)
 * (the kernel stores the gathered features with aligned 16-byte LDS writes), and the output is addressed by strides --
 * out[b * out_bs + s * out_ss + c * out_cs] -- so that a caller writes (B,C,S) like torch.max(new_points, 2)[0] (out_bs = C S,
 * out_ss = 1, out_cs = S), the (B,S,C) rows the next layer gathers from (out_bs = S C, out_ss = C, out_cs = 1), or one scale's
 * channel slice of a multi-scale layer's concatenated output.  append_xyz = n (3 <= n <= 16, else 0): channels [C, C + n) of every
 * output row additionally receive new_xyz (3 floats) followed by zeros -- the rows [features | xyz | pad] the next level's group-all
 * GEMM chain reads, written by the kernel that produced the features (no concatenation pass). */
int cg_sa_tile_mlp_max(const float* xyz, const float* points, const float* new_xyz, const long long* idx, int B, int N, int S, int K,
                       int D, int n_layers, const int* h_cin, const int* h_cout, const float* const* h_w_packed,
                       const float* const* h_bias, float* out, long out_bs, long out_ss, long out_cs, int append_xyz, int* err_flag,
                       void* stream);

/* Input matrix of the group-all layer when it runs as a GEMM chain (few rows: cg_gemm_bias_act per hidden layer, then
 * cg_gemm_bias_relu_groupmax -- the last layer with the max over the points in its epilogue): rows of
 * [D features | xyz | zero pad to ld], the column order of cg_sa_tile_mlp_max's layer 0.  xyz (rows,3), points (rows,D) or NULL
 * -> out (rows, ld), ld >= D + 3.  (sample_and_group_all's cat([grouped_xyz, points]), pointnet2.py:145-148.) */
int cg_sa_concat_input(const float* xyz, const float* points, long rows, int D, int ld, float* out, void* stream);

/* Last layer of the group-all level with its max over the points in the epilogue: out[g][n] = max over the rows_per_group rows of
 * group g of relu(X[M,K] . W^T + bias) -- sample_and_group_all + Conv2d/BN/ReLU + torch.max(.., 2) (pointnet2.py:132-149) without
 * writing the (M, N) activation.  out (M / rows_per_group, N) is zeroed by the call (stream-ordered), then folded into with an
 * integer atomic max on the non-negative float bits.  That ordering is the float ordering ONLY because the ReLU is always applied
 * (every folded value is >= +0) and out starts at zero; a NaN activation (bits >= 0x7fc00000 as an integer) wins the max, i.e. NaN
 * propagates to the pooled feature like torch.max does (tests/test_pointnet2_encoder_gpu.py::test_group_all_gemm_epilogue_max...).
 * Same kernels, operands and argument rules as cg_gemm_bias_act. */
int cg_gemm_bias_relu_groupmax(const float* x, int M, int K, int ldx, const float* w_packed, int N, const float* bias,
                               int rows_per_group, float* out, void* stream);

/* get_ik_within_limits(...).size() > 0 (my_cpp/common.cpp:9-72, called at :230-236 of filterGraspPose): closed-form IK of the
 * KUKA LBR iiwa14 with the redundancy joint (index 2) at 0 -- what the reference's generated IKFast file solves -- one thread
 * per pose, float64.  ee_in_base (E,16) float32 row-major 4x4 on the device, h_upper7 / h_lower7 HOST joint limits,
 * ok[e] = 1 iff some solution lies inside the limits. */
int cg_iiwa_ik_within_limits(const float* ee_in_base, long E, const double* h_upper7, const double* h_lower7,
                             unsigned char* ok, void* stream);

#ifdef __cplusplus
}
#endif
#endif
