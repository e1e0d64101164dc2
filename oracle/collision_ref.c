/* TEST INFRASTRUCTURE ONLY (oracle) -- CPU restatement of the reference my_cpp collision filter.
 *
 * PARITY UNPINNED: the reference's arithmetic for this path lives in FCL (BVHModel<OBBRSSf> vs
 * fcl::OcTree, `fcl::collide`, my_cpp/collision_manager.cpp:93-111) and octomap
 * (`OcTree::updateNode`, collision_manager.cpp:63-67).  Neither library (nor a pinned version: the
 * reference takes them from an unpinned docker image, my_cpp/CMakeLists.txt:10) is present in
 * /root/reference or installable here, and the reference has no test or golden vector at this
 * boundary.  This file therefore restates
 *   - the CONTROL FLOW of filterGraspPose (my_cpp/common.cpp:156-321) and CollisionManager
 *     (my_cpp/collision_manager.cpp:15-111) line by line, and
 *   - octomap's published key discretisation (OcTreeBaseImpl::coordToKeyChecked: key =
 *     floor(x * (1/res)) + 32768 evaluated in double, depth-16 leaves, a leaf is occupied after one
 *     updateNode(p,true); out-of-range points are ignored) and
 *   - FCL's mesh-vs-octree result as the exact predicate  "some occupied leaf box intersects some
 *     posed mesh triangle"  (surface-only, as BVH-vs-octree is), evaluated with the 13-axis
 *     triangle/box separating-axis test in float32 instead of FCL's GJK.
 * Bit-exactness of the HIP path is defined against THIS predicate.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TREE_MAX_VAL 32768

/* ---------------- octomap key discretisation (collision_manager.cpp:63-67) ---------------- */

/* OcTreeBaseImpl::coordToKeyChecked: scaled = (int)floor(resolution_factor * coordinate) + tree_max_val */
static int coord_to_key(float x, double res_factor, int* key) {
  double s = floor(res_factor * (double)x);
  if (!(s >= -2147483000.0 && s <= 2147483000.0)) return 0;
  long k = (long)s + TREE_MAX_VAL;
  if (k >= 0 && k < 2 * TREE_MAX_VAL) { *key = (int)k; return 1; }
  return 0;
}

static int cmp_u64(const void* a, const void* b) {
  uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return (x > y) - (x < y);
}

/* registerPointCloud: the set of occupied depth-16 leaves.  keys_out: (P,3) int32 capacity (key - 32768),
 * returns the number of unique occupied leaves, sorted by (x,y,z). */
int cr_voxelize(const float* pts, int P, float resolution, int* keys_out) {
  const double res_factor = 1.0 / (double)resolution;   /* OcTree(double resolution) <- float argument */
  uint64_t* packed = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(P > 0 ? P : 1));
  int n = 0;
  for (int i = 0; i < P; ++i) {
    int k[3], ok = 1;
    for (int a = 0; a < 3; ++a) ok &= coord_to_key(pts[i * 3 + a], res_factor, &k[a]);
    if (!ok) continue;                                   /* updateNode returns NULL: point ignored */
    packed[n++] = ((uint64_t)k[0] << 32) | ((uint64_t)k[1] << 16) | (uint64_t)k[2];
  }
  qsort(packed, (size_t)n, sizeof(uint64_t), cmp_u64);
  int m = 0;
  for (int i = 0; i < n; ++i) {
    if (i > 0 && packed[i] == packed[i - 1]) continue;
    keys_out[m * 3 + 0] = (int)(packed[i] >> 32) - TREE_MAX_VAL;
    keys_out[m * 3 + 1] = (int)((packed[i] >> 16) & 0xffff) - TREE_MAX_VAL;
    keys_out[m * 3 + 2] = (int)(packed[i] & 0xffff) - TREE_MAX_VAL;
    ++m;
  }
  free(packed);
  return m;
}

/* ---------------- exact float32 triangle / axis-aligned box overlap (13-axis SAT) ---------------- */

static inline float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
static inline float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

static int plane_box_overlap(const float n[3], const float v[3], float h) {
  float vmin[3], vmax[3];
  for (int q = 0; q < 3; ++q) {
    if (n[q] > 0.0f) { vmin[q] = -h - v[q]; vmax[q] = h - v[q]; }
    else { vmin[q] = h - v[q]; vmax[q] = -h - v[q]; }
  }
  if ((n[0] * vmin[0] + n[1] * vmin[1]) + n[2] * vmin[2] > 0.0f) return 0;
  if ((n[0] * vmax[0] + n[1] * vmax[1]) + n[2] * vmax[2] >= 0.0f) return 1;
  return 0;
}

/* one cross-product axis: the two projections pa,pb of the triangle and the box radius */
#define AXIS(pa, pb, rad) { float mn = fminf(pa, pb), mx = fmaxf(pa, pb); if (mn > (rad) || mx < -(rad)) return 0; }

int cr_tri_box_overlap(const float c[3], float h, const float a[3], const float b[3], const float d[3]) {
  float v0[3], v1[3], v2[3], e0[3], e1[3], e2[3];
  for (int i = 0; i < 3; ++i) { v0[i] = a[i] - c[i]; v1[i] = b[i] - c[i]; v2[i] = d[i] - c[i]; }
  for (int i = 0; i < 3; ++i) { e0[i] = v1[i] - v0[i]; e1[i] = v2[i] - v1[i]; e2[i] = v0[i] - v2[i]; }
  float fex, fey, fez;
  /* edge 0 */
  fex = fabsf(e0[0]); fey = fabsf(e0[1]); fez = fabsf(e0[2]);
  AXIS(e0[2] * v0[1] - e0[1] * v0[2], e0[2] * v2[1] - e0[1] * v2[2], fez * h + fey * h);
  AXIS(-e0[2] * v0[0] + e0[0] * v0[2], -e0[2] * v2[0] + e0[0] * v2[2], fez * h + fex * h);
  AXIS(e0[1] * v1[0] - e0[0] * v1[1], e0[1] * v2[0] - e0[0] * v2[1], fey * h + fex * h);
  /* edge 1 */
  fex = fabsf(e1[0]); fey = fabsf(e1[1]); fez = fabsf(e1[2]);
  AXIS(e1[2] * v0[1] - e1[1] * v0[2], e1[2] * v2[1] - e1[1] * v2[2], fez * h + fey * h);
  AXIS(-e1[2] * v0[0] + e1[0] * v0[2], -e1[2] * v2[0] + e1[0] * v2[2], fez * h + fex * h);
  AXIS(e1[1] * v0[0] - e1[0] * v0[1], e1[1] * v1[0] - e1[0] * v1[1], fey * h + fex * h);
  /* edge 2 */
  fex = fabsf(e2[0]); fey = fabsf(e2[1]); fez = fabsf(e2[2]);
  AXIS(e2[2] * v0[1] - e2[1] * v0[2], e2[2] * v1[1] - e2[1] * v1[2], fez * h + fey * h);
  AXIS(-e2[2] * v0[0] + e2[0] * v0[2], -e2[2] * v1[0] + e2[0] * v1[2], fez * h + fex * h);
  AXIS(e2[1] * v1[0] - e2[0] * v1[1], e2[1] * v2[0] - e2[0] * v2[1], fey * h + fex * h);
  /* box axes */
  for (int i = 0; i < 3; ++i) {
    if (min3f(v0[i], v1[i], v2[i]) > h || max3f(v0[i], v1[i], v2[i]) < -h) return 0;
  }
  /* triangle plane */
  float n[3];
  n[0] = e0[1] * e1[2] - e0[2] * e1[1];
  n[1] = e0[2] * e1[0] - e0[0] * e1[2];
  n[2] = e0[0] * e1[1] - e0[1] * e1[0];
  return plane_box_overlap(n, v0, h);
}

/* ---------------- CollisionManager (collision_manager.cpp:15-111) ---------------- */

/* posed vertex: R v + t with the pose's upper 3x4 (setTransform(pose.block(0,0,3,3), pose.block(0,3,3,1))) */
static void pose_vertex(const float* T, const float* v, float* o) {
  for (int r = 0; r < 3; ++r)
    o[r] = fmaf(T[r * 4 + 0], v[0], fmaf(T[r * 4 + 1], v[1], fmaf(T[r * 4 + 2], v[2], T[r * 4 + 3])));
}

/* isAnyCollision for {mesh at `pose`, octree at identity}.  V:(nv,3) F:(nf,3) keys:(nk,3) */
int cr_mesh_voxels_collide(const float* V, int nv, const int* F, int nf, const float* pose, const int* keys, int nk,
                           float resolution) {
  (void)nv;
  if (nk == 0 || nf == 0) return 0;
  const float h = 0.5f * resolution;
  float* tri = (float*)malloc(sizeof(float) * 9 * (size_t)nf);
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int f = 0; f < nf; ++f)
    for (int k = 0; k < 3; ++k) {
      pose_vertex(pose, V + 3 * F[f * 3 + k], tri + f * 9 + k * 3);
      for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], tri[f * 9 + k * 3 + a]); hi[a] = fmaxf(hi[a], tri[f * 9 + k * 3 + a]); }
    }
  int hit = 0;
  for (int i = 0; i < nk && !hit; ++i) {
    float c[3];
    int out = 0;
    for (int a = 0; a < 3; ++a) {
      c[a] = ((float)keys[i * 3 + a] + 0.5f) * resolution;
      /* conservative early-out only (1e-5 m slack >> float rounding of the SAT); never changes the result */
      if (c[a] - h > hi[a] + 1e-5f || c[a] + h < lo[a] - 1e-5f) out = 1;
    }
    if (out) continue;
    for (int f = 0; f < nf; ++f)
      if (cr_tri_box_overlap(c, h, tri + f * 9, tri + f * 9 + 3, tri + f * 9 + 6)) { hit = 1; break; }
  }
  free(tri);
  return hit;
}

/* ---------------- filterGraspPose (common.cpp:156-321) ---------------- */

/* Eigen Matrix4f product without FMA: ((a0 b0 + a1 b1) + a2 b2) + a3 b3, row-major storage here */
static void mat4_mul(const float* A, const float* B, float* C) {
  float t[16];
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c)
      t[r * 4 + c] = ((A[r * 4 + 0] * B[0 * 4 + c] + A[r * 4 + 1] * B[1 * 4 + c]) + A[r * 4 + 2] * B[2 * 4 + c]) + A[r * 4 + 3] * B[3 * 4 + c];
  memcpy(C, t, sizeof(t));
}

/* VectorBlock::normalize(): z = squaredNorm(); if (z > 0) v /= sqrt(z) */
static void normalize_col(float* M, int col) {
  float x = M[0 * 4 + col], y = M[1 * 4 + col], z = M[2 * 4 + col];
  float s = (x * x + y * y) + z * z;
  if (s > 0.0f) { float n = sqrtf(s); M[0 * 4 + col] = x / n; M[1 * 4 + col] = y / n; M[2 * 4 + col] = z / n; }
}

typedef int (*cr_ik_fn)(const float* ee_in_base16, void* user);   /* 1 = at least one IK solution within limits */

/* Evaluate every (pose i, symmetry j) pair in input order.
 * codes[i*n_sym+j]: 0 keep, 1 approach-dir reject (:199-212), 2 IK reject (:214-226), 3 open-gripper
 * collision (:231-239, or "not found" in adjust mode :290-294), 4 enclosed-gripper collision (:241-249).
 * poses_out: (n_pose*n_sym,16) surviving grasp_in_cam (possibly nudged), zero otherwise.
 * nudge: index of the accepted nudge {0:+0, 1:+1mm, 2:-1mm, 3:+2mm, 4:-2mm} or -1. */
void cr_filter_grasp_pose(const float* grasp_poses, int n_pose, const float* symmetry_tfs, int n_sym,
                          const float* nocs_pose, const float* canonical_to_nocs, const float* cam_in_world,
                          const float* ee_in_grasp, const float* gripper_in_grasp, int filter_approach_dir_face_camera,
                          int filter_ik, int adjust_collision_pose, cr_ik_fn ik_fn, void* ik_user,
                          const float* gV, int gnv, const int* gF, int gnf, const float* eV, int env_, const int* eF, int enf,
                          const int* keys_open, int nk_open, const int* keys_bg, int nk_bg, float resolution,
                          signed char* codes, float* poses_out, signed char* nudge) {
  float c2c[16];
  mat4_mul(nocs_pose, canonical_to_nocs, c2c);                       /* :159 */
#pragma omp parallel for schedule(dynamic)
  for (int i = 0; i < n_pose; ++i) {
    for (int j = 0; j < n_sym; ++j) {
      const int e = i * n_sym + j;
      float tmp[16], gic[16];
      codes[e] = 0; nudge[e] = -1;
      memset(poses_out + (size_t)e * 16, 0, 16 * sizeof(float));
      mat4_mul(symmetry_tfs + j * 16, grasp_poses + (size_t)i * 16, tmp);     /* :191 */
      mat4_mul(c2c, tmp, gic);                                       /* :192 */
      for (int col = 0; col < 3; ++col) normalize_col(gic, col);     /* :194-197 */
      if (filter_approach_dir_face_camera) {
        /* :201-204: dot(normalized col0, (0,0,1)) < 0  <=>  z component < 0 */
        if (gic[2 * 4 + 0] < 0.0f) { codes[e] = 1; continue; }
      }
      if (filter_ik) {
        float t2[16], ee[16];
        mat4_mul(cam_in_world, gic, t2);
        mat4_mul(t2, ee_in_grasp, ee);                               /* :216 */
        if (!ik_fn || !ik_fn(ee, ik_user)) { codes[e] = 2; continue; }
      }
      if (!adjust_collision_pose) {
        float gripper_in_cam[16];
        mat4_mul(gic, gripper_in_grasp, gripper_in_cam);             /* :230 */
        if (cr_mesh_voxels_collide(gV, gnv, gF, gnf, gripper_in_cam, keys_open, nk_open, resolution)) { codes[e] = 3; continue; }
        if (cr_mesh_voxels_collide(eV, env_, eF, enf, gripper_in_cam, keys_bg, nk_bg, resolution)) { codes[e] = 4; continue; }
        nudge[e] = 0;
      } else {
        const float major[3] = {gic[0 * 4 + 1], gic[1 * 4 + 1], gic[2 * 4 + 1]};   /* :253 */
        int found = 0, idx = 0;
        /* :255-288.  float step: 0, 0.001f, 0.002f; 0.002f+0.001f > 0.003 (double) ends the loop */
        for (float step = 0.0f; (double)step <= 0.003 && !found; step += 0.001f) {
          const int nsign = (step == 0.0f) ? 1 : 2;
          for (int s = 0; s < nsign; ++s, ++idx) {
            const int sign = (s == 0) ? 1 : -1;
            float cur[16], gripper_in_cam[16];
            memcpy(cur, gic, sizeof(cur));
            for (int a = 0; a < 3; ++a) cur[a * 4 + 3] = cur[a * 4 + 3] + (step * major[a]) * (float)sign;   /* :265 */
            mat4_mul(cur, gripper_in_grasp, gripper_in_cam);
            if (cr_mesh_voxels_collide(gV, gnv, gF, gnf, gripper_in_cam, keys_open, nk_open, resolution)) continue;
            if (cr_mesh_voxels_collide(eV, env_, eF, enf, gripper_in_cam, keys_bg, nk_bg, resolution)) continue;
            memcpy(gic, cur, sizeof(cur));
            found = 1; nudge[e] = (signed char)idx;
            break;
          }
        }
        if (!found) { codes[e] = 3; continue; }                       /* :290-294 */
      }
      memcpy(poses_out + (size_t)e * 16, gic, 16 * sizeof(float));    /* :296-299 */
    }
  }
}

int cr_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
