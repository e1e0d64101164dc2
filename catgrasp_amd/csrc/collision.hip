// Per-candidate gripper-vs-scene collision filter: the device implementation of my_cpp.filterGraspPose
// (my_cpp/common.cpp:156-321) and CollisionManager.isAnyCollision for {posed triangle mesh, voxelised
// point cloud} pairs (my_cpp/collision_manager.cpp:15-111).
//
// Semantics (see DESIGN.md "collision predicate"): a point cloud registered at resolution `res` is the set
// of occupied octomap depth-16 leaves (key = floor(x/res) + 32768 in double); a posed mesh collides with
// it iff some occupied leaf box intersects some posed triangle (13-axis SAT in float32).  One wavefront
// evaluates one (grasp pose, symmetry transform) pair: its 64 lanes stride over the occupied voxels
// (8-byte int16x4 keys, coalesced), the posed triangles of the gripper live in LDS, and a wave ballot
// gives the early exit.  HBM traffic is 64 B of pose in and 66 B out per evaluation; the voxel and
// mesh arrays are L2 resident.
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"

namespace {

constexpr int TRI_CHUNK = 128;             // posed triangles staged per wave in LDS
constexpr int TRI_FLOATS = 15;             // 9 vertex floats + 6 AABB floats
constexpr int WAVES = 4;

struct Mat4 { float m[16]; };

__device__ __forceinline__ void mat4_mul(const float* A, const float* B, float* C) {
  float t[16];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c)
      t[r * 4 + c] = ((A[r * 4 + 0] * B[0 * 4 + c] + A[r * 4 + 1] * B[1 * 4 + c]) + A[r * 4 + 2] * B[2 * 4 + c]) + A[r * 4 + 3] * B[3 * 4 + c];
#pragma unroll
  for (int i = 0; i < 16; ++i) C[i] = t[i];
}

__device__ __forceinline__ void normalize_col(float* M, int col) {
  const float x = M[0 * 4 + col], y = M[1 * 4 + col], z = M[2 * 4 + col];
  const float s = (x * x + y * y) + z * z;
  if (s > 0.0f) { const float n = sqrtf(s); M[0 * 4 + col] = x / n; M[1 * 4 + col] = y / n; M[2 * 4 + col] = z / n; }
}

__device__ __forceinline__ bool plane_box_overlap(const float* n, const float* v, float h) {
  float vmin[3], vmax[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    if (n[q] > 0.0f) { vmin[q] = -h - v[q]; vmax[q] = h - v[q]; }
    else { vmin[q] = h - v[q]; vmax[q] = -h - v[q]; }
  }
  if ((n[0] * vmin[0] + n[1] * vmin[1]) + n[2] * vmin[2] > 0.0f) return false;
  if ((n[0] * vmax[0] + n[1] * vmax[1]) + n[2] * vmax[2] >= 0.0f) return true;
  return false;
}

#define CG_AXIS(pa, pb, rad) { const float _a = (pa), _b = (pb), _r = (rad); \
    const float mn = fminf(_a, _b), mx = fmaxf(_a, _b); if (mn > _r || mx < -_r) return false; }

// exact float32 triangle / axis-aligned cube overlap (separating axes: 9 edge crosses, 3 box axes, plane)
__device__ __forceinline__ bool tri_box_overlap(const float* c, float h, const float* a, const float* b, const float* d) {
  float v0[3], v1[3], v2[3], e0[3], e1[3], e2[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { v0[i] = a[i] - c[i]; v1[i] = b[i] - c[i]; v2[i] = d[i] - c[i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) { e0[i] = v1[i] - v0[i]; e1[i] = v2[i] - v1[i]; e2[i] = v0[i] - v2[i]; }
  float fex, fey, fez;
  fex = fabsf(e0[0]); fey = fabsf(e0[1]); fez = fabsf(e0[2]);
  CG_AXIS(e0[2] * v0[1] - e0[1] * v0[2], e0[2] * v2[1] - e0[1] * v2[2], fez * h + fey * h);
  CG_AXIS(-e0[2] * v0[0] + e0[0] * v0[2], -e0[2] * v2[0] + e0[0] * v2[2], fez * h + fex * h);
  CG_AXIS(e0[1] * v1[0] - e0[0] * v1[1], e0[1] * v2[0] - e0[0] * v2[1], fey * h + fex * h);
  fex = fabsf(e1[0]); fey = fabsf(e1[1]); fez = fabsf(e1[2]);
  CG_AXIS(e1[2] * v0[1] - e1[1] * v0[2], e1[2] * v2[1] - e1[1] * v2[2], fez * h + fey * h);
  CG_AXIS(-e1[2] * v0[0] + e1[0] * v0[2], -e1[2] * v2[0] + e1[0] * v2[2], fez * h + fex * h);
  CG_AXIS(e1[1] * v0[0] - e1[0] * v0[1], e1[1] * v1[0] - e1[0] * v1[1], fey * h + fex * h);
  fex = fabsf(e2[0]); fey = fabsf(e2[1]); fez = fabsf(e2[2]);
  CG_AXIS(e2[2] * v0[1] - e2[1] * v0[2], e2[2] * v1[1] - e2[1] * v1[2], fez * h + fey * h);
  CG_AXIS(-e2[2] * v0[0] + e2[0] * v0[2], -e2[2] * v1[0] + e2[0] * v1[2], fez * h + fex * h);
  CG_AXIS(e2[1] * v1[0] - e2[0] * v1[1], e2[1] * v2[0] - e2[0] * v2[1], fey * h + fex * h);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (fminf(fminf(v0[i], v1[i]), v2[i]) > h || fmaxf(fmaxf(v0[i], v1[i]), v2[i]) < -h) return false;
  }
  float n[3];
  n[0] = e0[1] * e1[2] - e0[2] * e1[1];
  n[1] = e0[2] * e1[0] - e0[0] * e1[2];
  n[2] = e0[0] * e1[1] - e0[1] * e1[0];
  return plane_box_overlap(n, v0, h);
}

struct Mesh { const float* V; const int* F; int nf; };
struct Voxels { const short* keys; int nk; };   // (nk,4) int16: key-32768 per axis, 4th unused

// wave-level: does the mesh posed by T (row-major 4x4, wave-uniform) hit any occupied voxel?
__device__ bool wave_mesh_voxels_collide(const Mesh& mesh, const float* T, const Voxels& vox, float res, float* tl, int lane) {
  if (vox.nk == 0 || mesh.nf == 0) return false;
  const float h = 0.5f * res;
  const float slack = 1e-5f;     // conservative culls only; never changes the predicate
  bool hit = false;
  for (int c0 = 0; c0 < mesh.nf && !hit; c0 += TRI_CHUNK) {
    const int nt = min(TRI_CHUNK, mesh.nf - c0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int t = lane; t < nt; t += 64) {
      float tlo[3] = {INFINITY, INFINITY, INFINITY}, thi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float* v = mesh.V + 3 * (size_t)mesh.F[(size_t)(c0 + t) * 3 + k];
        const float vx = v[0], vy = v[1], vz = v[2];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float o = fmaf(T[r * 4 + 0], vx, fmaf(T[r * 4 + 1], vy, fmaf(T[r * 4 + 2], vz, T[r * 4 + 3])));
          tl[t * TRI_FLOATS + k * 3 + r] = o;
          tlo[r] = fminf(tlo[r], o); thi[r] = fmaxf(thi[r], o);
        }
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        tl[t * TRI_FLOATS + 9 + r] = tlo[r] - slack; tl[t * TRI_FLOATS + 12 + r] = thi[r] + slack;
        lo[r] = fminf(lo[r], tlo[r]); hi[r] = fmaxf(hi[r], thi[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      for (int o = 32; o > 0; o >>= 1) { lo[r] = fminf(lo[r], __shfl_xor(lo[r], o)); hi[r] = fmaxf(hi[r], __shfl_xor(hi[r], o)); }
      lo[r] -= slack; hi[r] += slack;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int v0 = 0; v0 < vox.nk; v0 += 64) {
      const int v = v0 + lane;
      bool hv = false;
      if (v < vox.nk) {
        const short4 k = ((const short4*)vox.keys)[v];
        float c[3];
        c[0] = ((float)k.x + 0.5f) * res; c[1] = ((float)k.y + 0.5f) * res; c[2] = ((float)k.z + 0.5f) * res;
        const bool out = (c[0] - h > hi[0]) || (c[0] + h < lo[0]) || (c[1] - h > hi[1]) || (c[1] + h < lo[1]) ||
                         (c[2] - h > hi[2]) || (c[2] + h < lo[2]);
        if (!out) {
          for (int t = 0; t < nt && !hv; ++t) {
            const float* q = tl + t * TRI_FLOATS;
            if ((c[0] - h > q[12]) || (c[0] + h < q[9]) || (c[1] - h > q[13]) || (c[1] + h < q[10]) ||
                (c[2] - h > q[14]) || (c[2] + h < q[11])) continue;
            hv = tri_box_overlap(c, h, q, q + 3, q + 6);
          }
        }
      }
      if (__ballot(hv) != 0ull) { hit = true; break; }
    }
  }
  return hit;
}

struct FilterArgs {
  const float* grasp_poses; int n_pose;       // (n_pose,16)
  const float* symmetry_tfs; int n_sym;       // (n_sym,16)
  Mat4 nocs_pose, canonical_to_nocs, cam_in_world, ee_in_grasp, gripper_in_grasp;
  int filter_dir, adjust;
  const unsigned char* ik_ok;                 // optional (E): 0 -> IK reject
  Mesh open_mesh, enc_mesh;
  Voxels vox_open, vox_bg;
  float res;
  signed char* codes; float* poses_out; signed char* nudge;
  float* ee_out;                              // optional (E,16): ee_in_base for the host IK pass; stops after the dir test
};

__global__ __launch_bounds__(64 * WAVES) void filter_grasp_pose_kernel(FilterArgs a) {
  __shared__ float tl_all[WAVES][TRI_CHUNK * TRI_FLOATS];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  float* tl = tl_all[wv];
  const long E = (long)a.n_pose * a.n_sym;
  float c2c[16];
  mat4_mul(a.nocs_pose.m, a.canonical_to_nocs.m, c2c);
  for (long e = (long)blockIdx.x * WAVES + wv; e < E; e += (long)gridDim.x * WAVES) {
    const int i = (int)(e / a.n_sym), j = (int)(e - (long)i * a.n_sym);
    float P[16], S[16], tmp[16], gic[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) { P[k] = a.grasp_poses[(size_t)i * 16 + k]; S[k] = a.symmetry_tfs[(size_t)j * 16 + k]; }
    mat4_mul(S, P, tmp);
    mat4_mul(c2c, tmp, gic);
    normalize_col(gic, 0); normalize_col(gic, 1); normalize_col(gic, 2);
    int code = 0, nud = -1;
    if (a.filter_dir && gic[2 * 4 + 0] < 0.0f) code = 1;
    if (a.ee_out) {
      if (lane < 16) {
        float t2[16], ee[16];
        mat4_mul(a.cam_in_world.m, gic, t2);
        mat4_mul(t2, a.ee_in_grasp.m, ee);
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) if (lane == k) v = ee[k];
        a.ee_out[e * 16 + lane] = v;
      }
      if (lane == 0) a.codes[e] = (signed char)code;
      continue;
    }
    if (code == 0 && a.ik_ok && a.ik_ok[e] == 0) code = 2;
    if (code == 0) {
      if (!a.adjust) {
        float gcam[16];
        mat4_mul(gic, a.gripper_in_grasp.m, gcam);
        if (wave_mesh_voxels_collide(a.open_mesh, gcam, a.vox_open, a.res, tl, lane)) code = 3;
        else if (wave_mesh_voxels_collide(a.enc_mesh, gcam, a.vox_bg, a.res, tl, lane)) code = 4;
        else nud = 0;
      } else {
        const float major[3] = {gic[1], gic[5], gic[9]};
        bool found = false;
        int idx = 0;
        for (float step = 0.0f; (double)step <= 0.003 && !found; step += 0.001f) {
          const int nsign = (step == 0.0f) ? 1 : 2;
          for (int s = 0; s < nsign; ++s, ++idx) {
            const float sign = (s == 0) ? 1.0f : -1.0f;
            float cur[16], gcam[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) cur[k] = gic[k];
#pragma unroll
            for (int r = 0; r < 3; ++r) cur[r * 4 + 3] = cur[r * 4 + 3] + (step * major[r]) * sign;
            mat4_mul(cur, a.gripper_in_grasp.m, gcam);
            if (wave_mesh_voxels_collide(a.open_mesh, gcam, a.vox_open, a.res, tl, lane)) continue;
            if (wave_mesh_voxels_collide(a.enc_mesh, gcam, a.vox_bg, a.res, tl, lane)) continue;
#pragma unroll
            for (int k = 0; k < 16; ++k) gic[k] = cur[k];
            found = true; nud = idx;
            break;
          }
        }
        if (!found) code = 3;
      }
    }
    if (lane < 16) {
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) if (lane == k) v = gic[k];
      a.poses_out[e * 16 + lane] = (code == 0) ? v : 0.f;
    }
    if (lane == 0) { a.codes[e] = (signed char)code; a.nudge[e] = (signed char)nud; }
  }
}

// CollisionManager.isAnyCollision for E independent mesh poses against one voxel set
__global__ __launch_bounds__(64 * WAVES) void mesh_voxels_collide_kernel(Mesh mesh, const float* poses, long E, Voxels vox,
                                                                         float res, unsigned char* out) {
  __shared__ float tl_all[WAVES][TRI_CHUNK * TRI_FLOATS];
  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  for (long e = (long)blockIdx.x * WAVES + wv; e < E; e += (long)gridDim.x * WAVES) {
    float T[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) T[k] = poses[e * 16 + k];
    const bool hit = wave_mesh_voxels_collide(mesh, T, vox, res, tl_all[wv], lane);
    if (lane == 0) out[e] = hit ? 1 : 0;
  }
}

// octomap coordToKeyChecked for every point: packed = (kx<<32 | ky<<16 | kz) with keys in [0,65536), or -1
__global__ void voxel_keys_kernel(const float* __restrict__ pts, long P, double res_factor, long long* __restrict__ packed) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  long long pk = 0; bool ok = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const double s = floor(res_factor * (double)pts[i * 3 + a]);
    if (!(s >= -2147483000.0 && s <= 2147483000.0)) { ok = false; continue; }
    const long long k = (long long)s + 32768;
    if (k < 0 || k >= 65536) ok = false;
    pk = (pk << 16) | (k & 0xffff);
  }
  packed[i] = ok ? pk : -1ll;
}

__global__ void unpack_keys_kernel(const long long* __restrict__ packed, long n, short* __restrict__ keys4) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long pk = packed[i];
  short4 k;
  k.x = (short)((int)((pk >> 32) & 0xffff) - 32768);
  k.y = (short)((int)((pk >> 16) & 0xffff) - 32768);
  k.z = (short)((int)(pk & 0xffff) - 32768);
  k.w = 0;
  ((short4*)keys4)[i] = k;
}

inline Mat4 load_mat(const float* h) { Mat4 m; for (int i = 0; i < 16; ++i) m.m[i] = h[i]; return m; }

}  // namespace

extern "C" int cg_filter_grasp_pose(const float* grasp_poses, int n_pose, const float* symmetry_tfs, int n_sym,
                                    const float* h_nocs_pose, const float* h_canonical_to_nocs, const float* h_cam_in_world,
                                    const float* h_ee_in_grasp, const float* h_gripper_in_grasp,
                                    int filter_approach_dir_face_camera, int adjust_collision_pose,
                                    const unsigned char* ik_ok,
                                    const float* gripper_vertices, const int* gripper_faces, int n_gripper_faces,
                                    const float* enclosed_vertices, const int* enclosed_faces, int n_enclosed_faces,
                                    const short* open_keys, int n_open_keys, const short* bg_keys, int n_bg_keys,
                                    float resolution, signed char* codes, float* poses_out, signed char* nudge,
                                    float* ee_in_base_out, void* stream) {
  if (n_pose < 0 || n_sym < 0) return CG_ERR_ARG;
  if ((long)n_pose * n_sym == 0) return CG_OK;
  if (!grasp_poses || !symmetry_tfs || !h_nocs_pose || !h_canonical_to_nocs || !h_cam_in_world || !h_ee_in_grasp ||
      !h_gripper_in_grasp || !codes)
    return CG_ERR_ARG;
  if ( n_gripper_faces < 0 || n_enclosed_faces < 0 || n_open_keys < 0 || n_bg_keys < 0) return CG_ERR_ARG;
  if (!ee_in_base_out && (!poses_out || !nudge)) return CG_ERR_ARG;
  if ((n_gripper_faces > 0 && (!gripper_vertices || !gripper_faces)) || (n_enclosed_faces > 0 && (!enclosed_vertices || !enclosed_faces)))
    return CG_ERR_ARG;
  if ((n_open_keys > 0 && !open_keys) || (n_bg_keys > 0 && !bg_keys)) return CG_ERR_ARG;
  if (!(resolution > 0.f)) return CG_ERR_ARG;
  const long E = (long)n_pose * n_sym;
  if (E == 0) return CG_OK;
  FilterArgs a;
  a.grasp_poses = grasp_poses; a.n_pose = n_pose; a.symmetry_tfs = symmetry_tfs; a.n_sym = n_sym;
  a.nocs_pose = load_mat(h_nocs_pose); a.canonical_to_nocs = load_mat(h_canonical_to_nocs);
  a.cam_in_world = load_mat(h_cam_in_world); a.ee_in_grasp = load_mat(h_ee_in_grasp);
  a.gripper_in_grasp = load_mat(h_gripper_in_grasp);
  a.filter_dir = filter_approach_dir_face_camera; a.adjust = adjust_collision_pose; a.ik_ok = ik_ok;
  a.open_mesh = Mesh{gripper_vertices, gripper_faces, n_gripper_faces};
  a.enc_mesh = Mesh{enclosed_vertices, enclosed_faces, n_enclosed_faces};
  a.vox_open = Voxels{open_keys, n_open_keys}; a.vox_bg = Voxels{bg_keys, n_bg_keys};
  a.res = resolution; a.codes = codes; a.poses_out = poses_out; a.nudge = nudge; a.ee_out = ee_in_base_out;
  long blocks = (E + WAVES - 1) / WAVES;
  if (blocks > 256 * 16) blocks = 256 * 16;      // grid-stride: 16 blocks per CU
  hipLaunchKernelGGL(filter_grasp_pose_kernel, dim3((unsigned)blocks), dim3(64 * WAVES), 0, (hipStream_t)stream, a);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_mesh_voxels_collide(const float* vertices, const int* faces, int n_faces, const float* poses, long n_poses,
                                      const short* keys, int n_keys, float resolution, unsigned char* out, void* stream) {
  if (n_faces < 0 || n_keys < 0 || n_poses < 0 || !(resolution > 0.f)) return CG_ERR_ARG;
  if (n_poses == 0) return CG_OK;
  if (!poses || !out) return CG_ERR_ARG;
  if ((n_faces > 0 && (!vertices || !faces)) || (n_keys > 0 && !keys)) return CG_ERR_ARG;
  long blocks = (n_poses + WAVES - 1) / WAVES;
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(mesh_voxels_collide_kernel, dim3((unsigned)blocks), dim3(64 * WAVES), 0, (hipStream_t)stream,
                     Mesh{vertices, faces, n_faces}, poses, n_poses, Voxels{keys, n_keys}, resolution, out);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_voxel_keys(const float* pts, long n_pts, float resolution, long long* packed, void* stream) {
  if (n_pts < 0 || !(resolution > 0.f)) return CG_ERR_ARG;
  if (n_pts == 0) return CG_OK;
  if (!pts || !packed) return CG_ERR_ARG;
  const double res_factor = 1.0 / (double)resolution;
  hipLaunchKernelGGL(voxel_keys_kernel, dim3((unsigned)((n_pts + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     pts, n_pts, res_factor, packed);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_unpack_voxel_keys(const long long* packed, long n, short* keys4, void* stream) {
  if (n < 0) return CG_ERR_ARG;
  if (n == 0) return CG_OK;
  if (!packed || !keys4) return CG_ERR_ARG;
  hipLaunchKernelGGL(unpack_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, packed, n, keys4);
  return cg_hip_status(hipGetLastError());
}
