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

// Optional broad phase: a uniform grid in the MESH frame; cell (i,j,k) lists every triangle whose box, inflated by the
// grid's build margin, overlaps the cell (CSR: cell_start, tri_ids).  Built on the host (my_cpp.build_mesh_grid).
struct Grid { float ox, oy, oz, inv_cell; int nx, ny, nz; const int* cell_start; const int* tri_ids; float res_built; };
struct Mesh { const float* V; const int* F; int nf; Grid grid; int has_grid; };
struct Voxels { const short* keys; int nk; };   // (nk,4) int16: key-32768 per axis, 4th unused

// wave-level: does the mesh posed by T (row-major 4x4, wave-uniform) hit any occupied voxel?
// Broad-phase path.  The grid lists are built for voxels of resolution `res_built` and poses whose linear part A
// satisfies sigma_min(A) >= 0.5: a leaf box that touches a posed triangle has its centre within r = res*sqrt(3)/2 of it
// in the camera frame, hence within r/sigma_min <= 2r in the mesh frame, which is the build margin (plus slack for the
// float32 inverse).  The narrow phase is the SAME float32 SAT on the SAME posed vertices as the exhaustive path, so the
// result is identical; poses that do not satisfy the bound take the exhaustive path.
__device__ bool wave_grid_collide(const Mesh& mesh, const float* T, const Voxels& vox, float res, int lane, bool* usable) {
  const Grid& g = mesh.grid;
  const float a00 = T[0], a01 = T[1], a02 = T[2], a10 = T[4], a11 = T[5], a12 = T[6], a20 = T[8], a21 = T[9], a22 = T[10];
  const float c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
  const float det = a00 * c00 + a01 * c01 + a02 * c02;
  *usable = false;
  if (!(fabsf(det) > 1e-12f) || res != g.res_built) return false;
  const float id = 1.0f / det;
  float I[9];
  I[0] = c00 * id; I[1] = (a02 * a21 - a01 * a22) * id; I[2] = (a01 * a12 - a02 * a11) * id;
  I[3] = c01 * id; I[4] = (a00 * a22 - a02 * a20) * id; I[5] = (a02 * a10 - a00 * a12) * id;
  I[6] = c02 * id; I[7] = (a01 * a20 - a00 * a21) * id; I[8] = (a00 * a11 - a01 * a10) * id;
  float fro = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) fro += I[k] * I[k];
  if (!(fro <= 3.96f)) return false;            // 1/||A^-1||_F >= 0.5025  =>  sigma_min(A) >= 0.5 (with margin)
  *usable = true;
  const float h = 0.5f * res;
  const float tx = T[3], ty = T[7], tz = T[11];
  for (int v0 = 0; v0 < vox.nk; v0 += 64) {
    const int v = v0 + lane;
    bool hv = false;
    if (v < vox.nk) {
      const short4 k = ((const short4*)vox.keys)[v];
      float c[3];
      c[0] = ((float)k.x + 0.5f) * res; c[1] = ((float)k.y + 0.5f) * res; c[2] = ((float)k.z + 0.5f) * res;
      const float dx = c[0] - tx, dy = c[1] - ty, dz = c[2] - tz;
      const float qx = I[0] * dx + I[1] * dy + I[2] * dz, qy = I[3] * dx + I[4] * dy + I[5] * dz, qz = I[6] * dx + I[7] * dy + I[8] * dz;
      const float fx = (qx - g.ox) * g.inv_cell, fy = (qy - g.oy) * g.inv_cell, fz = (qz - g.oz) * g.inv_cell;
      if (fx >= 0.f && fy >= 0.f && fz >= 0.f && fx < (float)g.nx && fy < (float)g.ny && fz < (float)g.nz) {
        const int cell = ((int)fx * g.ny + (int)fy) * g.nz + (int)fz;
        const int e0 = g.cell_start[cell], e1 = g.cell_start[cell + 1];
        for (int e = e0; e < e1 && !hv; ++e) {
          const int t = g.tri_ids[e];
          float pv[9];
#pragma unroll
          for (int kk = 0; kk < 3; ++kk) {
            const float* vtx = mesh.V + 3 * (size_t)mesh.F[(size_t)t * 3 + kk];
            const float vx = vtx[0], vy = vtx[1], vz = vtx[2];
#pragma unroll
            for (int r = 0; r < 3; ++r)
              pv[kk * 3 + r] = fmaf(T[r * 4 + 0], vx, fmaf(T[r * 4 + 1], vy, fmaf(T[r * 4 + 2], vz, T[r * 4 + 3])));
          }
          hv = tri_box_overlap(c, h, pv, pv + 3, pv + 6);
        }
      }
    }
    if (__ballot(hv) != 0ull) return true;
  }
  return false;
}

__device__ bool wave_mesh_voxels_collide(const Mesh& mesh, const float* T, const Voxels& vox, float res, float* tl, int lane) {
  if (vox.nk == 0 || mesh.nf == 0) return false;
  if (mesh.has_grid) {
    bool usable;
    const bool r = wave_grid_collide(mesh, T, vox, res, lane, &usable);
    if (usable) return r;
  }
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
  int keep_rejected_pose;                     // poses_out of a rejected evaluation: 0 -> zeros, 1 -> its (un-nudged) grasp_in_cam
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
      a.poses_out[e * 16 + lane] = (code == 0 || a.keep_rejected_pose) ? v : 0.f;
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

inline Mesh make_mesh(const float* V, const int* F, int nf, const cg_mesh_grid* hg) {
  Mesh m; m.V = V; m.F = F; m.nf = nf; m.has_grid = 0; m.grid = Grid{};
  if (hg && hg->cell_start && hg->tri_ids && hg->cell > 0.f) {
    m.grid = Grid{hg->origin[0], hg->origin[1], hg->origin[2], 1.0f / hg->cell, hg->dims[0], hg->dims[1], hg->dims[2], hg->cell_start,
                  hg->tri_ids, hg->resolution};
    m.has_grid = 1;
  }
  return m;
}

inline Mat4 load_mat(const float* h) { Mat4 m; for (int i = 0; i < 16; ++i) m.m[i] = h[i]; return m; }

}  // namespace

extern "C" int cg_filter_grasp_pose_accel(const float* grasp_poses, int n_pose, const float* symmetry_tfs, int n_sym,
                                    const float* h_nocs_pose, const float* h_canonical_to_nocs, const float* h_cam_in_world,
                                    const float* h_ee_in_grasp, const float* h_gripper_in_grasp,
                                    int filter_approach_dir_face_camera, int adjust_collision_pose,
                                    const unsigned char* ik_ok,
                                    const float* gripper_vertices, const int* gripper_faces, int n_gripper_faces,
                                    const float* enclosed_vertices, const int* enclosed_faces, int n_enclosed_faces,
                                    const short* open_keys, int n_open_keys, const short* bg_keys, int n_bg_keys,
                                    float resolution, signed char* codes, float* poses_out, signed char* nudge,
                                    float* ee_in_base_out, const cg_mesh_grid* h_open_grid, const cg_mesh_grid* h_enc_grid,
                                    int keep_rejected_pose, void* stream) {
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
  a.open_mesh = make_mesh(gripper_vertices, gripper_faces, n_gripper_faces, h_open_grid);
  a.enc_mesh = make_mesh(enclosed_vertices, enclosed_faces, n_enclosed_faces, h_enc_grid);
  a.vox_open = Voxels{open_keys, n_open_keys}; a.vox_bg = Voxels{bg_keys, n_bg_keys};
  a.res = resolution; a.codes = codes; a.poses_out = poses_out; a.nudge = nudge; a.ee_out = ee_in_base_out;
  a.keep_rejected_pose = keep_rejected_pose;
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
                     make_mesh(vertices, faces, n_faces, nullptr), poses, n_poses, Voxels{keys, n_keys}, resolution, out);
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
  return cg_filter_grasp_pose_accel(grasp_poses, n_pose, symmetry_tfs, n_sym, h_nocs_pose, h_canonical_to_nocs, h_cam_in_world, h_ee_in_grasp,
                                    h_gripper_in_grasp, filter_approach_dir_face_camera, adjust_collision_pose, ik_ok, gripper_vertices,
                                    gripper_faces, n_gripper_faces, enclosed_vertices, enclosed_faces, n_enclosed_faces, open_keys, n_open_keys,
                                    bg_keys, n_bg_keys, resolution, codes, poses_out, nudge, ee_in_base_out, nullptr, nullptr, 0, stream);
}
