// Per-candidate gripper-vs-scene collision filter: the device implementation of my_cpp.filterGraspPose
// (my_cpp/common.cpp:156-321) and CollisionManager.isAnyCollision for {posed triangle mesh, voxelised
// point cloud} pairs (my_cpp/collision_manager.cpp:15-111).
//
// Semantics (see DESIGN.md "collision predicate"): a point cloud registered at resolution `res` is the set
// of occupied octomap depth-16 leaves (key = floor(x/res) + 32768 in double); a posed mesh collides with
// it iff some occupied leaf box intersects some posed triangle (13-axis SAT in float32).  One wavefront
// evaluates one (grasp pose, symmetry transform) pair: its 64 lanes stride over the occupied voxels
// (8-byte int16x4 keys, coalesced), queue the (voxel, triangle) pairs the mesh-frame grid hands them in LDS and
// test the queue 64 pairs at a time; a wave ballot gives the early exit.  HBM traffic is 64 B of pose in and 66 B out per evaluation; the voxel and
// mesh arrays are L2 resident.
#include <stddef.h>
#include <stdlib.h>
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

// A value every lane of the wave holds identically, moved to a scalar register.  The 4x4 matrices of an evaluation are wave-uniform
// but computed by the vector ALU (gfx950 has no scalar float unit): left in VGPRs they cost ~100 registers per lane across the
// collision loops; as SGPRs they are free operands of the lanes' v_fma.
__device__ __forceinline__ float uni(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }

__device__ __forceinline__ void normalize_col(float* M, int col) {
  const float x = M[0 * 4 + col], y = M[1 * 4 + col], z = M[2 * 4 + col];
  const float s = (x * x + y * y) + z * z;
  if (s > 0.0f) { const float n = sqrtf(s); M[0 * 4 + col] = x / n; M[1 * 4 + col] = y / n; M[2 * 4 + col] = z / n; }
}

// Exact float32 triangle / axis-aligned cube overlap (separating axes: 9 edge crosses, 3 box axes, plane), BRANCH-FREE: every axis
// is evaluated and the "separated" bits are OR-ed.  In a wavefront the lanes test different (voxel, triangle) pairs, so an early
// exit per axis saves nothing unless all 64 lanes take it, while each `return` costs an exec-mask save in SGPRs (the version with
// 13 early exits spilled 364 of them).  Every expression is the one the early-exit version evaluated (same operation order, no
// FMA contraction), so the predicate is bit-identical to it and to oracle/collision_ref.c.
#define CG_AXIS(pa, pb, rad) { const float _a = (pa), _b = (pb), _r = (rad); \
    sep = fminf(_a, _b) > _r ? 1 : sep; sep = fmaxf(_a, _b) < -_r ? 1 : sep; }

__device__ __forceinline__ bool tri_box_overlap(const float* c, float h, const float* a, const float* b, const float* d) {
  float v0[3], v1[3], v2[3], e0[3], e1[3], e2[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) { v0[i] = a[i] - c[i]; v1[i] = b[i] - c[i]; v2[i] = d[i] - c[i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) { e0[i] = v1[i] - v0[i]; e1[i] = v2[i] - v1[i]; e2[i] = v0[i] - v2[i]; }
  int sep = 0;
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
  for (int i = 0; i < 3; ++i)
  { sep = fminf(fminf(v0[i], v1[i]), v2[i]) > h ? 1 : sep; sep = fmaxf(fmaxf(v0[i], v1[i]), v2[i]) < -h ? 1 : sep; }
  float n[3];
  n[0] = e0[1] * e1[2] - e0[2] * e1[1];
  n[1] = e0[2] * e1[0] - e0[0] * e1[2];
  n[2] = e0[0] * e1[1] - e0[1] * e1[0];
  // plane / box: the box vertex extremal along n, on either side (vmin: most negative, vmax: most positive)
  float vmin[3], vmax[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const bool pos = n[q] > 0.0f;
    vmin[q] = (pos ? -h : h) - v0[q];
    vmax[q] = (pos ? h : -h) - v0[q];
  }
  const float dmin = (n[0] * vmin[0] + n[1] * vmin[1]) + n[2] * vmin[2];
  const float dmax = (n[0] * vmax[0] + n[1] * vmax[1]) + n[2] * vmax[2];
  return !sep && !(dmin > 0.0f) && (dmax >= 0.0f);
}

// Optional broad phase: a uniform grid in the MESH frame; cell (i,j,k) lists every triangle whose box, inflated by the
// grid's build margin, overlaps the cell (CSR: cell_start, tri_ids).  Built on the device (cg_mesh_grid_count / _fill).
// tri_verts (optional): the triangles as a flat (nf,12) float array [v0 v1 v2 pad] -- one indirection less than F -> V.
struct Grid { float ox, oy, oz, inv_cell; int nx, ny, nz; const int* cell_start; const int* tri_ids; const float* tri_verts; float res_built;
              const unsigned char* coarse; };    // coarse (optional): 1 per 4 x 4 x 4 block of cells that holds a non-empty cell
struct Mesh { const float* V; const int* F; int nf; Grid grid; int has_grid; };
struct Voxels { const short* keys; int nk; const short* blocks; };   // keys (nk,4) int16: key-32768 per axis, 4th unused; blocks (optional):
                                                                     // (ceil(nk/64), 2, 4) int16 = lowest / highest key of every run of 64 keys

// How the collision loops see a voxel set.  Single call: the kernel argument itself (its loads are rematerialisable, so they cost no
// register between uses).  MULTI: the segment's sets live in the wave's LDS row and every use reads them back from there -- loaded
// from the device table into registers they would have to stay alive across the loops, which have no scalar register to spare
// (36 spilled in the first form of the MULTI kernel).
struct SegLds { const short* keys[2]; const short* blocks[2]; int nk[2]; int adjust; int sg; };
template <bool MULTI> struct VoxView;
template <> struct VoxView<false> {
  const Voxels& v;
  __device__ __forceinline__ const short* keys() const { return v.keys; }
  __device__ __forceinline__ const short* blocks() const { return v.blocks; }
  __device__ __forceinline__ int nk() const { return v.nk; }
};
template <> struct VoxView<true> {
  const SegLds* sl; int m;
  __device__ __forceinline__ const short* keys() const { return sl->keys[m]; }
  __device__ __forceinline__ const short* blocks() const { return sl->blocks[m]; }
  __device__ __forceinline__ int nk() const { return __builtin_amdgcn_readfirstlane(sl->nk[m]); }
};

constexpr int PAIR_CAP = 512;              // (voxel, triangle) pairs a wave queues in LDS before it tests them
constexpr int PAIR_DRAIN = 256;            // ... and the fill from which it does (a test round occupies all 64 lanes but the last)
constexpr int BMASK_WORDS = 64;           // voxel blocks whose relevance is decided up front: 64 x 64 blocks = 262,144 voxels per set
struct alignas(16) PairList { short4 key[PAIR_CAP]; int tri[PAIR_CAP]; float T[12]; float Ab[12]; unsigned long long bmask[BMASK_WORDS]; };   // Ab: the key -> cell map   // + the posed-gripper matrix (rows 0..2) of the test in flight

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// The grid lists are built for voxels of resolution `res_built` and poses whose linear part A satisfies sigma_min(A) >= 0.5: a leaf
// box that touches a posed triangle has its centre within r = res*sqrt(3)/2 of it in the camera frame, hence within
// r/sigma_min <= 2r in the mesh frame, which is the build margin (plus slack for the float32 inverse).  -> usable, and I = A^-1.
__device__ __forceinline__ bool grid_usable(const Grid& g, const float* T, float res, float* I) {
  const float a00 = T[0], a01 = T[1], a02 = T[2], a10 = T[4], a11 = T[5], a12 = T[6], a20 = T[8], a21 = T[9], a22 = T[10];
  const float c00 = a11 * a22 - a12 * a21, c01 = a12 * a20 - a10 * a22, c02 = a10 * a21 - a11 * a20;
  const float det = a00 * c00 + a01 * c01 + a02 * c02;
  if (!(fabsf(det) > 1e-12f) || res != g.res_built) return false;
  const float id = 1.0f / det;
  I[0] = c00 * id; I[1] = (a02 * a21 - a01 * a22) * id; I[2] = (a01 * a12 - a02 * a11) * id;
  I[3] = c01 * id; I[4] = (a00 * a22 - a02 * a20) * id; I[5] = (a02 * a10 - a00 * a12) * id;
  I[6] = c02 * id; I[7] = (a01 * a20 - a00 * a21) * id; I[8] = (a00 * a11 - a01 * a10) * id;
  float fro = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) fro += I[k] * I[k];
  return fro <= 3.96f;                                     // 1/||A^-1||_F >= 0.5025  =>  sigma_min(A) >= 0.5 (with margin)
}

// voxel key -> grid coordinate as ONE affine map: f = inv_cell (I (res (k + 1/2) - t) - origin) = A k + b.  (12 scalars and 9 FMAs per
// voxel instead of centre, difference, 3x3 product, offset and scale: the broad phase is 60 % of the grid kernel's instructions.  The
// rounding differs from the step-by-step form by ~1e-7 m, against a list margin of 3e-5 m.)
// Written to LDS (pl->Ab: A row-major, then b) and read back per pass of the voxel loop: as 12 scalars it would sit in scalar registers
// across the whole loop, which has none to spare.
__device__ __forceinline__ void grid_affine(const Grid& g, const float* T, const float* I, float res, PairList* pl, int lane) {
  float A[9], b[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float o = r == 0 ? g.ox : (r == 1 ? g.oy : g.oz);
#pragma unroll
    for (int c = 0; c < 3; ++c) A[r * 3 + c] = (I[r * 3 + c] * res) * g.inv_cell;
    b[r] = ((I[r * 3] * (0.5f * res - T[3]) + I[r * 3 + 1] * (0.5f * res - T[7])) + I[r * 3 + 2] * (0.5f * res - T[11]) - o) * g.inv_cell;
  }
  wave_lds_sync();
  if (lane == 0) {
    *(float4*)(pl->Ab) = make_float4(A[0], A[1], A[2], A[3]);
    *(float4*)(pl->Ab + 4) = make_float4(A[4], A[5], A[6], A[7]);
    *(float4*)(pl->Ab + 8) = make_float4(A[8], b[0], b[1], b[2]);
  }
  wave_lds_sync();
}

// narrow phase over the queued pairs, 64 at a time: every lane poses ITS triangle by T and runs the float32 SAT against ITS voxel
__device__ __forceinline__ bool wave_test_pairs(const Mesh& mesh, const PairList* pl, int fill, float res, int lane, unsigned* work) {
  const float h = 0.5f * res;
  wave_lds_sync();
  // The pose is read back from LDS here (a broadcast read per queue flush) instead of being held in 12 scalar registers across the
  // voxel loop: scalar registers are what this kernel is short of.
  float T[12];
#pragma unroll
  for (int k = 0; k < 12; k += 4) *(float4*)(T + k) = *(const float4*)(pl->T + k);
  for (int p0 = 0; p0 < fill; p0 += 64) {
    const int p = p0 + lane;
    bool hv = false;
    if (p < fill) {
      work[2] += 1u;
      const short4 k = pl->key[p];
      const float4* tv = (const float4*)(mesh.grid.tri_verts + (size_t)pl->tri[p] * 12);
      const float4 q0 = tv[0], q1 = tv[1], q2 = tv[2];
      float c[3], pv[9];
      const float m[9] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x};
      c[0] = ((float)k.x + 0.5f) * res; c[1] = ((float)k.y + 0.5f) * res; c[2] = ((float)k.z + 0.5f) * res;
#pragma unroll
      for (int kk = 0; kk < 3; ++kk)
#pragma unroll
        for (int r = 0; r < 3; ++r)
          pv[kk * 3 + r] = fmaf(T[r * 4 + 0], m[kk * 3], fmaf(T[r * 4 + 1], m[kk * 3 + 1], fmaf(T[r * 4 + 2], m[kk * 3 + 2], T[r * 4 + 3])));
      hv = tri_box_overlap(c, h, pv, pv + 3, pv + 6);
    }
    if (__ballot(hv) != 0ull) return true;
  }
  wave_lds_sync();                                         // the list is refilled next
  return false;
}

// wave-level: does the mesh posed by T (row-major 4x4, wave-uniform; I = inverse of its linear part) hit any occupied voxel?
// Broad phase: the lanes stride over the voxels, move each centre into the mesh frame and look its grid cell up; the triangles
// listed there are QUEUED as (voxel, triangle) pairs in LDS rather than tested by the lane that found them -- the lists are short,
// uneven and most voxels have none (6 % of the lane slots of a per-lane loop did useful work on the 9k-triangle gripper), so the
// narrow phase runs over the flat queue with every lane busy.  The narrow phase is the SAME float32 SAT on the SAME posed vertices
// as the exhaustive kernel, and "any pair hits" does not depend on the order of the pairs: identical results.
// Which of the 64 voxel blocks bb .. bb+63 can hold a voxel whose cell has a triangle list?  Lane j looks at block bb + j: the box of
// its keys, mapped into grid coordinates (centre by the affine map, extent by |A|), against the grid's bounds and -- when the box covers
// at most 3 x 3 x 3 coarse cells -- against the coarse occupancy.  Conservative (1e-3 cells of slack for the two roundings of f).
template <class VOX>
__device__ __forceinline__ unsigned long long relevant_blocks(const Grid& g, const PairList* pl, const VOX& vox, int nblocks, int bb, int lane) {
  float A[9], b[3];
  { const float4 q0 = *(const float4*)(pl->Ab), q1 = *(const float4*)(pl->Ab + 4), q2 = *(const float4*)(pl->Ab + 8);
    A[0] = q0.x; A[1] = q0.y; A[2] = q0.z; A[3] = q0.w; A[4] = q1.x; A[5] = q1.y; A[6] = q1.z; A[7] = q1.w; A[8] = q2.x; b[0] = q2.y; b[1] = q2.z; b[2] = q2.w; }
  const int blk = bb + lane;
  bool keep = blk < nblocks;
  const short* vblocks = vox.blocks();
  if (keep && vblocks) {
    const short4 lo = ((const short4*)vblocks)[2 * blk], hi = ((const short4*)vblocks)[2 * blk + 1];
    const float cx = 0.5f * ((float)lo.x + (float)hi.x), cy = 0.5f * ((float)lo.y + (float)hi.y), cz = 0.5f * ((float)lo.z + (float)hi.z);
    const float hx = 0.5f * ((float)hi.x - (float)lo.x), hy = 0.5f * ((float)hi.y - (float)lo.y), hz = 0.5f * ((float)hi.z - (float)lo.z);
    int c0[3], c1[3];
    const int n[3] = {g.nx, g.ny, g.nz};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float fc = fmaf(A[r * 3], cx, fmaf(A[r * 3 + 1], cy, fmaf(A[r * 3 + 2], cz, b[r])));
      const float rad = fmaf(fabsf(A[r * 3]), hx, fmaf(fabsf(A[r * 3 + 1]), hy, fabsf(A[r * 3 + 2]) * hz)) + 1e-3f;
      const float f0 = fc - rad, f1 = fc + rad;
      if (f1 < 0.f || f0 >= (float)n[r]) keep = false;
      c0[r] = f0 > 0.f ? (int)f0 : 0;
      c1[r] = f1 < (float)(n[r] - 1) ? (int)f1 : n[r] - 1;
    }
    if (keep && g.coarse) {
      const int s0x = c0[0] >> 2, s1x = c1[0] >> 2, s0y = c0[1] >> 2, s1y = c1[1] >> 2, s0z = c0[2] >> 2, s1z = c1[2] >> 2;
      if ((s1x - s0x) <= 2 && (s1y - s0y) <= 2 && (s1z - s0z) <= 2) {
        const int nsy = (g.ny + 3) >> 2, nsz = (g.nz + 3) >> 2;
        int occ = 0;
        for (int x = s0x; x <= s1x; ++x)
          for (int y = s0y; y <= s1y; ++y)
            for (int z = s0z; z <= s1z; ++z) occ |= g.coarse[(x * nsy + y) * nsz + z];
        keep = occ != 0;
      }
    }
  }
  return __ballot(keep);
}

template <class VOX>
__device__ __forceinline__ bool wave_grid_collide(const Mesh& mesh, const VOX& vox, float res, PairList* pl, int lane, unsigned* work) {
  const Grid& g = mesh.grid;
  int fill = 0;
  const int nblocks = (vox.nk() + 63) >> 6;
  // which blocks of 64 voxels are worth a visit: decided for the whole set up front (a bit per block, kept in LDS), so that the state of
  // that decision -- block boxes, coarse occupancy -- is not alive in the loop below
  const int nwords = (nblocks + 63) >> 6;
  wave_lds_sync();
  for (int wd = 0; wd < nwords && wd < BMASK_WORDS; ++wd) {
    const unsigned long long m = relevant_blocks(g, pl, vox, nblocks, wd * 64, lane);
    if (lane == 0) pl->bmask[wd] = m;
  }
  wave_lds_sync();
  int bb = -64;                                            // the 64 blocks under consideration: bb .. bb+63, `km` = those still to visit
  unsigned long long km = 0ull;
  // One loop, one place where the queue is tested: a pass either looks up the 2 x 64 voxels of the next two relevant blocks (two per
  // lane: the two dependent load chains key -> cell -> list run side by side) or, once the blocks are exhausted, only flushes the queue.
  for (;;) {
    int blk[2] = {-1, -1};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      while (km == 0ull && bb + 64 < nblocks) {
        bb += 64;
        const int wd = bb >> 6;
        const int left = nblocks - bb;                     // blocks beyond the up-front table (a set of > 262,144 voxels) are all visited
        unsigned long long w = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
        if (wd < BMASK_WORDS) {
          const unsigned long long t = pl->bmask[wd];        // the same word in every lane: keep it in scalar registers
          w = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(t >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)t);
        }
        km = w;
      }
      if (km != 0ull) { blk[u] = bb + __builtin_ctzll(km); km &= km - 1ull; }
    }
    const bool last = blk[0] < 0;
    int cnt[2] = {0, 0}, e0[2] = {0, 0};
    short4 k[2];
    float A[9], b[3];
    { const float4 q0 = *(const float4*)(pl->Ab), q1 = *(const float4*)(pl->Ab + 4), q2 = *(const float4*)(pl->Ab + 8);
      A[0] = q0.x; A[1] = q0.y; A[2] = q0.z; A[3] = q0.w; A[4] = q1.x; A[5] = q1.y; A[6] = q1.z; A[7] = q1.w; A[8] = q2.x; b[0] = q2.y; b[1] = q2.z; b[2] = q2.w; }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int v = blk[u] * 64 + lane;
      k[u] = make_short4(0, 0, 0, 0);
      if (blk[u] >= 0 && v < vox.nk()) {
        k[u] = ((const short4*)vox.keys())[v];
        work[0] += 1u;
        const float kx = (float)k[u].x, ky = (float)k[u].y, kz = (float)k[u].z;
        const float fx = fmaf(A[0], kx, fmaf(A[1], ky, fmaf(A[2], kz, b[0])));
        const float fy = fmaf(A[3], kx, fmaf(A[4], ky, fmaf(A[5], kz, b[1])));
        const float fz = fmaf(A[6], kx, fmaf(A[7], ky, fmaf(A[8], kz, b[2])));
        if (fx >= 0.f && fy >= 0.f && fz >= 0.f && fx < (float)g.nx && fy < (float)g.ny && fz < (float)g.nz) {
          const int cell = ((int)fx * g.ny + (int)fy) * g.nz + (int)fz;
          e0[u] = g.cell_start[cell];
          cnt[u] = g.cell_start[cell + 1] - e0[u];
          work[1] += 1u;
        }
      }
    }
    const int total = cnt[0] + cnt[1];
    for (int r = 0;;) {                                    // entry r of every lane's two lists, compacted into the queue
      const bool act = total > r;
      const unsigned long long m = __ballot(act);
      if (m != 0ull && fill + 64 <= PAIR_CAP) {
        if (act) {
          const int slot = fill + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
          const bool first = r < cnt[0];
          pl->key[slot] = first ? k[0] : k[1];
          pl->tri[slot] = g.tri_ids[first ? e0[0] + r : e0[1] + (r - cnt[0])];
        }
        fill += __popcll(m);
        ++r;
        continue;
      }
      // the lists of this pass are queued (m == 0), or the queue is full: test it when full, worth a round, or at the very end
      if (fill && (m != 0ull || last || fill >= PAIR_DRAIN)) {
        if (wave_test_pairs(mesh, pl, fill, res, lane, work)) return true;
        fill = 0;
      }
      if (m == 0ull) break;
    }
    if (last) return false;
  }
}

// Exhaustive form (no grid, or a pose outside the grid's validity): posed triangles staged in LDS, every voxel against every
// triangle whose box it touches.
template <class VOX>
__device__ bool wave_mesh_voxels_collide(const Mesh& mesh, const float* T, const VOX& vox, float res, float* tl, int lane) {
  if (vox.nk() == 0 || mesh.nf == 0) return false;
  const float h = 0.5f * res;
  const float slack = 1e-5f;     // conservative culls only; never changes the predicate
  bool hit = false;
  for (int c0 = 0; c0 < mesh.nf && !hit; c0 += TRI_CHUNK) {
    const int nt = min(TRI_CHUNK, mesh.nf - c0);
    wave_lds_sync();
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
    wave_lds_sync();
    for (int v0 = 0; v0 < vox.nk(); v0 += 64) {
      const int v = v0 + lane;
      bool hv = false;
      if (v < vox.nk()) {
        const short4 k = ((const short4*)vox.keys())[v];
        float c[3];
        c[0] = ((float)k.x + 0.5f) * res; c[1] = ((float)k.y + 0.5f) * res; c[2] = ((float)k.z + 0.5f) * res;
        int out = 0;                                   // (selects, not ||: a chain of boolean ORs lives in scalar register pairs)
#pragma unroll
        for (int r = 0; r < 3; ++r) { out = c[r] - h > hi[r] ? 1 : out; out = c[r] + h < lo[r] ? 1 : out; }
        if (!out) {
          for (int t = 0; t < nt && !hv; ++t) {
            const float* q = tl + t * TRI_FLOATS;
            int cull = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r) { cull = c[r] - h > q[12 + r] ? 1 : cull; cull = c[r] + h < q[9 + r] ? 1 : cull; }
            if (cull) continue;
            hv = tri_box_overlap(c, h, q, q + 3, q + 6);
          }
        }
      }
      if (__ballot(hv) != 0ull) { hit = true; break; }
    }
  }
  return hit;
}


// ---- stage 1: pose composition, one THREAD per evaluation (common.cpp:184-231) -------------------------------------------------
// grasp_in_cam = nocs_pose . canonical_to_nocs . symmetry_j . grasp_pose_i with unit rotation columns; the approach-direction test;
// the IK verdict of the host/device solver.  Writes the composed pose into poses_out and the code so far (0 / 1 / 2) into codes;
// stage 2 reads both.  (Every lane of a wave used to repeat this wave-uniform arithmetic ahead of its collision loops.)
struct ComposeArgs {
  const float* grasp_poses; int n_pose;       // (n_pose,16)
  const float* symmetry_tfs; int n_sym;       // (n_sym,16)
  Mat4 c2c, cam_in_world, ee_in_grasp;        // c2c = nocs_pose . canonical_to_nocs (host, same float32 operation order)
  int filter_dir;
  const unsigned char* ik_ok;                 // optional (E): 0 -> IK reject
  signed char* codes; float* poses_out; signed char* nudge;
  float* ee_out;                              // optional (E,16): ee_in_base for the IK pass (then nothing else is written but codes)
};

__global__ __launch_bounds__(256) void compose_grasp_pose_kernel(ComposeArgs a) {
  const long E = (long)a.n_pose * a.n_sym;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int i = (int)(e / a.n_sym), j = (int)(e - (long)i * a.n_sym);
  float P[16], S[16], tmp[16], gic[16];
#pragma unroll
  for (int k = 0; k < 16; k += 4) {
    *(float4*)(P + k) = *(const float4*)(a.grasp_poses + (size_t)i * 16 + k);
    *(float4*)(S + k) = *(const float4*)(a.symmetry_tfs + (size_t)j * 16 + k);
  }
  mat4_mul(S, P, tmp);
  mat4_mul(a.c2c.m, tmp, gic);
  normalize_col(gic, 0); normalize_col(gic, 1); normalize_col(gic, 2);
  int code = 0;
  if (a.filter_dir && gic[2 * 4 + 0] < 0.0f) code = 1;
  if (a.ee_out) {
    float t2[16], ee[16];
    mat4_mul(a.cam_in_world.m, gic, t2);
    mat4_mul(t2, a.ee_in_grasp.m, ee);
#pragma unroll
    for (int k = 0; k < 16; k += 4) *(float4*)(a.ee_out + e * 16 + k) = *(float4*)(ee + k);
    a.codes[e] = (signed char)code;
    return;
  }
  if (code == 0 && a.ik_ok && a.ik_ok[e] == 0) code = 2;
#pragma unroll
  for (int k = 0; k < 16; k += 4) *(float4*)(a.poses_out + e * 16 + k) = *(float4*)(gic + k);
  a.codes[e] = (signed char)code;
  a.nudge[e] = (signed char)-1;
}

// The same for the evaluations of SEVERAL filterGraspPose calls at once (cg_filter_grasp_pose_multi): evaluation e belongs to the segment
// whose [first, first + n_pose * n_sym) holds it (binary search in the device table), inside it e - first = i * n_sym + j.
struct ComposeMultiArgs {
  const cg_filter_segment* segs; int n_segs; long E;
  int filter_dir;
  const unsigned char* ik_ok;
  signed char* codes; float* poses_out; signed char* nudge;
};

__global__ __launch_bounds__(256) void compose_grasp_pose_multi_kernel(ComposeMultiArgs a) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.E) return;
  int lo = 0, hi = a.n_segs - 1;
  while (lo < hi) {                                        // the last segment whose first <= e
    const int mid = (lo + hi + 1) >> 1;
    if ((long)a.segs[mid].first <= e) lo = mid; else hi = mid - 1;
  }
  const cg_filter_segment& sg = a.segs[lo];
  const long le = e - (long)sg.first;
  const int i = (int)(le / sg.n_sym), j = (int)(le - (long)i * sg.n_sym);
  float P[16], S[16], C[16], tmp[16], gic[16];
#pragma unroll
  for (int k = 0; k < 16; k += 4) {
    *(float4*)(P + k) = *(const float4*)(sg.grasp_poses + (size_t)i * 16 + k);
    *(float4*)(S + k) = *(const float4*)(sg.symmetry_tfs + (size_t)j * 16 + k);
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) C[k] = sg.c2c[k];
  mat4_mul(S, P, tmp);
  mat4_mul(C, tmp, gic);
  normalize_col(gic, 0); normalize_col(gic, 1); normalize_col(gic, 2);
  int code = 0;
  if (a.filter_dir && gic[2 * 4 + 0] < 0.0f) code = 1;
  if (code == 0 && a.ik_ok && a.ik_ok[e] == 0) code = 2;
#pragma unroll
  for (int k = 0; k < 16; k += 4) *(float4*)(a.poses_out + e * 16 + k) = *(float4*)(gic + k);
  a.codes[e] = (signed char)code;
  a.nudge[e] = (signed char)-1;
}

// ---- stage 2: collision, one WAVEFRONT per evaluation that is still alive (common.cpp:233-299) -------------------------------
constexpr signed char CODE_PENDING = -128;    // written by the grid kernel for an evaluation it leaves to the exhaustive kernel

struct FilterArgs {
  long E;
  Mat4 gripper_in_grasp;
  int adjust;
  Mesh mesh[2];                               // open gripper, enclosed gripper
  Voxels vox[2];                              // ... against the object's own voxels, the background voxels
  float res;
  signed char* codes; float* poses_out; signed char* nudge;
  int keep_rejected_pose;                     // poses_out of a rejected evaluation: 0 -> zeros, 1 -> its (un-nudged) grasp_in_cam
  int only_pending;                           // exhaustive kernel: evaluate only what the grid kernel marked CODE_PENDING
  unsigned long long* work_stats;             // optional (3): voxel keys read, grid cells looked up, (voxel, triangle) pairs tested
  const cg_filter_segment* segs; int n_segs;  // MULTI: the device table; `adjust` and `vox` above are per segment then
};

// GRID: both meshes through their broad-phase grids (an evaluation whose pose a grid does not cover is marked CODE_PENDING);
// !GRID: the exhaustive collider.
// Fields of the kernel argument block that are touched once per evaluation (output pointers, flags) are read from the kernarg segment
// WHERE they are used (a volatile scalar load the compiler may not hoist): loaded up front they sat in ~20 scalar registers across
// the collision loops and were spilled to VGPR lanes.
// (an explicit s_load at a constant offset from the kernarg pointer: as a C++ volatile load the compiler materialised every field's
// ADDRESS in a scalar register pair and kept those alive instead)
template <typename T, int OFF>
__device__ __forceinline__ T karg_load(const __attribute__((address_space(4))) char* kargs) {
  static_assert(sizeof(T) == 4 || sizeof(T) == 8, "dword or qword fields");
  if constexpr (sizeof(T) == 8) {
    unsigned long long v;
    asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(kargs), "n"(OFF));
    if constexpr (__is_pointer(T)) return reinterpret_cast<T>(v); else return (T)v;
  } else {
    unsigned v;
    asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(kargs), "n"(OFF));
    return (T)v;
  }
}
#define CG_KARG(field) karg_load<decltype(FilterArgs::field), (int)offsetof(FilterArgs, field)>(kargs)

// MULTI (cg_filter_grasp_pose_multi): the evaluations of several calls in one launch -- the voxel sets and the nudge flag of an
// evaluation come from its segment's row of the device table (wave-uniform: scalar loads where they are used) instead of the kernel
// arguments; the gripper meshes, the resolution and gripper_in_grasp are the launch's.  Everything else is the same code.
template <bool GRID, bool MULTI>
__global__ __launch_bounds__(64 * WAVES, 4) void filter_grasp_pose_kernel(FilterArgs a) {
  const __attribute__((address_space(4))) char* kargs = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
  __shared__ __attribute__((aligned(16))) char lds_raw[WAVES * (GRID ? sizeof(PairList) : sizeof(float) * TRI_CHUNK * TRI_FLOATS)];   // cast to PairList (alignas 16)
  __shared__ float gig_lds[16];                // gripper_in_grasp: read back (broadcast) where a pose is composed, costs no register between
  __shared__ SegLds seg_lds[MULTI ? WAVES : 1];   // MULTI: the current segment of every wave (its voxel sets, nudge flag, table row)
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  PairList* pl = (PairList*)lds_raw + wv;
  float* tl = (float*)lds_raw + wv * TRI_CHUNK * TRI_FLOATS;
  if (threadIdx.x < 16) gig_lds[threadIdx.x] = a.gripper_in_grasp.m[threadIdx.x];
  if (MULTI && lane == 0) seg_lds[wv].sg = -1;
  __syncthreads();
  unsigned work[3] = {0u, 0u, 0u};             // per-lane counts of the grid kernel's memory work (reported only when asked for)
  // One wavefront per evaluation.  (One WORKGROUP per evaluation, its wavefronts dealing the voxel passes among themselves, was
  // measured: 2.92 instead of 2.38 ms per 50,004 evaluations -- the kernel is bound by the number of vector instructions it issues
  // (~40 % of the VALU issue rate of the whole chip), not by the length of one evaluation's dependent chain.)
  for (int e = (int)blockIdx.x * WAVES + wv; e < (int)CG_KARG(E); e += (int)gridDim.x * WAVES) {      // E < 2^31 (checked by the launcher)
    const int code0 = CG_KARG(codes)[e];
    if (GRID ? (code0 != 0) : (CG_KARG(only_pending) ? code0 != CODE_PENDING : code0 != 0)) {
      if (code0 > 0 && !CG_KARG(keep_rejected_pose) && lane == 0) {                                        // rejected in stage 1
        float4* po = (float4*)(CG_KARG(poses_out) + (size_t)e * 16);
        po[0] = po[1] = po[2] = po[3] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      continue;
    }
    // the nudge loop of common.cpp:255-288 (float accumulation 0, 0.001f, 0.002f; +step before -step); without
    // adjust_collision_pose only its first pose is tried and the mesh that collided names the code (common.cpp:233-253)
    int code = 0, nud = -1;
    float acc_t[3] = {0.f, 0.f, 0.f};
    bool found = false, stop = false;
    int idx = 0;
    if constexpr (MULTI) {       // the segment of evaluation e (e only grows: the search resumes at the wave's current row)
      SegLds* sl = seg_lds + wv;
      const cg_filter_segment* tab = CG_KARG(segs);
      const int ns = CG_KARG(n_segs);
      const int sg0 = __builtin_amdgcn_readfirstlane(sl->sg);
      int sg = sg0 < 0 ? 0 : sg0;
      while (sg + 1 < ns && (int)tab[sg + 1].first <= e) ++sg;
      if (sg != sg0) {
        wave_lds_sync();
        if (lane == 0) {
          const cg_filter_segment& row = tab[sg];
          sl->keys[0] = row.open_keys; sl->keys[1] = row.bg_keys; sl->blocks[0] = row.open_blocks; sl->blocks[1] = row.bg_blocks;
          sl->nk[0] = row.n_open_keys; sl->nk[1] = row.n_bg_keys; sl->adjust = row.adjust_collision_pose; sl->sg = sg;
        }
        wave_lds_sync();
      }
    }
    // common.cpp:255: `for (float step = 0; step <= 0.003; step += 0.001f)` visits 0, 0.001f and 0.001f + 0.001f -- the third
    // addition gives 0.0030000000261 > 0.003 (the comparison is in double) -- so the loop is counted here and `step` takes exactly
    // those three float values (one float compare and one conversion less per trip, both evaluated by the vector unit)
    for (int si = 0; si < 3 && !found && !stop; ++si) {
      const float step = si == 0 ? 0.0f : (si == 1 ? 0.001f : 0.001f + 0.001f);
      const int nsign = si == 0 ? 1 : 2;
      for (int s = 0; s < nsign && !found && !stop; ++s, ++idx) {
        const float sign = (s == 0) ? 1.0f : -1.0f;
        // grasp_in_cam (stage 1 left it in poses_out) is re-read per tried pose rather than kept: scalar registers are what the
        // collision loops are short of
        float cur[16], gig[16], gcam[16], cur_t[3];
#pragma unroll
        for (int k = 0; k < 12; k += 4) *(float4*)(cur + k) = *(const float4*)(CG_KARG(poses_out) + (size_t)e * 16 + k);
        cur[12] = 0.f; cur[13] = 0.f; cur[14] = 0.f; cur[15] = 1.f;      // rows 0..2 of cur . gig do not read row 3 of cur
#pragma unroll
        for (int r = 0; r < 3; ++r) { cur[r * 4 + 3] = cur[r * 4 + 3] + (step * cur[r * 4 + 1]) * sign; cur_t[r] = cur[r * 4 + 3]; }
#pragma unroll
        for (int k = 0; k < 16; ++k) gig[k] = gig_lds[k];
        mat4_mul(cur, gig, gcam);
        if (GRID) {          // the posed-gripper matrix goes to LDS and is read back where it is used
          wave_lds_sync();
          if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 12; k += 4) *(float4*)(pl->T + k) = *(float4*)(gcam + k);
          }
          wave_lds_sync();
        }
        int hit = 0;
#pragma unroll 1
        for (int m = 0; m < 2 && !hit; ++m) {
          const Mesh& mesh = a.mesh[m];
          VoxView<MULTI> vox = [&]() { if constexpr (MULTI) return VoxView<true>{seg_lds + wv, m}; else return VoxView<false>{a.vox[m]}; }();
          if (vox.nk() == 0 || mesh.nf == 0) continue;
          if (GRID) {
            float T[12];
#pragma unroll
            for (int k = 0; k < 12; k += 4) *(float4*)(T + k) = *(const float4*)(pl->T + k);
            float I[9];
            const bool ok = mesh.has_grid && grid_usable(mesh.grid, T, a.res, I);
            if (!ok) { code = CODE_PENDING; stop = true; break; }
            grid_affine(mesh.grid, T, I, a.res, pl, lane);
            if (wave_grid_collide(mesh, vox, a.res, pl, lane, work)) hit = 3 + m;
          } else {
            if (wave_mesh_voxels_collide(mesh, gcam, vox, a.res, tl, lane)) hit = 3 + m;
          }
        }
        if (stop) break;
        if (!hit) {
#pragma unroll
          for (int r = 0; r < 3; ++r) acc_t[r] = cur_t[r];              // cur differs from grasp_in_cam in its translation only
          found = true; nud = idx;
        } else if (!(MULTI ? seg_lds[wv].adjust : CG_KARG(adjust))) {
          code = hit; stop = true;
        }
      }
    }
    if (!found && !stop) code = 3;
    if (GRID && code == CODE_PENDING) {
      if (lane == 0) CG_KARG(codes)[e] = CODE_PENDING;
      continue;
    }
    if (code == 0) {                                                    // the accepted (possibly nudged) translation
      if (lane == 0) {
        float* po = CG_KARG(poses_out) + (size_t)e * 16;
        po[3] = acc_t[0]; po[7] = acc_t[1]; po[11] = acc_t[2];
      }
    } else if (!CG_KARG(keep_rejected_pose)) {
      if (lane == 0) {
        float4* po = (float4*)(CG_KARG(poses_out) + (size_t)e * 16);
        po[0] = po[1] = po[2] = po[3] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (lane == 0) { CG_KARG(codes)[e] = (signed char)code; CG_KARG(nudge)[e] = (signed char)nud; }
  }
  if (GRID && CG_KARG(work_stats)) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      unsigned w = work[q];
      for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o);
      if (lane == 0 && w) atomicAdd(CG_KARG(work_stats) + q, (unsigned long long)w);
    }
  }
}

// CollisionManager.isAnyCollision for E independent mesh poses against one voxel set
__global__ __launch_bounds__(64 * WAVES) void mesh_voxels_collide_kernel(Mesh mesh, const float* poses, long E, Voxels vox,
                                                                         float res, unsigned char* out) {
  __shared__ float tl_all[WAVES][TRI_CHUNK * TRI_FLOATS];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (long e = (long)blockIdx.x * WAVES + wv; e < E; e += (long)gridDim.x * WAVES) {
    float T[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) T[k] = poses[e * 16 + k];
    const bool hit = wave_mesh_voxels_collide(mesh, T, VoxView<false>{vox}, res, tl_all[wv], lane);
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
  if (hg && hg->cell_start && hg->tri_ids && hg->tri_verts && hg->cell > 0.f) {      // a grid without the flat triangle array is not used
    m.grid = Grid{hg->origin[0], hg->origin[1], hg->origin[2], 1.0f / hg->cell, hg->dims[0], hg->dims[1], hg->dims[2], hg->cell_start,
                  hg->tri_ids, hg->tri_verts, hg->resolution, hg->coarse_occupancy};
    m.has_grid = 1;
  }
  return m;
}

// Workgroups the collision kernels are launched with at most (grid-stride beyond): 16 per CU.  CATGRASP_AMD_FILTER_BLOCKS_PER_CU (dev
// knob) changes it: with MORE workgroups than the chip holds at once the dispatcher hands evaluations out as slots free up.
inline long filter_block_cap() {
  static const long per_cu = getenv("CATGRASP_AMD_FILTER_BLOCKS_PER_CU") ? atol(getenv("CATGRASP_AMD_FILTER_BLOCKS_PER_CU")) : 32;
  return 256 * (per_cu > 0 ? per_cu : 32);
}

inline Mat4 load_mat(const float* h) { Mat4 m; for (int i = 0; i < 16; ++i) m.m[i] = h[i]; return m; }

// mat4_mul of the device code on the host: the same float32 expression per element (this file is compiled with -ffp-contract=off)
inline void host_mat4_mul(const float* A, const float* B, float* C) {
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c)
      C[r * 4 + c] = ((A[r * 4 + 0] * B[0 * 4 + c] + A[r * 4 + 1] * B[1 * 4 + c]) + A[r * 4 + 2] * B[2 * 4 + c]) + A[r * 4 + 3] * B[3 * 4 + c];
}

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
                                    int keep_rejected_pose, unsigned long long* work_stats, const short* open_blocks, const short* bg_blocks,
                                    void* stream) {
  if (n_pose < 0 || n_sym < 0) return CG_ERR_ARG;
  if ((long)n_pose * n_sym == 0) return CG_OK;
  if (!grasp_poses || !symmetry_tfs || !h_nocs_pose || !h_canonical_to_nocs || !h_cam_in_world || !h_ee_in_grasp ||
      !h_gripper_in_grasp || !codes)
    return CG_ERR_ARG;
  if ( n_gripper_faces < 0 || n_enclosed_faces < 0 || n_open_keys < 0 || n_bg_keys < 0) return CG_ERR_ARG;
  if (!ee_in_base_out && (!poses_out || !nudge)) return CG_ERR_ARG;
  if ((((uintptr_t)grasp_poses | (uintptr_t)symmetry_tfs | (uintptr_t)poses_out | (uintptr_t)ee_in_base_out) & 15) != 0) return CG_ERR_ARG;   // 16-byte rows
  if ((n_gripper_faces > 0 && (!gripper_vertices || !gripper_faces)) || (n_enclosed_faces > 0 && (!enclosed_vertices || !enclosed_faces)))
    return CG_ERR_ARG;
  if ((n_open_keys > 0 && !open_keys) || (n_bg_keys > 0 && !bg_keys)) return CG_ERR_ARG;
  if (!(resolution > 0.f)) return CG_ERR_ARG;
  const long E = (long)n_pose * n_sym;
  if (E == 0) return CG_OK;
  if (E >= (1L << 31) - 64 * 1024) return CG_ERR_ARG;      // evaluation indices are 32-bit in the kernels
  hipStream_t st = (hipStream_t)stream;
  ComposeArgs c;
  c.grasp_poses = grasp_poses; c.n_pose = n_pose; c.symmetry_tfs = symmetry_tfs; c.n_sym = n_sym;
  host_mat4_mul(h_nocs_pose, h_canonical_to_nocs, c.c2c.m);
  c.cam_in_world = load_mat(h_cam_in_world); c.ee_in_grasp = load_mat(h_ee_in_grasp);
  c.filter_dir = filter_approach_dir_face_camera; c.ik_ok = ik_ok;
  c.codes = codes; c.poses_out = poses_out; c.nudge = nudge; c.ee_out = ee_in_base_out;
  hipLaunchKernelGGL(compose_grasp_pose_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, c);
  if (ee_in_base_out) return cg_hip_status(hipGetLastError());
  FilterArgs a;
  a.E = E;
  a.gripper_in_grasp = load_mat(h_gripper_in_grasp);
  a.adjust = adjust_collision_pose;
  a.mesh[0] = make_mesh(gripper_vertices, gripper_faces, n_gripper_faces, h_open_grid);
  a.mesh[1] = make_mesh(enclosed_vertices, enclosed_faces, n_enclosed_faces, h_enc_grid);
  a.vox[0] = Voxels{open_keys, n_open_keys, open_blocks}; a.vox[1] = Voxels{bg_keys, n_bg_keys, bg_blocks};
  a.res = resolution; a.codes = codes; a.poses_out = poses_out; a.nudge = nudge;
  a.keep_rejected_pose = keep_rejected_pose; a.work_stats = work_stats;
  long blocks = (E + WAVES - 1) / WAVES;
  if (blocks > filter_block_cap()) blocks = filter_block_cap();      // grid-stride: 16 blocks per CU
  // Meshes that collide with anything (non-empty mesh against a non-empty voxel set) must all carry a grid for the grid kernel;
  // evaluations it cannot cover (a pose outside a grid's validity) come back CODE_PENDING and the exhaustive kernel, which skips
  // everything else, finishes them.  Without grids the exhaustive kernel does all of it.
  bool grids = true;
  for (int m = 0; m < 2; ++m)
    if (a.mesh[m].nf > 0 && a.vox[m].nk > 0 && !a.mesh[m].has_grid) grids = false;
  a.only_pending = 0;
  if (grids) {
    hipLaunchKernelGGL((filter_grasp_pose_kernel<true, false>), dim3((unsigned)blocks), dim3(64 * WAVES), 0, st, a);
    a.only_pending = 1;
  }
  hipLaunchKernelGGL((filter_grasp_pose_kernel<false, false>), dim3((unsigned)blocks), dim3(64 * WAVES), 0, st, a);
  return cg_hip_status(hipGetLastError());
}

// ---- several filterGraspPose calls in ONE launch sequence ------------------------------------------------------------------------------
// A pick cycle filters every object of the scene twice (cone poses; canonical grasps x symmetries) against that object's own voxel
// sets: 2 x objects calls of a few thousand evaluations each, three launches per call, every one too small to fill 256 CUs.  The
// segments of all those calls go through the three kernels together.
extern "C" long cg_filter_segments_prepare(cg_filter_segment* h_segments, int n_segments) {
  if (n_segments < 0 || (n_segments > 0 && !h_segments)) return CG_ERR_ARG;
  long E = 0;
  for (int s = 0; s < n_segments; ++s) {
    cg_filter_segment& g = h_segments[s];
    if (g.n_pose < 0 || g.n_sym < 0 || g.n_open_keys < 0 || g.n_bg_keys < 0) return CG_ERR_ARG;
    const long n = (long)g.n_pose * g.n_sym;
    if (n > 0 && (!g.grasp_poses || !g.symmetry_tfs)) return CG_ERR_ARG;
    if ((((uintptr_t)g.grasp_poses | (uintptr_t)g.symmetry_tfs) & 15) != 0) return CG_ERR_ARG;
    if ((g.n_open_keys > 0 && !g.open_keys) || (g.n_bg_keys > 0 && !g.bg_keys)) return CG_ERR_ARG;
    host_mat4_mul(g.nocs_pose, g.canonical_to_nocs, g.c2c);
    g.first = E;
    E += n;
    if (E >= (1L << 31) - 64 * 1024) return CG_ERR_ARG;
  }
  return E;
}

extern "C" int cg_filter_grasp_pose_multi(const cg_filter_segment* h_segments, const cg_filter_segment* d_segments, int n_segments,
                                          const float* h_gripper_in_grasp, int filter_approach_dir_face_camera, const unsigned char* ik_ok,
                                          const float* gripper_vertices, const int* gripper_faces, int n_gripper_faces,
                                          const float* enclosed_vertices, const int* enclosed_faces, int n_enclosed_faces,
                                          float resolution, signed char* codes, float* poses_out, signed char* nudge,
                                          const cg_mesh_grid* h_open_grid, const cg_mesh_grid* h_enc_grid, int keep_rejected_pose,
                                          unsigned long long* work_stats, void* stream) {
  if (n_segments < 0) return CG_ERR_ARG;
  if (n_segments == 0) return CG_OK;
  if (!h_segments || !d_segments || !h_gripper_in_grasp || !codes || !poses_out || !nudge) return CG_ERR_ARG;
  if (n_gripper_faces < 0 || n_enclosed_faces < 0 || !(resolution > 0.f)) return CG_ERR_ARG;
  if ((n_gripper_faces > 0 && (!gripper_vertices || !gripper_faces)) || (n_enclosed_faces > 0 && (!enclosed_vertices || !enclosed_faces)))
    return CG_ERR_ARG;
  if (((uintptr_t)poses_out & 15) != 0) return CG_ERR_ARG;
  // the table must be a prepared one: firsts consecutive from 0 (cg_filter_segments_prepare wrote them)
  long E = 0;
  bool any_open = false, any_bg = false;
  for (int s = 0; s < n_segments; ++s) {
    const cg_filter_segment& g = h_segments[s];
    if (g.n_pose < 0 || g.n_sym < 0 || (long)g.first != E) return CG_ERR_ARG;
    E += (long)g.n_pose * g.n_sym;
    any_open |= g.n_open_keys > 0; any_bg |= g.n_bg_keys > 0;
  }
  if (E == 0) return CG_OK;
  if (E >= (1L << 31) - 64 * 1024) return CG_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  ComposeMultiArgs c{d_segments, n_segments, E, filter_approach_dir_face_camera, ik_ok, codes, poses_out, nudge};
  hipLaunchKernelGGL(compose_grasp_pose_multi_kernel, dim3((unsigned)((E + 255) / 256)), dim3(256), 0, st, c);
  FilterArgs a;
  a.E = E;
  a.gripper_in_grasp = load_mat(h_gripper_in_grasp);
  a.adjust = 0;
  a.mesh[0] = make_mesh(gripper_vertices, gripper_faces, n_gripper_faces, h_open_grid);
  a.mesh[1] = make_mesh(enclosed_vertices, enclosed_faces, n_enclosed_faces, h_enc_grid);
  a.vox[0] = Voxels{nullptr, 0, nullptr}; a.vox[1] = Voxels{nullptr, 0, nullptr};
  a.res = resolution; a.codes = codes; a.poses_out = poses_out; a.nudge = nudge;
  a.keep_rejected_pose = keep_rejected_pose; a.work_stats = work_stats;
  a.segs = d_segments; a.n_segs = n_segments;
  long blocks = (E + WAVES - 1) / WAVES;
  if (blocks > filter_block_cap()) blocks = filter_block_cap();
  bool grids = true;
  if (a.mesh[0].nf > 0 && any_open && !a.mesh[0].has_grid) grids = false;
  if (a.mesh[1].nf > 0 && any_bg && !a.mesh[1].has_grid) grids = false;
  a.only_pending = 0;
  if (grids) {
    hipLaunchKernelGGL((filter_grasp_pose_kernel<true, true>), dim3((unsigned)blocks), dim3(64 * WAVES), 0, st, a);
    a.only_pending = 1;
  }
  hipLaunchKernelGGL((filter_grasp_pose_kernel<false, true>), dim3((unsigned)blocks), dim3(64 * WAVES), 0, st, a);
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
                     make_mesh(vertices, faces, n_faces, nullptr), poses, n_poses, Voxels{keys, n_keys, nullptr}, resolution, out);
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
                                    bg_keys, n_bg_keys, resolution, codes, poses_out, nudge, ee_in_base_out, nullptr, nullptr, 0, nullptr, nullptr, nullptr, stream);
}
