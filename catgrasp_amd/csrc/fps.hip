// farthest_point_sample of the reference's pointnet2.py:54-75 as CDNA4 HIP kernels: sequential over npoint, one workgroup per cloud, the
// cloud's points in VGPRs.  Float arithmetic mirrors the reference expression term by term (no FMA contraction: built with
// -ffp-contract=off), so the samples are exact.  This file alone is built with -amdgpu-use-divergent-register-indexing (build.py): the
// winner's coordinates are read out of register vectors with a wave-uniform index (s_set_gpr_idx_on + v_mov) instead of the select chain
// LLVM otherwise expands an 8-element extract into.
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"
#include <stdlib.h>

namespace {

// ---------------------------------------------------------------- farthest point sampling
// One workgroup per cloud.  Thread t owns points t, t+NT, ... (PPT of them) in registers.  Each of the npoint rounds: update the
// running min distance to the chosen set, then one arg-max over the cloud (first index on ties).  The round is a dependent chain on
// ONE CU (splitting a cloud over CUs would put a >= 4 us device-scope barrier into a ~1.5 us round: MI355X_MICROARCH.md, barrier
// table), so what counts is the VALU work per point and the latency of the exchange:
//  * inner loop, 5.5 VALU per point: two points per packed-f32 instruction for the bit-exact (dx*dx + dy*dy) + dz*dz (3 v_pk_add,
//    3 v_pk_mul, 2 v_pk_add per PAIR), one v_min_u32 per point for the running distance and ONE v_max3_u32 per pair for the running
//    maximum -- which slot holds the maximum is not tracked (that was a v_cmp + 2 v_cndmask per point, 3 of 8 VALU): the slots are
//    kept in groups of 8 with a maximum per group, and after the wave reduction the winner lane's group is made wave-uniform
//    (v_readlane) and searched (7 compare + select), in that group's branch only;
//  * the reduction travels as a 32-bit VALUE (v_max_u32 with the DPP row operation folded in by hand: 6 steps per wave -- the
//    builtin compiles to v_mov + v_mov_dpp + v_max per step); the winner lane is the single lane that holds it -- ties (duplicate
//    points, lattices) take a wave-uniform slow path that finds every lane's first slot and reduces the indices too, so the result
//    is still "first index among equal maxima";
//  * only the wave's winner needs its coordinates: its slot number is made wave-uniform (v_readlane) and a uniform binary search
//    inside the group picks the slot's registers, no per-slot select;
//  * the winners of the waves meet in LDS as (value, x, y, z) + index, ONE workgroup barrier per round (double-buffered), and the next
//    centre is taken out of the exchanged records with v_readlane -- no dependent LDS or global read at the top of the round.
// History (N = 20,000 -> 1,024, one cloud): 4.19 ms (round 1) -> 2.71 ms (round 2: 64-bit DPP keys) -> 1.73 ms (round 3: packed math,
// branch-free arg-max, 32-bit reduction on unsigned bit patterns) -> 1.58 ms (maximum only, slot looked up afterwards) -> 1.51 ms
// (hand-folded DPP): 1.47 us per round.  Ablations on the device (profiles/r3_fps_ablation.txt): distance update + wave reduction
// alone 0.73 us; + the workgroup exchange (LDS, barrier, second reduction -- and the lockstep it forces, which exposes every latency of
// the chain) + 0.54 us; + slot search and register pick + 0.27 us.  Measured and rejected: 1,024 threads x 20 points (1.65 us),
// select-chain instead of branch pick (1.60), vector-typed storage for an indexed register read (LLVM emits the same select chain:
// 1.48), batching the index stores (no change: the store is off the critical path).
// Running distances are >= +0 (sums of squares; 1e10 initially), so their BIT PATTERNS order like the values: min / max / compare run
// on them as unsigned integers -- one v_min_u32 / v_max_u32 where the float forms cost a compare + select or drag a canonicalising
// v_max along.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_min_i32(int v) {
  const int o = __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
  return o < v ? o : v;
}
// max / min over the 16 lanes of every DPP row (all lanes of the row get it)
// the DPP operand folded into v_max_u32 by hand (the compiler emits v_mov_b32 + v_mov_b32_dpp + v_max_u32 for the builtin); the two
// wait states a DPP read needs after a VALU write of the same register are ours to insert here
__device__ __forceinline__ unsigned row_max_u32(unsigned v) {
  asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
  return v;
}
// the same over the 8 lanes of every half row
__device__ __forceinline__ unsigned half_row_max_u32(unsigned v) {
  asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(v));
  return v;
}
__device__ __forceinline__ int row_min_i32(int v) {
  v = dpp_min_i32<0xB1, 0xf>(v); v = dpp_min_i32<0x4E, 0xf>(v); v = dpp_min_i32<0x141, 0xf>(v);
  return dpp_min_i32<0x140, 0xf>(v);
}
// over the wavefront, returned wave-uniform
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = row_max_u32(v);
  asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 0" : "+v"(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ int wave_min_i32(int v) {
  v = row_min_i32(v);
  v = dpp_min_i32<0x142, 0xa>(v);
  v = dpp_min_i32<0x143, 0xc>(v);
  return __builtin_amdgcn_readlane(v, 63);
}

// ---------------------------------------------------------------- small clouds: ONE wavefront per cloud (round 5)
// A cloud of <= 512 points -- the second level of a PointNet++ stack samples 128 of 512 -- fits one wavefront's registers (lane l
// holds points l, l + 64, ...).  With a single wavefront a round needs no LDS exchange and no workgroup barrier, which are what a round
// of fps_kernel<512, 4> mostly consists of at these sizes (0.6 us per round for 512 points, whatever the arithmetic): distance update,
// one DPP max over the wave, a search for the first slot that holds it, one DPP min over the candidate indices ("first index among
// equal maxima", pointnet2.py:74), the new centre's coordinates by a broadcast LDS read of the staged cloud.  Same float expression,
// same unsigned-bit-pattern compares as the kernels above: same samples.
template <int SLOTS>
__global__ __launch_bounds__(64) void fps_wave_kernel(const float* __restrict__ xyz, const long long* __restrict__ start, int N, int npoint,
                                                      long long* __restrict__ out, float* __restrict__ out_xyz) {
  static_assert(SLOTS % 2 == 0, "points are held in pairs (packed float math)");
  constexpr int H = SLOTS / 2;
  extern __shared__ float pts[];                     // the cloud, (N,3): where a round reads its centre from
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* xb = xyz + (size_t)b * N * 3;
  for (int i = lane; i < N * 3; i += 64) pts[i] = xb[i];
  // lane l holds the CONSECUTIVE points l * SLOTS .. l * SLOTS + SLOTS - 1: index order is (lane, slot) order, so "the first index among
  // equal maxima" is the first lane that holds the maximum (one ballot + s_ff1) and that lane's first slot -- no reduction over indices
  f32x2 px[H], py[H], pz[H];
  unsigned dist[SLOTS];                              // bit patterns of the running distances (all >= +0)
#pragma unroll
  for (int k = 0; k < SLOTS; ++k) {
    const int p = lane * SLOTS + k;
    float x = 0.f, y = 0.f, z = 0.f, d0 = 0.0f;      // padding (the highest indices): distance 0 never shrinks; it can only tie at maximum 0,
    if (p < N) { x = xb[p * 3 + 0]; y = xb[p * 3 + 1]; z = xb[p * 3 + 2]; d0 = 1e10f; }       // where point 0 (lane 0, slot 0) comes first
    px[k >> 1][k & 1] = x; py[k >> 1][k & 1] = y; pz[k >> 1][k & 1] = z; dist[k] = __float_as_uint(d0);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  int far = (int)start[b];
  long long* ob = out + (size_t)b * npoint;
  float* oxb = out_xyz ? out_xyz + (size_t)b * npoint * 3 : nullptr;
  for (int it = 0; it < npoint; ++it) {
    const float cx = pts[far * 3 + 0], cy = pts[far * 3 + 1], cz = pts[far * 3 + 2];
    if (lane == 0) {
      ob[it] = far;
      if (oxb) { oxb[it * 3 + 0] = cx; oxb[it * 3 + 1] = cy; oxb[it * 3 + 2] = cz; }
    }
    const f32x2 cx2 = {cx, cx}, cy2 = {cy, cy}, cz2 = {cz, cz};
    unsigned best = 0u;
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const f32x2 dx = px[h] - cx2, dy = py[h] - cy2, dz = pz[h] - cz2;
      const f32x2 d = (dx * dx + dy * dy) + dz * dz;                            // the reference's sum over the last axis, term by term
      const unsigned d0 = __float_as_uint(d[0]), d1 = __float_as_uint(d[1]);
      dist[2 * h] = d0 < dist[2 * h] ? d0 : dist[2 * h];
      dist[2 * h + 1] = d1 < dist[2 * h + 1] ? d1 : dist[2 * h + 1];
      const unsigned m = dist[2 * h] > dist[2 * h + 1] ? dist[2 * h] : dist[2 * h + 1];
      best = m > best ? m : best;
    }
    const unsigned wmax = wave_max_u32(best);
    int ms = 0;
#pragma unroll
    for (int k = SLOTS - 1; k >= 0; --k) ms = dist[k] == wmax ? k : ms;        // this lane's first slot holding the maximum (if any)
    const unsigned long long holders = __ballot(best == wmax);
    const int L = (int)__builtin_ctzll(holders);                                // never empty: some lane holds the maximum
    far = L * SLOTS + __builtin_amdgcn_readlane(ms, L);
  }
}

// registers of slot k (wave-uniform k) by a uniform binary search: log2(PPT) scalar branches instead of a select per slot
template <int LO, int HI, int H>
__device__ __forceinline__ void fps_pick(int k, const f32x2 (&px)[H], const f32x2 (&py)[H], const f32x2 (&pz)[H], float& x, float& y, float& z) {
  if constexpr (HI - LO == 1) {
    x = px[LO >> 1][LO & 1]; y = py[LO >> 1][LO & 1]; z = pz[LO >> 1][LO & 1];
    asm volatile("" : "+v"(x), "+v"(y), "+v"(z));     // keeps the leaves apart: merged, they become a dynamically indexed array in scratch
  }
  else {
    constexpr int MID = (LO + HI) / 2;
    if (k < MID) fps_pick<LO, MID, H>(k, px, py, pz, x, y, z);
    else fps_pick<MID, HI, H>(k, px, py, pz, x, y, z);
  }
}

// the winner lane's first slot at the lane maximum `bv`, searched in group gw only (wave-uniform gw: a chain of scalar branches over
// the groups, then GS-1 compare + select in every lane), and that slot's registers
template <int G, int NG, int GS, int PPT, int H>
__device__ __forceinline__ void fps_find(int gw, int wl, unsigned bv, const unsigned (&dist)[PPT], const f32x2 (&px)[H], const f32x2 (&py)[H],
                                         const f32x2 (&pz)[H], int& kw, float& x, float& y, float& z) {
  if (G == NG - 1 || gw == G) {
    asm volatile("" : "+v"(bv));                            // the search stays inside its branch (hoisted, all NG of them run every round)
    int k = G * GS + GS - 1;
#pragma unroll
    for (int j = GS - 2; j >= 0; --j) k = dist[G * GS + j] == bv ? G * GS + j : k;
    kw = __builtin_amdgcn_readlane(k, wl);
    fps_pick<G * GS, G * GS + GS, H>(kw, px, py, pz, x, y, z);
  } else if constexpr (G < NG - 1) {
    fps_find<G + 1, NG, GS, PPT, H>(gw, wl, bv, dist, px, py, pz, kw, x, y, z);
  }
}

// NT threads, thread t owns points t, t+NT, ...; PPT even.
template <int NT, int PPT>
__global__ __launch_bounds__(NT) void fps_kernel(const float* __restrict__ xyz, const long long* __restrict__ start, int N, int npoint,
                                                 long long* __restrict__ out, float* __restrict__ out_xyz) {
  static_assert(PPT % 2 == 0 && NT % 64 == 0 && NT <= 1024, "geometry");
  constexpr int H = PPT / 2;
  constexpr int GS = PPT < 8 ? PPT : 8, NG = PPT / GS;
  static_assert(PPT % GS == 0, "slots come in whole groups");
  __shared__ f32x4 red_v[2][16];       // per wave: (bits of the best distance, x, y, z) of its winner
  __shared__ int red_i[2][16];         //           its point index
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid < 32) { red_v[tid >> 4][tid & 15] = f32x4{0.f, 0.f, 0.f, 0.f}; red_i[tid >> 4][tid & 15] = 0x7fffffff; }   // absent waves: distance 0, index "none"
  __syncthreads();
  const float* xb = xyz + (size_t)b * N * 3;
  f32x2 px[H], py[H], pz[H];
  unsigned dist[PPT];                  // bit patterns of the running distances (all >= +0)
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int p = tid + k * NT;
    float x = 0.f, y = 0.f, z = 0.f, d0 = 0.0f;     // padding: running distance 0 never shrinks (d >= 0) and never beats a real point first
    if (p < N) { x = xb[p * 3 + 0]; y = xb[p * 3 + 1]; z = xb[p * 3 + 2]; d0 = 1e10f; }
    px[k >> 1][k & 1] = x; py[k >> 1][k & 1] = y; pz[k >> 1][k & 1] = z;
    dist[k] = __float_as_uint(d0);
  }
  int farthest = (int)start[b];
  float cx = xb[farthest * 3 + 0], cy = xb[farthest * 3 + 1], cz = xb[farthest * 3 + 2];
  for (int it = 0; it < npoint; ++it) {
    if (tid == 0) {
      out[(size_t)b * npoint + it] = farthest;
      if (out_xyz) { float* o = out_xyz + ((size_t)b * npoint + it) * 3; o[0] = cx; o[1] = cy; o[2] = cz; }   // = index_points(xyz, out), for free
    }
    const f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
    // the thread's slots in groups of GS: only the running MAXIMUM is tracked while the distances are updated (one v_max3_u32 per
    // pair), per group and over all; which slot holds it is looked up afterwards, in the winner's group only
    unsigned gmax[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      unsigned m = 0u;
#pragma unroll
      for (int h = g * (GS / 2); h < (g + 1) * (GS / 2); ++h) {
        const f32x2 dx = px[h] - c2x, dy = py[h] - c2y, dz = pz[h] - c2z;
        const f32x2 d = (dx * dx + dy * dy) + dz * dz;      // torch.sum((xyz - centroid) ** 2, -1), two points per instruction
        const unsigned d0 = __float_as_uint(d[0]), d1 = __float_as_uint(d[1]);
        const unsigned n0 = d0 < dist[2 * h] ? d0 : dist[2 * h];              // mask = dist < distance; distance[mask] = dist[mask]
        const unsigned n1 = d1 < dist[2 * h + 1] ? d1 : dist[2 * h + 1];
        dist[2 * h] = n0; dist[2 * h + 1] = n1;
        const unsigned a = m > n0 ? m : n0;
        m = a > n1 ? a : n1;
      }
      gmax[g] = m;
    }
    unsigned bv = gmax[0];
#pragma unroll
    for (int g = 1; g < NG; ++g) bv = bv > gmax[g] ? bv : gmax[g];
    // ---- wave: who holds the largest running distance (smallest index among equals) ----
    const unsigned wmax = wave_max_u32(bv);
    unsigned long long cand = __ballot(bv == wmax);
    if (__builtin_popcountll(cand) != 1) {                   // ties inside the wave: the smallest point index wins
      unsigned bt = bv;
      asm volatile("" : "+v"(bt));                           // keeps the slot search of the rare path from being hoisted into every round
      int bk = PPT - 1;                                      // the lane's first slot at its maximum (ascending slot = ascending index)
#pragma unroll
      for (int k = PPT - 2; k >= 0; --k) bk = dist[k] == bt ? k : bk;
      const int bi = tid + bk * NT;
      const int mi = wave_min_i32(bv == wmax ? bi : 0x7fffffff);
      cand = __ballot(bv == wmax && bi == mi);
    }
    const int wl = __builtin_ctzll(cand);
    int gi = NG - 1;                                         // per lane: the first group that holds the lane's maximum
#pragma unroll
    for (int g = NG - 2; g >= 0; --g) gi = gmax[g] == bv ? g : gi;
    const int gw = __builtin_amdgcn_readlane(gi, wl);
    int kw; float bx, by, bz;
    fps_find<0, NG, GS, PPT, H>(gw, wl, bv, dist, px, py, pz, kw, bx, by, bz);   // the winner's slot (wave-uniform) and coordinates
    const int iw = (wv * 64 + wl) + kw * NT;
    const int buf = it & 1;
    if (lane == wl) { red_v[buf][wv] = f32x4{__uint_as_float(wmax), bx, by, bz}; red_i[buf][wv] = iw; }
    __syncthreads();
    // ---- workgroup: the same over the (<= 16) wave winners; every wave redoes it on its own copy ----
    const f32x4 e = red_v[buf][lane & 15];
    const int ei = red_i[buf][lane & 15];
    const unsigned ev = __float_as_uint(e[0]);
    const unsigned best = row_max_u32(ev);
    unsigned c16 = (unsigned)__ballot(ev == best) & 0xffffu;
    if (__builtin_popcount(c16) != 1) {
      const int mi = row_min_i32(ev == best ? ei : 0x7fffffff);
      c16 = (unsigned)__ballot(ev == best && ei == mi) & 0xffffu;
    }
    const int win = __builtin_ctz(c16);
    farthest = __builtin_amdgcn_readlane(ei, win);
    cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e[1]), win));
    cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e[2]), win));
    cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e[3]), win));
  }
}

// ---------------------------------------------------------------- farthest point sampling that skips what cannot change
// The same rounds, the same arithmetic per point, fewer points per round.  A point's running distance only changes when the new centre
// is nearer than its present value, and no running distance exceeds the one the new centre had when it was chosen (it was the maximum):
// a set of points whose bounding box lies farther from the centre than that needs no update at all.  The lower bound is evaluated with
// the very instruction sequence of the update -- sub, mul, add, add on the box's nearest corner -- and every one of those float
// operations is monotone in |argument|, so the bound holds in float arithmetic, not merely in real arithmetic: a skipped point would
// have computed d >= bound >= maximum >= its running value and stayed as it is (pointnet2.py:70-72 `mask = dist < distance`).  Samples
// stay bit-identical to the plain loop (tests/test_primitives_gpu.py; scripts/fps_blob_lane_sim.py is the host emulation lane by lane,
// scripts/fps_blob_sim.py counts the work: 21 % of the blobs per round for a uniform volume, 14 % for a surface, at 512 points per blob).
//  * prologue (once per cloud, in the same launch): the points are binned into 16 x 16 x 16 cells of the cloud's box, cells in Morton
//    order (LDS histogram, scan, scatter; the order inside a cell is whatever the atomics give -- any order yields the same samples);
//    `perm` (LDS, 16-bit) maps a sorted position back to the point index;
//  * a blob = the GS slots x 64 lanes of one (wavefront, group) = 64 GS consecutive sorted positions; consecutive blobs go to different
//    wavefronts: a round lasts as long as its busiest SIMD, and the blobs a centre touches are neighbours (measured with a wavefront's
//    blobs adjacent instead: 1.29 us per round against 1.06);
//  * per round each wavefront tests its NG blobs in lanes 0..NG-1 at once (the boxes live there) and updates the groups the ballot names;
//    the winner is found as in fps_kernel (lane maxima, one wave reduction, the winner lane's group and slot);
//  * ties (equal running distances: duplicates, lattices) must resolve to the smallest POINT index, and sorted order is not index order:
//    any tie -- two lanes, two groups or two slots at the maximum -- takes a wave-uniform slow path that looks the indices up in `perm`;
//  * the exchange between wavefronts is the one of fps_kernel, carrying sorted positions; the samples are translated through `perm` when
//    the rounds are over.
// Measured and rejected (profiles/r4_fps_blob.json holds the kept ones; DESIGN.md 4.6 has the sequence): the blob's own running maximum
// as the bound instead of the cloud's (a wave reduction per updated group: 2 % fewer updates, 1.06 us per round against 1.05; with
// 256-point blobs 1.16 against 1.07); 256-point blobs above 8,192 points (more skipped, more bookkeeping: 0.97 against 0.94); keeping a
// wavefront's previous winner when it updated nothing (1.08 against 1.06: the branch costs more than the search it saves); a binary
// branch pick of the winner's coordinates (1.03 against 0.96 for the register-indexed read).
constexpr int FPS_CELL_BITS = 4, FPS_BINS = 1 << (3 * FPS_CELL_BITS);

__device__ __forceinline__ int fps_cell(float x, float y, float z, const float (&lo)[3], const float (&inv)[3]) {
  constexpr float TOP = (float)((1 << FPS_CELL_BITS) - 1);
  const unsigned a = (unsigned)fminf(fmaxf((x - lo[0]) * inv[0], 0.f), TOP);      // NaN -> 0: every point lands in some cell
  const unsigned b = (unsigned)fminf(fmaxf((y - lo[1]) * inv[1], 0.f), TOP);
  const unsigned c = (unsigned)fminf(fmaxf((z - lo[2]) * inv[2], 0.f), TOP);
  auto spread = [](unsigned v) {       // 4 bits -> bits 0, 3, 6, 9
    v = (v | (v << 4)) & 0x0c3u;
    return (v | (v << 2)) & 0x249u;
  };
  static_assert(FPS_CELL_BITS == 4, "spread() is written for 4 bits per axis");
  return (int)(spread(a) | (spread(b) << 1) | (spread(c) << 2));
}

// sorted position of slot k of (wavefront wv, lane): blob (k / GS) * NW + wv, then slot-major inside the blob
template <int NW, int GS>
__device__ __forceinline__ int fps_slot_pos(int wv, int lane, int k) {
  const unsigned g = (unsigned)k / GS, j = (unsigned)k % GS;
  return (int)(((((g * NW + wv) * GS) + j) << 6) + lane);
}

// bits = 2 * bits + (a == b): one v_cmp + one v_addc per element (the compare's carry shifts itself in), against compare + select + add
// for "first match and how many" -- both are read off the accumulated bits afterwards, on the scalar side, for the one lane that matters
__device__ __forceinline__ void push_eq_bit(unsigned& bits, unsigned a, unsigned b) {
  asm("v_cmp_eq_u32_e32 vcc, %1, %2\n\ts_nop 1\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(a), "v"(b) : "vcc");   // gfx950: 2 wait states between a VALU write of vcc and a VALU read of it
}

// a group's coordinates as one register vector, so that a wave-uniform slot number can index it (s_set_gpr_idx_on + v_mov)
template <int GS> struct fps_group_vec;
template <> struct fps_group_vec<4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct fps_group_vec<8> { typedef float type __attribute__((ext_vector_type(8))); };

// coordinates of slot j (wave-uniform) of group g (wave-uniform; binary search over the groups)
template <int GLO, int GHI, typename V>
__device__ __forceinline__ void fps_group_pick(int g, int j, const V* PX, const V* PY, const V* PZ, float& x, float& y, float& z) {
  if constexpr (GHI - GLO == 1) {
    x = PX[GLO][j]; y = PY[GLO][j]; z = PZ[GLO][j];
    asm volatile("" : "+v"(x), "+v"(y), "+v"(z));           // keeps the leaves apart (merged, the group is copied first: 3 GS moves)
  } else {
    constexpr int MID = (GLO + GHI) / 2;
    if (g < MID) fps_group_pick<GLO, MID, V>(g, j, PX, PY, PZ, x, y, z);
    else fps_group_pick<MID, GHI, V>(g, j, PX, PY, PZ, x, y, z);
  }
}

// inside group gw of the winner lane wl (both wave-uniform; binary search over the groups): its first slot at the wave maximum, how
// many of its slots hold it, the slot's coordinates
template <int GLO, int GHI, int GS, int PPT, int NW, typename V>
__device__ __forceinline__ void fps_ball_find(int gw, int wl, unsigned wmax, const unsigned (&dist)[PPT], const V* PX, const V* PY, const V* PZ,
                                              int& pw, int& cw, float& x, float& y, float& z) {
  if constexpr (GHI - GLO == 1) {
    constexpr int G = GLO;
    unsigned bv = wmax;
    asm volatile("" : "+v"(bv));                            // the search stays inside its branch
    unsigned bits = 0u;                                     // bit GS-1-j: slot j of the group holds the maximum
#pragma unroll
    for (int j = 0; j < GS; ++j) push_eq_bit(bits, dist[G * GS + j], bv);
    const unsigned bw = (unsigned)__builtin_amdgcn_readlane((int)bits, wl);      // != 0: the winner lane holds the maximum in this group
    const int jw = (GS - 1) - (31 - __builtin_clz(bw));
    pw = (G * NW * GS + jw) << 6;                           // fps_slot_pos of slot G * GS + jw, without the wave's and the lane's share
    cw = __builtin_popcount(bw);
    x = PX[G][jw]; y = PY[G][jw]; z = PZ[G][jw];
    asm volatile("" : "+v"(x), "+v"(y), "+v"(z));
  } else {
    constexpr int MID = (GLO + GHI) / 2;
    if (gw < MID) fps_ball_find<GLO, MID, GS, PPT, NW, V>(gw, wl, wmax, dist, PX, PY, PZ, pw, cw, x, y, z);
    else fps_ball_find<MID, GHI, GS, PPT, NW, V>(gw, wl, wmax, dist, PX, PY, PZ, pw, cw, x, y, z);
  }
}

template <int NT, int PPT, int GS>
__global__ __launch_bounds__(NT) void fps_blob_kernel(const float* __restrict__ xyz, const long long* __restrict__ start, int N, int npoint,
                                                      long long* __restrict__ out, float* __restrict__ out_xyz) {
  static_assert(NT % 64 == 0 && NT <= 1024 && GS % 2 == 0 && PPT % GS == 0 && (GS & (GS - 1)) == 0, "geometry");
  static_assert(FPS_BINS % NT == 0 && NT * PPT < 0xffff, "one scan chunk per thread; 16-bit point indices with 0xffff = none");
  constexpr int H = PPT / 2, NG = PPT / GS, NW = NT / 64, CAP = NT * PPT, BPT = FPS_BINS / NT;   // H: slot pairs per thread
  static_assert(NG <= 64, "one lane per group for the skip test");
  __shared__ __attribute__((aligned(16))) unsigned short perm[CAP];   // sorted position -> point index (0xffff: padding)
  __shared__ __attribute__((aligned(16))) int scr[FPS_BINS];   // prologue: box partials, then the cell histogram; rounds: the wave records
  int* wsum = reinterpret_cast<int*>(perm);                // the scan's wave totals (perm is written after the scan); 64 KB of LDS in all at 48 slots
  f32x4 (*red_v)[16] = reinterpret_cast<f32x4 (*)[16]>(scr);         // [2][16]: (bits of the best distance, x, y, z) of each wave's winner
  int (*red_i)[16] = reinterpret_cast<int (*)[16]>(scr + 128);       // [2][16]: its sorted position
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* xb = xyz + (size_t)b * N * 3;
  long long* ob = out + (size_t)b * npoint;
  float* oxb = out_xyz ? out_xyz + (size_t)b * npoint * 3 : nullptr;

  // ---- prologue 1: the cloud's box (the thread's share of the cloud in index order, CH points in flight at a time) ----
  constexpr int CH = PPT / 2;
  float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
  for (int k0 = 0; k0 < PPT; k0 += CH) {
    float r[CH][3];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int p = tid + (k0 + j) * NT, q = p < N ? p : N - 1;      // beyond N: the last point again, the box does not mind
#pragma unroll
      for (int a = 0; a < 3; ++a) r[j][a] = xb[q * 3 + a];
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
#pragma unroll
      for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], r[j][a]); hi[a] = fmaxf(hi[a], r[j][a]); }
    }
    asm volatile("" ::: "memory");
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o)); }
  }
  float* fscr = reinterpret_cast<float*>(scr);
  if (lane == 0) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { fscr[wv * 8 + a] = lo[a]; fscr[wv * 8 + 4 + a] = hi[a]; }
  }
  __syncthreads();
  float inv[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    for (int w = 0; w < NW; ++w) { lo[a] = fminf(lo[a], fscr[w * 8 + a]); hi[a] = fmaxf(hi[a], fscr[w * 8 + 4 + a]); }
    inv[a] = (float)(1 << FPS_CELL_BITS) / fmaxf(hi[a] - lo[a], 1e-30f);
  }
  __syncthreads();
  // ---- prologue 2: counting sort by cell ----
  for (int i = tid; i < FPS_BINS; i += NT) scr[i] = 0;
  __syncthreads();
  unsigned cell[H];                    // the cells of the thread's points, two per register
  int tid2 = tid;
  asm volatile("" : "+v"(tid2));       // the addresses are formed again here (kept from the first pass they would be 2 registers per point)
#pragma unroll
  for (int k0 = 0; k0 < PPT; k0 += CH) {
    float r[CH][3];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int p = tid2 + (k0 + j) * NT, q = p < N ? p : N - 1;
#pragma unroll
      for (int a = 0; a < 3; ++a) r[j][a] = xb[q * 3 + a];
    }
#pragma unroll
    for (int j = 0; j < CH; ++j) {
      const int k = k0 + j;
      const unsigned c = (unsigned)fps_cell(r[j][0], r[j][1], r[j][2], lo, inv);
      cell[k >> 1] = (k & 1) ? (cell[k >> 1] | (c << 16)) : c;
      if (tid + k * NT < N) atomicAdd(&scr[c], 1);
    }
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  {
    int c[BPT], s = 0;
#pragma unroll
    for (int i = 0; i < BPT; ++i) { c[i] = scr[tid * BPT + i]; s += c[i]; }
    int inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int base = inc - s;
    for (int w = 0; w < wv; ++w) base += wsum[w];
#pragma unroll
    for (int i = 0; i < BPT; ++i) { scr[tid * BPT + i] = base; base += c[i]; }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int p = tid + k * NT;
    if (p < N) perm[atomicAdd(&scr[(cell[k >> 1] >> (16 * (k & 1))) & 0xffffu], 1)] = (unsigned short)p;
  }
  for (int s = N + tid; s < CAP; s += NT) perm[s] = 0xffffu;
  __syncthreads();
  // ---- prologue 3: the thread's slots (again all loads in flight), the blob boxes (lane g of every wave holds group g's), the maxima ----
  typedef typename fps_group_vec<GS>::type V;
  V PX[NG], PY[NG], PZ[NG];            // the thread's points, a register vector per group and axis
  unsigned dist[PPT];                  // bit patterns of the running distances (all >= +0)
#pragma unroll
  for (int g = 0; g < NG; ++g) {       // a group at a time: GS positions out of LDS, then GS gathers in flight
    int sp[GS];
#pragma unroll
    for (int j = 0; j < GS; ++j) sp[j] = perm[fps_slot_pos<NW, GS>(wv, lane, g * GS + j)];
#pragma unroll
    for (int j = 0; j < GS; ++j) {
      const int k = g * GS + j;
      const bool real = sp[j] != 0xffff;
      const int q = real ? sp[j] : 0;
      const float x = xb[q * 3 + 0], y = xb[q * 3 + 1], z = xb[q * 3 + 2];
      // padding: at the origin with running distance 0, which never shrinks and never beats a real point (its index reads 0xffff)
      PX[g][j] = real ? x : 0.f; PY[g][j] = real ? y : 0.f; PZ[g][j] = real ? z : 0.f;
      dist[k] = real ? __float_as_uint(1e10f) : 0u;
    }
    asm volatile("" ::: "memory");     // keeps the groups apart (all at once would need every register twice)
  }
  unsigned gmax[NG];                   // per lane: the largest running distance among the GS slots of group g
  float blo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, bhi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};   // lane g < NG: blob g's box
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    float l3[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, h3[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    unsigned m = 0u;
#pragma unroll
    for (int k = g * GS; k < (g + 1) * GS; ++k) {
      if (dist[k] != 0u) {             // a real point
        l3[0] = fminf(l3[0], PX[g][k - g * GS]); h3[0] = fmaxf(h3[0], PX[g][k - g * GS]);
        l3[1] = fminf(l3[1], PY[g][k - g * GS]); h3[1] = fmaxf(h3[1], PY[g][k - g * GS]);
        l3[2] = fminf(l3[2], PZ[g][k - g * GS]); h3[2] = fmaxf(h3[2], PZ[g][k - g * GS]);
      }
      m = m > dist[k] ? m : dist[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
      for (int a = 0; a < 3; ++a) { l3[a] = fminf(l3[a], __shfl_xor(l3[a], o)); h3[a] = fmaxf(h3[a], __shfl_xor(h3[a], o)); }
    }
    gmax[g] = m;
    if (lane == g) {
#pragma unroll
      for (int a = 0; a < 3; ++a) { blo[a] = l3[a]; bhi[a] = h3[a]; }
    }
  }
  if (tid < 32) { red_v[tid >> 4][tid & 15] = f32x4{0.f, 0.f, 0.f, 0.f}; red_i[tid >> 4][tid & 15] = 0x7fffffff; }   // absent waves: distance 0, position "none"
  __syncthreads();

  const int first = (int)start[b];
  float cx = xb[first * 3 + 0], cy = xb[first * 3 + 1], cz = xb[first * 3 + 2];
  if (tid == 0 && npoint > 0) {
    ob[0] = first;
    if (oxb) { oxb[0] = cx; oxb[1] = cy; oxb[2] = cz; }
  }
  float rad = 1e10f;                   // the largest running distance of the whole cloud (= the new centre's, when it was chosen)
  for (int it = 1; it < npoint; ++it) {
    // ---- which of the wave's blobs can change: distance from the centre to the blob's box, rounded exactly like a point's ----
    const f32x2 cxy = {cx, cy};
    const f32x2 lo2 = f32x2{blo[0], blo[1]} - cxy, hi2 = cxy - f32x2{bhi[0], bhi[1]};      // x and y in one packed subtraction each
    const float qx = fmaxf(fmaxf(lo2[0], hi2[0]), 0.f), qy = fmaxf(fmaxf(lo2[1], hi2[1]), 0.f), qz = fmaxf(fmaxf(blo[2] - cz, cz - bhi[2]), 0.f);
    const float lb = (qx * qx + qy * qy) + qz * qz;
    const unsigned need = (unsigned)__builtin_amdgcn_fcmpf(lb, rad, 4 /* ordered < */) & ((1u << NG) - 1u);   // lanes >= NG hold no box
    const f32x2 c2x = {cx, cx}, c2y = {cy, cy}, c2z = {cz, cz};
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if ((need >> g) & 1u) {
        unsigned m = 0u;
#pragma unroll
        for (int h = g * (GS / 2); h < (g + 1) * (GS / 2); ++h) {
          const int j = 2 * (h - g * (GS / 2));               // the pair's first slot inside the group
          const f32x2 ax = {PX[g][j], PX[g][j + 1]}, ay = {PY[g][j], PY[g][j + 1]}, az = {PZ[g][j], PZ[g][j + 1]};
          const f32x2 dx = ax - c2x, dy = ay - c2y, dz = az - c2z;
          const f32x2 d = (dx * dx + dy * dy) + dz * dz;      // torch.sum((xyz - centroid) ** 2, -1), two points per instruction
          const unsigned d0 = __float_as_uint(d[0]), d1 = __float_as_uint(d[1]);
          const unsigned n0 = d0 < dist[2 * h] ? d0 : dist[2 * h];              // mask = dist < distance; distance[mask] = dist[mask]
          const unsigned n1 = d1 < dist[2 * h + 1] ? d1 : dist[2 * h + 1];
          dist[2 * h] = n0; dist[2 * h + 1] = n1;
          const unsigned a = m > n0 ? m : n0;
          m = a > n1 ? a : n1;
        }
        gmax[g] = m;
      }
    }
    // ---- the wave's winner: one reduction of the lanes' maxima, then the winner lane's group and slot ----
    unsigned bv = gmax[0];
#pragma unroll
    for (int g = 1; g < NG; ++g) bv = bv > gmax[g] ? bv : gmax[g];
    const unsigned wmax = wave_max_u32(bv);
    const unsigned long long cand = __ballot(bv == wmax);
    int wl = __builtin_ctzll(cand);
    unsigned gbits = 0u;                                     // per lane, bit NG-1-g: group g holds the lane's maximum
#pragma unroll
    for (int g = 0; g < NG; ++g) push_eq_bit(gbits, gmax[g], bv);
    const unsigned gbw = (unsigned)__builtin_amdgcn_readlane((int)gbits, wl);     // the winner lane's: != 0
    const int gw = (NG - 1) - (31 - __builtin_clz(gbw));     // its first group at the maximum
    int pw, cw; float bx, by, bz;
    fps_ball_find<0, NG, GS, PPT, NW, V>(gw, wl, wmax, dist, PX, PY, PZ, pw, cw, bx, by, bz);
    int iw = pw + ((wv * GS) << 6) + wl;                     // the winner's sorted position (fps_slot_pos)
    if (__builtin_popcountll(cand) + __builtin_popcount(gbw) + cw != 3) {   // more than one lane, group or slot at the maximum                                               // equal maxima somewhere in the wave: the smallest point index wins
      int bi = 0x7fffffff, bk = 0;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        if (__ballot(gmax[g] == wmax) != 0ull) {
          int oi[GS];
#pragma unroll
          for (int j = 0; j < GS; ++j) oi[j] = perm[fps_slot_pos<NW, GS>(wv, lane, g * GS + j)];
#pragma unroll
          for (int j = 0; j < GS; ++j) {
            const bool better = dist[g * GS + j] == wmax && oi[j] < bi;
            bi = better ? oi[j] : bi; bk = better ? g * GS + j : bk;
          }
        }
      }
      const int mi = wave_min_i32(bi);
      wl = __builtin_ctzll(__ballot(bi == mi));
      const int kw = __builtin_amdgcn_readlane(bk, wl);        // (signed on purpose: with an unsigned slot number LLVM turns the pick below into one
      fps_group_pick<0, NG, V>(kw / GS, kw % GS, PX, PY, PZ, bx, by, bz);   //  dynamically indexed array and puts the coordinates into scratch)
      iw = fps_slot_pos<NW, GS>(wv, wl, kw);
    }
    const int buf = it & 1;
    if (lane == wl) { red_v[buf][wv] = f32x4{__uint_as_float(wmax), bx, by, bz}; red_i[buf][wv] = iw; }
    __syncthreads();
    // ---- workgroup: the same over the wave winners; every wave redoes it on its own copy ----
    const f32x4 e = red_v[buf][lane & 15];
    const int ei = red_i[buf][lane & 15];
    const unsigned ev = __float_as_uint(e[0]);
    constexpr unsigned RECS = NW <= 8 ? 0xffu : 0xffffu;     // lanes 0..NW-1 (rounded up) of the wave look at the records
    const unsigned best = NW <= 8 ? half_row_max_u32(ev) : row_max_u32(ev);
    unsigned c16 = (unsigned)__ballot(ev == best) & RECS;
    if (__builtin_popcount(c16) != 1) {
      const int oi = (unsigned)ei < (unsigned)CAP ? (int)perm[ei] : 0x7fffffff;
      const int mi = row_min_i32(ev == best ? oi : 0x7fffffff);
      c16 = (unsigned)__ballot(ev == best && oi == mi) & RECS;
    }
    const int win = __builtin_ctz(c16);
    const int farthest = __builtin_amdgcn_readlane(ei, win);
    cx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e[1]), win));
    cy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e[2]), win));
    cz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e[3]), win));
    rad = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)ev, win));     // wave-uniform (`best` is only valid in the lanes that look at records)
    if (tid == 0) {
      ob[it] = farthest;                                     // a sorted position for now
      if (oxb) { oxb[it * 3 + 0] = cx; oxb[it * 3 + 1] = cy; oxb[it * 3 + 2] = cz; }   // = index_points(xyz, out), for free
    }
  }
  __syncthreads();                                           // thread 0's stores are visible to the workgroup
  for (int i = 1 + tid; i < npoint; i += NT) ob[i] = perm[(int)ob[i]];
}

// generic fallback for clouds larger than the register path: running distances live in a global scratch row.
__global__ __launch_bounds__(1024) void fps_kernel_global(const float* __restrict__ xyz, const long long* __restrict__ start, int N, int npoint,
                                                          float* __restrict__ dist_scratch, long long* __restrict__ out, float* __restrict__ out_xyz) {
  __shared__ float red_v[2][16];
  __shared__ int red_i[2][16];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const float* xb = xyz + (size_t)b * N * 3;
  float* db = dist_scratch + (size_t)b * N;
  for (int p = tid; p < N; p += 1024) db[p] = 1e10f;
  int farthest = (int)start[b];
  for (int it = 0; it < npoint; ++it) {
    const float cx = xb[farthest * 3 + 0], cy = xb[farthest * 3 + 1], cz = xb[farthest * 3 + 2];
    if (tid == 0) {
      out[(size_t)b * npoint + it] = farthest;
      if (out_xyz) { float* o = out_xyz + ((size_t)b * npoint + it) * 3; o[0] = cx; o[1] = cy; o[2] = cz; }
    }
    float bv = -2.0f; int bi = 0x7fffffff;
    for (int p = tid; p < N; p += 1024) {
      const float dx = xb[p * 3 + 0] - cx, dy = xb[p * 3 + 1] - cy, dz = xb[p * 3 + 2] - cz;
      const float d = (dx * dx + dy * dy) + dz * dz;
      float cur = db[p];
      if (d < cur) { cur = d; db[p] = d; }
      if (cur > bv) { bv = cur; bi = p; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    const int buf = it & 1;
    if (lane == 0) { red_v[buf][wv] = bv; red_i[buf][wv] = bi; }
    __syncthreads();
    bv = red_v[buf][0]; bi = red_i[buf][0];
#pragma unroll
    for (int k = 1; k < 16; ++k) {
      const float ov = red_v[buf][k]; const int oi = red_i[buf][k];
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    farthest = bi;
  }
}

}  // namespace

// CATGRASP_AMD_FPS=plain (development switch, read per call): fps_kernel for every size, the round that updates every point -- what the
// tests and scripts/fps_blob_check.py compare fps_blob_kernel with
static bool fps_plain() {
  const char* e = getenv("CATGRASP_AMD_FPS");
  return e && e[0] == 'p';
}

static int fps_launch(const float* xyz, const long long* start, int B, int N, int npoint, float* dist_scratch, long long* out, float* out_xyz,
                      void* stream) {
  if (B < 0 || N <= 0 || npoint < 0) return CG_ERR_ARG;
  if ((long)B * npoint == 0) return CG_OK;
  if (!xyz || !start || !out) return CG_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)B), block(1024);
  if (N > 512 * 48) {
    if (!dist_scratch) return CG_ERR_ARG;   // (B,N) floats needed for clouds beyond the register paths
    hipLaunchKernelGGL(fps_kernel_global, grid, block, 0, s, xyz, start, N, npoint, dist_scratch, out, out_xyz);
  }
  // <= 512 points: one wavefront per cloud, no exchange between wavefronts (fps_wave_kernel); CATGRASP_AMD_FPS=plain keeps the
  // workgroup kernel below for comparison.  us per round incl. the launch (scripts/fps_time.py, one / eight clouds): 512 -> 128:
  // 0.81 / 0.72 against 0.94 / 0.84; 256 -> 64: 0.77 against 1.19; at 1,024 points (16 slots per lane) the one wavefront LOSES,
  // 0.70 against 0.64 -- a round is then 16 dependent distance updates long and nothing hides them -- so it stops at 512.
  else if (N <= 512 && !fps_plain()) {
    const size_t lds = (size_t)N * 3 * sizeof(float);
    if (N <= 256) hipLaunchKernelGGL((fps_wave_kernel<4>), grid, dim3(64), lds, s, xyz, start, N, npoint, out, out_xyz);
    else hipLaunchKernelGGL((fps_wave_kernel<8>), grid, dim3(64), lds, s, xyz, start, N, npoint, out, out_xyz);
  }
  // <= 2,048 points: 512 threads x 4 points, every point every round (0.54 us per round at N = 2,048; the blob-skipping kernel with one
  // 256-point blob per wavefront: 0.56, and its prologue is not amortised over few rounds -- 0.76 against 0.56 at 1,024 -> 512)
  else if (N <= 512 * 4) hipLaunchKernelGGL((fps_kernel<512, 4>), grid, dim3(512), 0, s, xyz, start, N, npoint, out, out_xyz);
  else if (fps_plain()) {
    if (N <= 1024 * 8) hipLaunchKernelGGL((fps_kernel<1024, 8>), grid, block, 0, s, xyz, start, N, npoint, out, out_xyz);
    else if (N <= 512 * 40) hipLaunchKernelGGL((fps_kernel<512, 40>), grid, dim3(512), 0, s, xyz, start, N, npoint, out, out_xyz);
    else hipLaunchKernelGGL((fps_kernel<512, 48>), grid, dim3(512), 0, s, xyz, start, N, npoint, out, out_xyz);
  }
  // 2,049 .. 24,576 points: 512 threads (two waves per SIMD) x 8 .. 48 points, skipping the blobs a round cannot change; 256-point
  // blobs up to 8,192 points, 512-point blobs above.  us per round, uniform volume / surface cloud, against fps_kernel (which updates
  // every point every round; profiles/r4_fps_blob.json): 8,192 points 0.73 / 0.68 against 0.96; 12,288: 0.82 / 0.80 against 1.43;
  // 20,000: 0.91 / 0.86 against 1.44; 24,576: 0.97 / 0.91 against 1.62.
#define CG_FPS_BLOB(PPT, GS) hipLaunchKernelGGL((fps_blob_kernel<512, PPT, GS>), grid, dim3(512), 0, s, xyz, start, N, npoint, out, out_xyz)
  else if (N <= 512 * 8) CG_FPS_BLOB(8, 4);
  else if (N <= 512 * 16) CG_FPS_BLOB(16, 4);
  else if (N <= 512 * 24) CG_FPS_BLOB(24, 8);
  else if (N <= 512 * 32) CG_FPS_BLOB(32, 8);
  else if (N <= 512 * 40) CG_FPS_BLOB(40, 8);
  else CG_FPS_BLOB(48, 8);
#undef CG_FPS_BLOB
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_farthest_point_sample(const float* xyz, const long long* start, int B, int N, int npoint, float* dist_scratch,
                                        long long* out, void* stream) {
  return fps_launch(xyz, start, B, N, npoint, dist_scratch, out, nullptr, stream);
}

extern "C" int cg_farthest_point_sample_xyz(const float* xyz, const long long* start, int B, int N, int npoint, float* dist_scratch,
                                            long long* out, float* out_xyz, void* stream) {
  if ((long)B * npoint > 0 && !out_xyz) return CG_ERR_ARG;
  return fps_launch(xyz, start, B, N, npoint, dist_scratch, out, out_xyz, stream);
}
