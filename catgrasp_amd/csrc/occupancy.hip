// my_cpp.makeOccupancyGridFromCloudScan (my_cpp/common.cpp:324-431) on the device.
// The reference builds an octomap from the scan, then for every point of a regular lattice around the cloud casts a
// ray from the sensor origin through it (OccupancyOcTreeBase::castRay, ignoreUnknownCells=true) and keeps the lattice
// point iff the ray hits an occupied leaf whose centre is not farther than the point: "every voxel at or behind the
// observed surface".  With unknown cells ignored only the OCCUPIED leaf set matters, held here as a dense bitmap over
// the key bounding box.  One thread per lattice point runs the same Amanatides-Woo traversal in double precision
// (identical operation order to oracle/collision_ref.c, so the selected set is bit-identical).
#include "cg_common.hpp"
#include "../../include/catgrasp_amd.h"

namespace {

struct Bitmap { const unsigned int* bits; int x0, y0, z0, dx, dy, dz; };   // key offsets are key-32768 of the box corner

__device__ __forceinline__ bool occupied(const Bitmap& b, int kx, int ky, int kz) {   // k* = key - 32768
  const int ix = kx - b.x0, iy = ky - b.y0, iz = kz - b.z0;
  if ((unsigned)ix >= (unsigned)b.dx || (unsigned)iy >= (unsigned)b.dy || (unsigned)iz >= (unsigned)b.dz) return false;
  const size_t i = ((size_t)ix * b.dy + iy) * b.dz + iz;
  return (b.bits[i >> 5] >> (i & 31)) & 1u;
}

__global__ void set_bits_kernel(const short* __restrict__ keys4, long n, unsigned int* __restrict__ bits, int x0, int y0, int z0,
                                int dy, int dz) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const short4 k = ((const short4*)keys4)[i];
  const size_t b = ((size_t)(k.x - x0) * dy + (k.y - y0)) * dz + (k.z - z0);
  atomicOr(bits + (b >> 5), 1u << (b & 31));
}

__global__ __launch_bounds__(256) void occupancy_rays_kernel(Bitmap bm, float ox, float oy, float oz, float resolution, int nx, int ny,
                                                             int nz, double max_range, float* __restrict__ lattice,
                                                             unsigned char* __restrict__ keep) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)nx * ny * nz;
  if (i >= total) return;
  const int zi = (int)(i % nz), yi = (int)((i / nz) % ny), xi = (int)(i / ((long)nz * ny));
  const float x = ox + xi * resolution, y = oy + yi * resolution, z = oz + zi * resolution;
  lattice[i * 3 + 0] = x; lattice[i * 3 + 1] = y; lattice[i * 3 + 2] = z;
  float dir[3] = {x, y, z};
  const float sq = (x * x + y * y) + z * z;
  if (sq > 0.0f) { const float nrm = sqrtf(sq); dir[0] = x / nrm; dir[1] = y / nrm; dir[2] = z / nrm; }
  const double res = (double)resolution;
  int key[3] = {0, 0, 0};                                   // key - 32768 of the origin leaf
  float end[3];
  bool hit = false, done = false;
  if (occupied(bm, 0, 0, 0)) {
    for (int a = 0; a < 3; ++a) end[a] = (float)(((double)key[a] + 0.5) * res);
    hit = true; done = true;
  }
  float d[3];
  {
    const double len = sqrt((double)(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]));
    for (int a = 0; a < 3; ++a) d[a] = (len > 0) ? dir[a] / (float)len : dir[a];
  }
  int step[3]; double tmax[3], tdelta[3];
  for (int a = 0; a < 3; ++a) {
    step[a] = d[a] > 0.0f ? 1 : (d[a] < 0.0f ? -1 : 0);
    if (step[a] != 0) {
      double border = ((double)key[a] + 0.5) * res;
      border += (double)(step[a] * res * 0.5);
      tmax[a] = (border - 0.0) / (double)d[a];
      tdelta[a] = res / fabs((double)d[a]);
    } else { tmax[a] = 1.7976931348623157e308; tdelta[a] = 1.7976931348623157e308; }
  }
  if (step[0] == 0 && step[1] == 0 && step[2] == 0) done = true;
  const double max_range_sq = max_range * max_range;
  while (!done) {
    int dim;
    if (tmax[0] < tmax[1]) dim = (tmax[0] < tmax[2]) ? 0 : 2; else dim = (tmax[1] < tmax[2]) ? 1 : 2;
    // static indexing (no scratch): update the chosen axis
    int kcur = dim == 0 ? key[0] : (dim == 1 ? key[1] : key[2]);
    const int st = dim == 0 ? step[0] : (dim == 1 ? step[1] : step[2]);
    if ((st < 0 && kcur == -32768) || (st > 0 && kcur == 32767)) break;
    kcur += st;
    if (dim == 0) { key[0] = kcur; tmax[0] += tdelta[0]; }
    else if (dim == 1) { key[1] = kcur; tmax[1] += tdelta[1]; }
    else { key[2] = kcur; tmax[2] += tdelta[2]; }
    for (int a = 0; a < 3; ++a) end[a] = (float)(((double)key[a] + 0.5) * res);
    if (max_range > 0.0) {
      double dsq = 0.0;
      for (int a = 0; a < 3; ++a) dsq += ((double)end[a] - 0.0) * ((double)end[a] - 0.0);
      if (dsq > max_range_sq) break;
    }
    if (occupied(bm, key[0], key[1], key[2])) { hit = true; break; }
  }
  bool k = false;
  if (hit) {
    const float dist_query = sqrtf(x * x + y * y + z * z);
    const float dist = (float)sqrt((double)(end[0] * end[0] + end[1] * end[1] + end[2] * end[2]));
    k = dist <= dist_query;
  }
  keep[i] = k ? 1 : 0;
}

}  // namespace

extern "C" int cg_occupancy_set_bits(const short* keys4, long n_keys, unsigned int* bits, int x0, int y0, int z0, int dx, int dy,
                                     int dz, void* stream) {
  if (n_keys < 0 || dx <= 0 || dy <= 0 || dz <= 0) return CG_ERR_ARG;
  if (n_keys == 0) return CG_OK;
  if (!keys4 || !bits) return CG_ERR_ARG;
  hipLaunchKernelGGL(set_bits_kernel, dim3((unsigned)((n_keys + 255) / 256)), dim3(256), 0, (hipStream_t)stream, keys4, n_keys, bits,
                     x0, y0, z0, dy, dz);
  return cg_hip_status(hipGetLastError());
}

extern "C" int cg_occupancy_grid_rays(const unsigned int* bits, int x0, int y0, int z0, int dx, int dy, int dz, float origin_x,
                                      float origin_y, float origin_z, float resolution, int nx, int ny, int nz, double max_range,
                                      float* lattice, unsigned char* keep, void* stream) {
  if (dx <= 0 || dy <= 0 || dz <= 0 || nx < 0 || ny < 0 || nz < 0 || !(resolution > 0.f)) return CG_ERR_ARG;
  const long total = (long)nx * ny * nz;
  if (total == 0) return CG_OK;
  if (!bits || !lattice || !keep) return CG_ERR_ARG;
  Bitmap bm{bits, x0, y0, z0, dx, dy, dz};
  hipLaunchKernelGGL(occupancy_rays_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bm, origin_x,
                     origin_y, origin_z, resolution, nx, ny, nz, max_range, lattice, keep);
  return cg_hip_status(hipGetLastError());
}
