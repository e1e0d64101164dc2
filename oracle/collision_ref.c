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
 *     posed mesh triangle"  (surface-only, as BVH-vs-octree is).  The oracle decides it by CLIPPING the
 *     triangle against the box's six half-spaces in float64 (cr_tri_box_clip64: no separating axes) -- a
 *     different decision procedure from the HIP kernel's 13-axis float32 separating-axis test, so that
 *     kernel == oracle is not one formulation compared with itself.  The float32 SAT twin of the kernel
 *     (cr_tri_box_overlap) and libccd's MPR as FCL's default solver runs it (cr_tri_box_mpr) are selectable
 *     through cr_set_variant for the sensitivity study (oracle/collision_sensitivity.py).
 * Bit-exactness of the HIP path's codes / nudges / poses is defined against THIS file's default predicate.
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

/* ---------------- formulations of the predicate (sensitivity study, profiles/r4_collision_sensitivity.json) ----------------------
 * FCL and octomap are absent, so what CAN be measured is how much the answer depends on the details this restatement had to choose:
 *   leaf_mode 1  the leaf box FCL actually hands to its narrow phase: not ((float)k + 0.5f) * res, but the box reached by 16
 *                float halvings of the root BV [-d, d]^3, d = (float)((1 << 16) * resolution / 2) (fcl::OcTree<float>::getRootBV,
 *                computeChildBV: child.min/max = (bv.min + bv.max) * 0.5 per set / unset bit of the child index), then
 *                constructBox(): side = max - min, centre = (min + max) * 0.5 -- a different rounding path, ~1 ulp of 0.6 m;
 *   dh           the cube's half edge grown / shrunk by dh metres (+-1e-6 m: the scale of libccd's tolerances at contact);
 *   narrow 0     (DEFAULT, the parity oracle) closed-set intersection by Sutherland-Hodgman clipping of the triangle against the six
 *                half-spaces in float64 -- no separating axes at all (the C twin of oracle/tribox_exact.py);
 *   narrow 1     the 13-axis separating-axis test in float32, operation for operation what csrc/collision.hip evaluates;
 *   narrow 2     libccd's MPR (Minkowski portal refinement) intersection test on {box, triangle} support functions, as FCL's default
 *                GJKSolver_libccd runs shapeTriangleIntersect without contact output: ccdMPRIntersect, mpr_tolerance 1e-6 (FCL's
 *                collision_tolerance), double precision -- restated from the published algorithm (libccd src/mpr.c, FCL
 *                narrowphase/detail/convexity_based_algorithm/gjk_libccd-inl.h: supportBox / supportTriangle / centerShape /
 *                centerTriangle); the sources are not in this container, so this is a restatement from general knowledge, not a pin.
 * cr_set_variant(0, 0, 0) is the parity oracle. */
static struct { int leaf_mode; float dh; int narrow; } g_variant = {0, 0.0f, 0};
void cr_set_variant(int leaf_mode, float dh, int narrow) { g_variant.leaf_mode = leaf_mode; g_variant.dh = dh; g_variant.narrow = narrow; }

/* FCL's leaf box of key k (0 .. 65535 per axis) by recursive float halving; out: centre c[3], half extents h[3] */
static void fcl_leaf_box(const int key_minus[3], float resolution, float c[3], float h[3]) {
  const float d = (float)((double)(1 << 16) * (double)resolution / 2.0);
  for (int a = 0; a < 3; ++a) {
    float lo = -d, hi = d;
    const int k = key_minus[a] + TREE_MAX_VAL;
    for (int lvl = 15; lvl >= 0; --lvl) {
      const float mid = (lo + hi) * 0.5f;
      if ((k >> lvl) & 1) lo = mid; else hi = mid;
    }
    const float side = hi - lo;                 /* constructBox: Box(bv.max_ - bv.min_), tf.translation() = bv.center() */
    c[a] = (lo + hi) * 0.5f;
    h[a] = side * 0.5f;                         /* the narrow phase sees the box through its half sides */
  }
}

/* 13-axis SAT with per-axis half extents (identical to cr_tri_box_overlap when h[0] == h[1] == h[2]) */
int cr_tri_box_overlap_h3(const float c[3], const float h[3], const float a[3], const float b[3], const float d[3]) {
  float v0[3], v1[3], v2[3], e[3][3];
  for (int i = 0; i < 3; ++i) { v0[i] = a[i] - c[i]; v1[i] = b[i] - c[i]; v2[i] = d[i] - c[i]; }
  for (int i = 0; i < 3; ++i) { e[0][i] = v1[i] - v0[i]; e[1][i] = v2[i] - v1[i]; e[2][i] = v0[i] - v2[i]; }
  const float* vs[3] = {v0, v1, v2};
  for (int k = 0; k < 3; ++k) {
    const float fex = fabsf(e[k][0]), fey = fabsf(e[k][1]), fez = fabsf(e[k][2]);
    const float* p = vs[k == 1 ? 0 : (k == 2 ? 0 : 0)]; (void)p;
    /* the two distinct projections per axis are those of a vertex ON the edge and of the vertex OPPOSITE to it */
    const float* on = vs[k]; const float* opp = vs[(k + 2) % 3];
    { float pa = e[k][2] * on[1] - e[k][1] * on[2], pb = e[k][2] * opp[1] - e[k][1] * opp[2]; AXIS(pa, pb, fez * h[1] + fey * h[2]); }
    { float pa = -e[k][2] * on[0] + e[k][0] * on[2], pb = -e[k][2] * opp[0] + e[k][0] * opp[2]; AXIS(pa, pb, fez * h[0] + fex * h[2]); }
    { float pa = e[k][1] * on[0] - e[k][0] * on[1], pb = e[k][1] * opp[0] - e[k][0] * opp[1]; AXIS(pa, pb, fey * h[0] + fex * h[1]); }
  }
  for (int i = 0; i < 3; ++i)
    if (min3f(v0[i], v1[i], v2[i]) > h[i] || max3f(v0[i], v1[i], v2[i]) < -h[i]) return 0;
  float n[3];
  n[0] = e[0][1] * e[1][2] - e[0][2] * e[1][1];
  n[1] = e[0][2] * e[1][0] - e[0][0] * e[1][2];
  n[2] = e[0][0] * e[1][1] - e[0][1] * e[1][0];
  float vmin[3], vmax[3];
  for (int q = 0; q < 3; ++q) {
    if (n[q] > 0.0f) { vmin[q] = -h[q] - v0[q]; vmax[q] = h[q] - v0[q]; }
    else { vmin[q] = h[q] - v0[q]; vmax[q] = -h[q] - v0[q]; }
  }
  if ((n[0] * vmin[0] + n[1] * vmin[1]) + n[2] * vmin[2] > 0.0f) return 0;
  return (n[0] * vmax[0] + n[1] * vmax[1]) + n[2] * vmax[2] >= 0.0f;
}

/* closed triangle vs closed box by polygon clipping in float64 (no separating axes): 1 iff the clipped polygon is non-empty */
int cr_tri_box_clip64(const float c[3], const float h[3], const float a[3], const float b[3], const float d[3]) {
  double poly[16][3], tmp[16][3];
  int n = 3;
  for (int k = 0; k < 3; ++k) { poly[0][k] = (double)a[k] - (double)c[k]; poly[1][k] = (double)b[k] - (double)c[k]; poly[2][k] = (double)d[k] - (double)c[k]; }
  for (int axis = 0; axis < 3; ++axis)
    for (int sgn = 1; sgn >= -1; sgn -= 2) {
      int m = 0;
      for (int i = 0; i < n; ++i) {
        const double* p = poly[i]; const double* q = poly[(i + 1) % n];
        const double sp = (double)h[axis] - sgn * p[axis], sq = (double)h[axis] - sgn * q[axis];      /* >= 0 inside */
        if (sp >= 0) { memcpy(tmp[m++], p, sizeof(double) * 3); }
        if ((sp >= 0) != (sq >= 0)) {
          const double t = sp / (sp - sq);
          for (int k = 0; k < 3; ++k) tmp[m][k] = p[k] + t * (q[k] - p[k]);
          ++m;
        }
      }
      n = m;
      if (n == 0) return 0;
      memcpy(poly, tmp, sizeof(double) * 3 * (size_t)n);
    }
  return 1;
}

/* ---- libccd MPR on {box centred at c with half sides h, triangle a b d}, everything in double (ccd_real_t of a CCD_DOUBLE build) ---- */
#define MPR_EPS 2.220446049250313e-16          /* CCD_EPS = DBL_EPSILON */
typedef struct { double v[3]; } mv3;
static inline int mpr_is_zero(double x) { return fabs(x) < MPR_EPS; }
static inline int mpr_eq(double a, double b) {
  const double ab = fabs(a - b);
  if (ab < MPR_EPS) return 1;
  const double fa = fabs(a), fb = fabs(b);
  return fb > fa ? ab < MPR_EPS * fb : ab < MPR_EPS * fa;
}
static inline double mpr_dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void mpr_cross(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static inline void mpr_normalize(double* a) { const double n = sqrt(mpr_dot(a, a)); a[0] /= n; a[1] /= n; a[2] /= n; }
typedef struct { double c[3], h[3], tri[3][3], tc[3]; } mpr_pair;
/* Minkowski-difference support point box - triangle in direction dir (__ccdSupport: v1 = support1(dir), v2 = support2(-dir), v = v1 - v2) */
static void mpr_support(const mpr_pair* p, const double* dir, double* out) {
  double s1[3], s2[3];
  for (int k = 0; k < 3; ++k) {                       /* supportBox: ccdSign(dir) * half side, + centre */
    const double sg = mpr_is_zero(dir[k]) ? 0.0 : (dir[k] < 0.0 ? -1.0 : 1.0);
    s1[k] = sg * p->h[k] + p->c[k];
  }
  double best = -1.7976931348623157e308; int bi = 0;  /* supportTriangle: the vertex with the largest dot of (vertex - centroid) */
  for (int i = 0; i < 3; ++i) {
    double q[3] = {p->tri[i][0] - p->tc[0], p->tri[i][1] - p->tc[1], p->tri[i][2] - p->tc[2]};
    const double d = -(dir[0] * q[0] + dir[1] * q[1] + dir[2] * q[2]);
    if (d > best) { best = d; bi = i; }
  }
  for (int k = 0; k < 3; ++k) { s2[k] = p->tri[bi][k]; out[k] = s1[k] - s2[k]; }
}
/* -> 1 intersect, 0 not (ccdMPRIntersect: discoverPortal + refinePortal) */
int cr_tri_box_mpr(const float c[3], const float h[3], const float a[3], const float b[3], const float d[3]) {
  mpr_pair p;
  for (int k = 0; k < 3; ++k) {
    p.c[k] = c[k]; p.h[k] = h[k]; p.tri[0][k] = a[k]; p.tri[1][k] = b[k]; p.tri[2][k] = d[k];
    p.tc[k] = (p.tri[0][k] + p.tri[1][k] + p.tri[2][k]) / 3.0;
  }
  const double tol = 1e-6;                             /* FCL GJKSolver_libccd: collision_tolerance -> ccd.mpr_tolerance */
  double v0[3], v1[3], v2[3], v3[3], v4[3], dir[3], va[3], vb[3], dot;
  for (int k = 0; k < 3; ++k) v0[k] = p.c[k] - p.tc[k];                                  /* findOrigin: centre1 - centre2 */
  if (mpr_eq(v0[0], 0.0) && mpr_eq(v0[1], 0.0) && mpr_eq(v0[2], 0.0)) v0[0] += MPR_EPS * 10.0;
  for (int k = 0; k < 3; ++k) dir[k] = -v0[k];
  mpr_normalize(dir);
  mpr_support(&p, dir, v1);
  dot = mpr_dot(v1, dir);
  if (mpr_is_zero(dot) || dot < 0.0) return 0;
  mpr_cross(v0, v1, dir);
  if (mpr_is_zero(mpr_dot(dir, dir))) return 1;        /* origin on v1, or on the segment v0-v1: both count as intersection */
  mpr_normalize(dir);
  mpr_support(&p, dir, v2);
  dot = mpr_dot(v2, dir);
  if (mpr_is_zero(dot) || dot < 0.0) return 0;
  for (int k = 0; k < 3; ++k) { va[k] = v1[k] - v0[k]; vb[k] = v2[k] - v0[k]; }
  mpr_cross(va, vb, dir); mpr_normalize(dir);
  if (mpr_dot(dir, v0) > 0.0) {                        /* portal faces oriented "outside" the origin */
    for (int k = 0; k < 3; ++k) { const double t = v1[k]; v1[k] = v2[k]; v2[k] = t; dir[k] = -dir[k]; }
  }
  for (int it = 0;; ++it) {                            /* discoverPortal: until (v1, v2, v3) encloses the ray v0 -> origin */
    if (it > 500) return 0;                            /* (libccd loops unbounded here; FCL's max_collision_iterations as a guard) */
    mpr_support(&p, dir, v3);
    dot = mpr_dot(v3, dir);
    if (mpr_is_zero(dot) || dot < 0.0) return 0;
    int cont = 0;
    mpr_cross(v1, v3, va); dot = mpr_dot(va, v0);
    if (dot < 0.0 && !mpr_is_zero(dot)) { memcpy(v2, v3, sizeof v2); cont = 1; }
    if (!cont) {
      mpr_cross(v3, v2, va); dot = mpr_dot(va, v0);
      if (dot < 0.0 && !mpr_is_zero(dot)) { memcpy(v1, v3, sizeof v1); cont = 1; }
    }
    if (!cont) break;
    for (int k = 0; k < 3; ++k) { va[k] = v1[k] - v0[k]; vb[k] = v2[k] - v0[k]; }
    mpr_cross(va, vb, dir); mpr_normalize(dir);
  }
  for (int it = 0;; ++it) {                            /* refinePortal */
    if (it > 500) return 0;
    for (int k = 0; k < 3; ++k) { va[k] = v2[k] - v1[k]; vb[k] = v3[k] - v1[k]; }
    mpr_cross(va, vb, dir); mpr_normalize(dir);        /* portalDir */
    dot = mpr_dot(dir, v1);
    if (mpr_is_zero(dot) || dot > 0.0) return 1;       /* portalEncapsulesOrigin */
    mpr_support(&p, dir, v4);
    dot = mpr_dot(v4, dir);
    if (!(mpr_is_zero(dot) || dot > 0.0)) return 0;   /* !portalCanEncapsuleOrigin */
    {                                                  /* portalReachTolerance */
      const double dv4 = dot;
      double m = dv4 - mpr_dot(v1, dir);
      const double m2 = dv4 - mpr_dot(v2, dir), m3 = dv4 - mpr_dot(v3, dir);
      if (m2 < m) m = m2;
      if (m3 < m) m = m3;
      if (mpr_eq(m, tol) || m < tol) return 0;
    }
    mpr_cross(v4, v0, va);                             /* expandPortal */
    dot = mpr_dot(v1, va);
    if (dot > 0.0) {
      dot = mpr_dot(v2, va);
      if (dot > 0.0) memcpy(v1, v4, sizeof v1); else memcpy(v3, v4, sizeof v3);
    } else {
      dot = mpr_dot(v3, va);
      if (dot > 0.0) memcpy(v2, v4, sizeof v2); else memcpy(v1, v4, sizeof v1);
    }
  }
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
  /* per-triangle bounds for a conservative cull (1e-5 m slack >> any rounding in the predicates; never changes a result) */
  float* tb = (float*)malloc(sizeof(float) * 6 * (size_t)nf);
  for (int f = 0; f < nf; ++f)
    for (int a = 0; a < 3; ++a) {
      tb[f * 6 + a] = min3f(tri[f * 9 + a], tri[f * 9 + 3 + a], tri[f * 9 + 6 + a]) - 1e-5f;
      tb[f * 6 + 3 + a] = max3f(tri[f * 9 + a], tri[f * 9 + 3 + a], tri[f * 9 + 6 + a]) + 1e-5f;
    }
  int hit = 0;
  for (int i = 0; i < nk && !hit; ++i) {
    float c[3], h3[3];
    if (g_variant.leaf_mode == 1) fcl_leaf_box(keys + i * 3, resolution, c, h3);
    else for (int a = 0; a < 3; ++a) { c[a] = ((float)keys[i * 3 + a] + 0.5f) * resolution; h3[a] = h; }
    int out = 0;
    for (int a = 0; a < 3; ++a) {
      h3[a] += g_variant.dh;
      if (c[a] - h3[a] > hi[a] + 1e-5f || c[a] + h3[a] < lo[a] - 1e-5f) out = 1;
    }
    if (out) continue;
    for (int f = 0; f < nf; ++f) {
      const float* q = tb + f * 6;
      if (c[0] - h3[0] > q[3] || c[0] + h3[0] < q[0] || c[1] - h3[1] > q[4] || c[1] + h3[1] < q[1] || c[2] - h3[2] > q[5] || c[2] + h3[2] < q[2])
        continue;
      const float* t = tri + f * 9;
      const int o = g_variant.narrow == 0 ? cr_tri_box_clip64(c, h3, t, t + 3, t + 6)
                  : g_variant.narrow == 1 ? cr_tri_box_overlap_h3(c, h3, t, t + 3, t + 6)
                                          : cr_tri_box_mpr(c, h3, t, t + 3, t + 6);
      if (o) { hit = 1; break; }
    }
  }
  free(tb);
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

/* ---------------- isAnyCollision for mesh/mesh and cloud/cloud pairs (collision_manager.cpp:93-111 tests EVERY pair) ----------------
 * PARITY UNPINNED like the rest of this file.  Predicates: "some closed triangle of A meets some closed triangle of B" (FCL's
 * BVH-vs-BVH leaf test) and "some occupied leaf cube of A meets some occupied leaf cube of B" (octree-vs-octree, box-box narrow
 * phase).  Decided here WITHOUT separating axes, in float64 -- segment-through-triangle tests, polygon clipping -- so that the HIP
 * kernels' float32 separating-axis tests (csrc/collision_pairs.hip) are compared with a different procedure. */

static void d_sub(const double* a, const double* b, double* o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static void d_cross(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static double d_dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

/* 2-D closed convex polygons (np, nq <= 3 points; a segment is a 2-gon, a point a 1-gon) by their edge normals */
static int overlap2d(const double (*P)[2], int np, const double (*Q)[2], int nq) {
  for (int pass = 0; pass < 2; ++pass) {
    const double (*A)[2] = pass ? Q : P; const int na = pass ? nq : np;
    for (int i = 0; i < na; ++i) {
      const double ex = A[(i + 1) % na][0] - A[i][0], ey = A[(i + 1) % na][1] - A[i][1];
      const double nx = -ey, ny = ex;
      if (nx == 0 && ny == 0) continue;
      double pmin = INFINITY, pmax = -INFINITY, qmin = INFINITY, qmax = -INFINITY;
      for (int k = 0; k < np; ++k) { const double d = nx * P[k][0] + ny * P[k][1]; pmin = fmin(pmin, d); pmax = fmax(pmax, d); }
      for (int k = 0; k < nq; ++k) { const double d = nx * Q[k][0] + ny * Q[k][1]; qmin = fmin(qmin, d); qmax = fmax(qmax, d); }
      if (pmax < qmin || qmax < pmin) return 0;
    }
  }
  return 1;
}

/* closed segment a-b against closed triangle v[3] (normal n, not zero) */
static int seg_tri64(const double* a, const double* b, const double (*v)[3], const double* n) {
  double ra[3], rb[3];
  d_sub(a, v[0], ra); d_sub(b, v[0], rb);
  const double da = d_dot(n, ra), db = d_dot(n, rb);
  if ((da > 0 && db > 0) || (da < 0 && db < 0)) return 0;
  /* drop the coordinate along which n is largest */
  int ax = 0;
  if (fabs(n[1]) > fabs(n[ax])) ax = 1;
  if (fabs(n[2]) > fabs(n[ax])) ax = 2;
  const int u = (ax + 1) % 3, w = (ax + 2) % 3;
  double T[3][2] = {{v[0][u], v[0][w]}, {v[1][u], v[1][w]}, {v[2][u], v[2][w]}};
  if (da == 0 && db == 0) {                      /* the segment lies in the triangle's plane */
    double S[2][2] = {{a[u], a[w]}, {b[u], b[w]}};
    return overlap2d(S, 2, T, 3);
  }
  const double t = da / (da - db);
  double X[1][2] = {{a[u] + t * (b[u] - a[u]), a[w] + t * (b[w] - a[w])}};
  return overlap2d(X, 1, T, 3);
}

int cr_tri_tri_overlap64(const float* Pf, const float* Qf) {
  double P[3][3], Q[3][3];
  for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) { P[i][k] = (double)Pf[3 * i + k] - (double)Pf[k]; Q[i][k] = (double)Qf[3 * i + k] - (double)Pf[k]; }
  double e0[3], e1[3], np_[3], nq[3];
  d_sub(P[1], P[0], e0); d_sub(P[2], P[0], e1); d_cross(e0, e1, np_);
  d_sub(Q[1], Q[0], e0); d_sub(Q[2], Q[0], e1); d_cross(e0, e1, nq);
  const int pdeg = np_[0] == 0 && np_[1] == 0 && np_[2] == 0, qdeg = nq[0] == 0 && nq[1] == 0 && nq[2] == 0;
  if (pdeg || qdeg) {                           /* a degenerate triangle is its three edges (segments) against the other one */
    if (pdeg && qdeg) return 0;                 /* segment vs segment: not a case the tests form; FCL's answer is unpinned too */
    const double (*S)[3] = pdeg ? P : Q; const double (*Tt)[3] = pdeg ? Q : P; const double* n = pdeg ? nq : np_;
    for (int i = 0; i < 3; ++i) if (seg_tri64(S[i], S[(i + 1) % 3], Tt, n)) return 1;
    return 0;
  }
  for (int i = 0; i < 3; ++i) {
    if (seg_tri64(P[i], P[(i + 1) % 3], Q, nq)) return 1;
    if (seg_tri64(Q[i], Q[(i + 1) % 3], P, np_)) return 1;
  }
  return 0;
}

int cr_mesh_mesh_collide(const float* VA, const int* FA, int nfa, const float* VB, const int* FB, int nfb, const float* poseA, const float* poseB) {
  if (nfa == 0 || nfb == 0) return 0;
  float* ta = (float*)malloc(sizeof(float) * 9 * (size_t)nfa);
  float* tb = (float*)malloc(sizeof(float) * 9 * (size_t)nfb);
  for (int f = 0; f < nfa; ++f) for (int k = 0; k < 3; ++k) pose_vertex(poseA, VA + 3 * FA[f * 3 + k], ta + f * 9 + k * 3);
  for (int f = 0; f < nfb; ++f) for (int k = 0; k < 3; ++k) pose_vertex(poseB, VB + 3 * FB[f * 3 + k], tb + f * 9 + k * 3);
  int hit = 0;
#pragma omp parallel for schedule(dynamic, 8) shared(hit)
  for (int i = 0; i < nfa; ++i) {
    if (hit) continue;
    const float* p = ta + i * 9;
    float lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = min3f(p[a], p[3 + a], p[6 + a]); hi[a] = max3f(p[a], p[3 + a], p[6 + a]); }
    for (int j = 0; j < nfb && !hit; ++j) {
      const float* q = tb + j * 9;
      int out = 0;
      for (int a = 0; a < 3; ++a)
        if (lo[a] > max3f(q[a], q[3 + a], q[6 + a]) || hi[a] < min3f(q[a], q[3 + a], q[6 + a])) out = 1;
      if (out) continue;
      if (cr_tri_tri_overlap64(p, q)) {
#pragma omp atomic write
        hit = 1;
      }
    }
  }
  free(ta); free(tb);
  return hit;
}

/* cube A (centre ca, half edge ha, axis-aligned) vs cube B (centre cb, half edge hb, axes = columns of R): some face of B (two
 * triangles) survives clipping against A, or A's centre lies inside B */
int cr_box_box_overlap64(const float* ca, float ha, const float* cb, float hb, const float* R) {
  const float h3[3] = {ha, ha, ha};
  float corner[8][3];
  for (int m = 0; m < 8; ++m)
    for (int r = 0; r < 3; ++r) {
      double x = cb[r];
      for (int c = 0; c < 3; ++c) x += (double)R[3 * r + c] * ((m >> c) & 1 ? (double)hb : -(double)hb);
      corner[m][r] = (float)x;                   /* rounded once: the clipping below works on these corners */
    }
  static const int quad[6][4] = {{0, 1, 3, 2}, {4, 5, 7, 6}, {0, 1, 5, 4}, {2, 3, 7, 6}, {0, 2, 6, 4}, {1, 3, 7, 5}};
  for (int f = 0; f < 6; ++f) {
    if (cr_tri_box_clip64(ca, h3, corner[quad[f][0]], corner[quad[f][1]], corner[quad[f][2]])) return 1;
    if (cr_tri_box_clip64(ca, h3, corner[quad[f][0]], corner[quad[f][2]], corner[quad[f][3]])) return 1;
  }
  for (int c = 0; c < 3; ++c) {                  /* A's centre in B's coordinates */
    double x = 0;
    for (int r = 0; r < 3; ++r) x += (double)R[3 * r + c] * ((double)ca[r] - (double)cb[r]);
    if (fabs(x) > (double)hb) return 0;
  }
  return 1;
}

/* keys (n,3) int32 (key - 32768); rel: B's frame -> A's frame (4x4 row-major) */
int cr_voxels_voxels_collide(const int* keysA, int na, float resA, const int* keysB, int nb, float resB, const float* rel) {
  if (na == 0 || nb == 0) return 0;
  const float ha = 0.5f * resA, hb = 0.5f * resB;
  const float R[9] = {rel[0], rel[1], rel[2], rel[4], rel[5], rel[6], rel[8], rel[9], rel[10]};
  const float reach = 1.7320508f * (ha + hb) * 1.0001f;
  float* cbs = (float*)malloc(sizeof(float) * 3 * (size_t)nb);
  for (int j = 0; j < nb; ++j) {
    float cb[3];
    for (int a = 0; a < 3; ++a) cb[a] = ((float)keysB[3 * j + a] + 0.5f) * resB;
    pose_vertex(rel, cb, cbs + 3 * j);
  }
  int hit = 0;
#pragma omp parallel for schedule(dynamic, 16) shared(hit)
  for (int i = 0; i < na; ++i) {
    if (hit) continue;
    float ca[3];
    for (int a = 0; a < 3; ++a) ca[a] = ((float)keysA[3 * i + a] + 0.5f) * resA;
    for (int j = 0; j < nb && !hit; ++j) {
      const float* cb = cbs + 3 * j;
      if (fabsf(cb[0] - ca[0]) > reach || fabsf(cb[1] - ca[1]) > reach || fabsf(cb[2] - ca[2]) > reach) continue;
      if (cr_box_box_overlap64(ca, ha, cb, hb, R)) {
#pragma omp atomic write
        hit = 1;
      }
    }
  }
  free(cbs);
  return hit;
}

int cr_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---------------- makeOccupancyGridFromCloudScan (my_cpp/common.cpp:324-431) ----------------
 * PARITY UNPINNED (octomap absent).  Restated from the call site plus octomap's published behaviour:
 *  - insertPointCloud(p, origin): every scan point's leaf becomes occupied (hit log-odds +0.85 >= threshold);
 *    leaves only traversed by rays become free; a leaf that is both stays occupied.  castRay(...,
 *    ignoreUnknownCells=true, ...) skips free and unknown leaves alike, so only the occupied set matters.
 *  - OccupancyOcTreeBase::castRay: Amanatides-Woo DDA over the key lattice in double precision, first occupied
 *    leaf -> end = leaf centre (float), stops at maxRange (distance of the leaf centre from the origin) or at
 *    the tree border.
 *  - a lattice point (x,y,z) is emitted iff its viewing ray hits and |end| <= |(x,y,z)| (float).
 * Output order here is lattice order (xi, yi, zi); the reference's is thread-dependent (omp critical merge). */

typedef struct { uint64_t* slots; size_t cap; } keyset_t;

static uint64_t pack_key(int kx, int ky, int kz) { return ((uint64_t)kx << 32) | ((uint64_t)ky << 16) | (uint64_t)kz; }

static void keyset_init(keyset_t* s, size_t n) {
  size_t cap = 16; while (cap < 2 * n + 1) cap <<= 1;
  s->cap = cap; s->slots = (uint64_t*)malloc(cap * sizeof(uint64_t));
  for (size_t i = 0; i < cap; ++i) s->slots[i] = ~(uint64_t)0;
}
static void keyset_insert(keyset_t* s, uint64_t k) {
  size_t h = (size_t)(k * 0x9E3779B97F4A7C15ull) & (s->cap - 1);
  while (s->slots[h] != ~(uint64_t)0 && s->slots[h] != k) h = (h + 1) & (s->cap - 1);
  s->slots[h] = k;
}
static int keyset_has(const keyset_t* s, uint64_t k) {
  size_t h = (size_t)(k * 0x9E3779B97F4A7C15ull) & (s->cap - 1);
  while (s->slots[h] != ~(uint64_t)0) { if (s->slots[h] == k) return 1; h = (h + 1) & (s->cap - 1); }
  return 0;
}

/* castRay from the origin (0,0,0) along `dirf` (float, already normalised by the caller, normalised again here
 * like octomap does).  Returns 1 and the hit leaf centre (float) on a hit. */
/* Points where octomap implementations / readings of castRay can differ (the library is absent: PARITY UNPINNED); selectable for the
 * sensitivity study oracle/occupancy_sensitivity.py.  All 0 = the parity oracle (and csrc/occupancy.hip).
 *   tie        0: the dimension with the smallest tMax, ties broken as `<` does (x before y before z only on strict order);  1: `<=`
 *   range_last 0: the maxRange test on the new leaf's centre comes BEFORE its occupancy test;  1: after (an occupied leaf just beyond
 *                 maxRange is still reported)
 *   fcoord     0: leaf centre = (float)(((double)key - 32768 + 0.5) * res);  1: float arithmetic throughout, ((float)(key - 32768) + 0.5f) * (float)res
 *   fdir       0: direction normalised by a double square root of the float dot product (octomath::Vector3::normalized as restated);
 *                 1: float square root
 *   strict     0: lattice point kept iff dist <= dist_query;  1: iff dist < dist_query  (common.cpp:399 reads `<=`; the variant
 *                 measures how many points sit exactly on the boundary) */
static struct { int tie, range_last, fcoord, fdir, strict; } g_occ_variant = {0, 0, 0, 0, 0};
void cr_set_occupancy_variant(int tie, int range_last, int fcoord, int fdir, int strict) {
  g_occ_variant.tie = tie; g_occ_variant.range_last = range_last; g_occ_variant.fcoord = fcoord; g_occ_variant.fdir = fdir; g_occ_variant.strict = strict;
}
static float occ_leaf_coord(int key, double resolution) {
  if (g_occ_variant.fcoord) return ((float)(key - TREE_MAX_VAL) + 0.5f) * (float)resolution;
  return (float)(((double)key - TREE_MAX_VAL + 0.5) * resolution);
}

static int cast_ray_origin(const keyset_t* occ, const float dirf[3], double resolution, double max_range, float end[3]) {
  int key[3] = {TREE_MAX_VAL, TREE_MAX_VAL, TREE_MAX_VAL};                 /* coordToKey(0) = floor(0) + 32768 */
  if (keyset_has(occ, pack_key(key[0], key[1], key[2]))) {
    for (int i = 0; i < 3; ++i) end[i] = occ_leaf_coord(key[i], resolution);
    return 1;
  }
  /* octomath::Vector3::normalized(): len = sqrt(x*x+y*y+z*z) (float products, double sqrt), components / (float)len */
  float d[3];
  {
    const float dot = dirf[0] * dirf[0] + dirf[1] * dirf[1] + dirf[2] * dirf[2];
    const double len = g_occ_variant.fdir ? (double)sqrtf(dot) : sqrt((double)dot);
    for (int i = 0; i < 3; ++i) d[i] = (len > 0) ? dirf[i] / (float)len : dirf[i];
  }
  int step[3]; double tmax[3], tdelta[3];
  for (int i = 0; i < 3; ++i) {
    step[i] = d[i] > 0.0f ? 1 : (d[i] < 0.0f ? -1 : 0);
    if (step[i] != 0) {
      double border = ((double)key[i] - TREE_MAX_VAL + 0.5) * resolution;
      border += (double)(step[i] * resolution * 0.5);
      tmax[i] = (border - 0.0) / (double)d[i];
      tdelta[i] = resolution / fabs((double)d[i]);
    } else { tmax[i] = 1.7976931348623157e308; tdelta[i] = 1.7976931348623157e308; }
  }
  if (step[0] == 0 && step[1] == 0 && step[2] == 0) return 0;
  const double max_range_sq = max_range * max_range;
  for (;;) {
    int dim;
    if (g_occ_variant.tie) { if (tmax[0] <= tmax[1]) dim = (tmax[0] <= tmax[2]) ? 0 : 2; else dim = (tmax[1] <= tmax[2]) ? 1 : 2; }
    else { if (tmax[0] < tmax[1]) dim = (tmax[0] < tmax[2]) ? 0 : 2; else dim = (tmax[1] < tmax[2]) ? 1 : 2; }
    if ((step[dim] < 0 && key[dim] == 0) || (step[dim] > 0 && key[dim] == 2 * TREE_MAX_VAL - 1)) return 0;
    key[dim] += step[dim];
    tmax[dim] += tdelta[dim];
    for (int i = 0; i < 3; ++i) end[i] = occ_leaf_coord(key[i], resolution);
    int beyond = 0;
    if (max_range > 0.0) {
      double dsq = 0.0;
      for (int i = 0; i < 3; ++i) dsq += ((double)end[i] - 0.0) * ((double)end[i] - 0.0);
      beyond = dsq > max_range_sq;
    }
    if (beyond && !g_occ_variant.range_last) return 0;
    if (keyset_has(occ, pack_key(key[0], key[1], key[2]))) return 1;
    if (beyond) return 0;
  }
}

/* one castRay from the origin over the occupied leaves of `pts` (tests of the variants above): 1 + the hit leaf centre, or 0 */
int cr_cast_ray(const float* pts, int P, float resolution, const float* dir, float max_range, float* end) {
  keyset_t occ; keyset_init(&occ, (size_t)(P > 0 ? P : 1));
  const double res_factor = 1.0 / (double)resolution;
  for (int i = 0; i < P; ++i) {
    int k[3], ok = 1;
    for (int a = 0; a < 3; ++a) ok &= coord_to_key(pts[i * 3 + a], res_factor, &k[a]);
    if (ok) keyset_insert(&occ, pack_key(k[0], k[1], k[2]));
  }
  const int hit = cast_ray_origin(&occ, dir, (double)resolution, (double)max_range, end);
  free(occ.slots);
  return hit;
}

/* Returns the number of occupied lattice points written to out (capacity cap points, (x,y,z) float each);
 * if more would be produced the count is still returned (call again with a larger buffer). */
long cr_make_occupancy_grid(const float* pts, int P, float resolution, float* out, long cap) {
  if (P <= 0) return 0;
  const double res_d = (double)resolution, res_factor = 1.0 / res_d;
  keyset_t occ; keyset_init(&occ, (size_t)P);
  float mx[3] = {-INFINITY, -INFINITY, -INFINITY}, mn[3] = {INFINITY, INFINITY, INFINITY};
  for (int i = 0; i < P; ++i) {
    int k[3], ok = 1;
    for (int a = 0; a < 3; ++a) {
      ok &= coord_to_key(pts[i * 3 + a], res_factor, &k[a]);
      mx[a] = fmaxf(mx[a], pts[i * 3 + a]); mn[a] = fminf(mn[a], pts[i * 3 + a]);
    }
    if (ok) keyset_insert(&occ, pack_key(k[0], k[1], k[2]));
  }
  const float pad = 0.005f;
  const int max_xi = (int)((mx[0] + pad - (mn[0] - pad)) / resolution);
  const int max_yi = (int)((mx[1] + pad - (mn[1] - pad)) / resolution);
  const int max_zi = (int)((mx[2] + pad - (mn[2] - pad)) / resolution);
  /* float max_range = std::sqrt(std::pow(xmax+pad,2) + ...): pow/sqrt in double, result stored in a float */
  const float max_range = (float)sqrt(pow((double)(mx[0] + pad), 2) + pow((double)(mx[1] + pad), 2) + pow((double)(mx[2] + pad), 2));
  long n = 0;
  long* counts = (long*)calloc((size_t)(max_xi > 0 ? max_xi : 1), sizeof(long));
  /* pass 1: count per x-slab, pass 2: write at prefix offsets (deterministic lattice order) */
  for (int pass = 0; pass < 2; ++pass) {
    long* offs = NULL;
    if (pass == 1) {
      offs = (long*)malloc(sizeof(long) * (size_t)(max_xi > 0 ? max_xi : 1));
      long acc = 0; for (int xi = 0; xi < max_xi; ++xi) { offs[xi] = acc; acc += counts[xi]; }
      n = acc;
    }
#pragma omp parallel for schedule(dynamic)
    for (int xi = 0; xi < max_xi; ++xi) {
      long c = 0;
      for (int yi = 0; yi < max_yi; ++yi)
        for (int zi = 0; zi < max_zi; ++zi) {
          const float x = mn[0] - pad + xi * resolution, y = mn[1] - pad + yi * resolution, z = mn[2] - pad + zi * resolution;
          float dir[3] = {x, y, z};
          const float sq = (x * x + y * y) + z * z;
          if (sq > 0.0f) { const float nrm = sqrtf(sq); dir[0] = x / nrm; dir[1] = y / nrm; dir[2] = z / nrm; }
          float end[3];
          if (!cast_ray_origin(&occ, dir, res_d, (double)max_range, end)) continue;
          const float dist_query = sqrtf(x * x + y * y + z * z);
          const float dist = (float)sqrt((double)(end[0] * end[0] + end[1] * end[1] + end[2] * end[2]));
          if (g_occ_variant.strict ? dist < dist_query : dist <= dist_query) {
            if (pass == 1 && offs[xi] + c < cap) { float* o = out + (offs[xi] + c) * 3; o[0] = x; o[1] = y; o[2] = z; }
            ++c;
          }
        }
      if (pass == 0) counts[xi] = c;
    }
    if (offs) free(offs);
  }
  free(counts); free(occ.slots);
  return n;
}
