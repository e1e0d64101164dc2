// HOST code (no kernel): numpy's legacy global generator, replayed in C so that the reference-exact resampling draw of
// GraspPredicter.predict_batch is not capped by python.
//
// The reference draws one `np.random.choice(np.arange(M), size=n_pts, replace=M<n_pts)` per grasp pose from numpy's GLOBAL
// RandomState (dataset_grasp.py:72-73 inside the loop of predicter.py:71-74), so a seeded reference run and a seeded run of the
// drop-in must consume the Mersenne Twister identically.  In numpy that call costs ~40 us (25k poses/s); its arithmetic is
//   replace=False:  permutation(M)[:n_pts]   = arange(M) shuffled by  for i = M-1 .. 1: j = random_interval(i); swap(a[i], a[j])
//   replace=True :  randint(0, M, n_pts)     = masked rejection per element
// with random_interval / the bounded draw = "next_uint32 & mask until <= max" (numpy/random/src/distributions/distributions.c,
// legacy-seeding MT19937 of numpy/random/src/mt19937/mt19937.c).  This file replays exactly that on the 624-word state the caller
// takes from np.random.get_state() and hands back through np.random.set_state(): ~14 us per pose, and -- being plain C without the
// GIL -- it runs on a worker thread while the device scores the previous chunk.  tests/test_cabi_and_host.py pins it to numpy
// itself (outputs and the generator state afterwards), for both branches and across state regenerations.
#include <stdint.h>
#include "../../include/catgrasp_amd.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;

// key: numpy's raw state words; tb: the same block after tempering (filled a whole block at a time, which the compiler vectorises,
// so the hot loops only index it)
struct MT { uint32_t* key; int pos; uint32_t tb[624]; };

inline uint32_t temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

inline void mt_gen(uint32_t* mt) {
  const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
  int kk;
  uint32_t y;
  for (kk = 0; kk < MT_N - MT_M; kk++) {
    y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
    mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
  }
  for (; kk < MT_N - 1; kk++) {
    y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
    mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
  }
  y = (mt[MT_N - 1] & UPPER) | (mt[0] & LOWER);
  mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
}

inline void temper_block(MT& s) {
  for (int i = 0; i < MT_N; ++i) s.tb[i] = temper(s.key[i]);
}

inline uint32_t mt_next(MT& s) {
  if (s.pos == MT_N) { mt_gen(s.key); temper_block(s); s.pos = 0; }
  return s.tb[s.pos++];
}

inline uint32_t gen_mask(uint32_t max) {
  uint32_t mask = max;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  return mask;
}

inline uint32_t bounded(MT& s, uint32_t max, uint32_t mask) {      // uniform in [0, max]
  if (max == 0) return 0;
  uint32_t v;
  while ((v = (mt_next(s) & mask)) > max) {}
  return v;
}

}  // namespace

extern "C" int cg_host_numpy_choice_rows(uint32_t* h_mt_key624, int* h_mt_pos, int n_valid, int n_pts, long count, int* h_scratch,
                                         int* h_out) {
  if (!h_mt_key624 || !h_mt_pos || n_valid <= 0 || n_pts <= 0 || count < 0 || *h_mt_pos < 0 || *h_mt_pos > MT_N) return CG_ERR_ARG;
  if (count == 0) return CG_OK;
  if (!h_out) return CG_ERR_ARG;
  MT s;
  s.key = h_mt_key624; s.pos = *h_mt_pos;
  temper_block(s);
  if (n_valid < n_pts) {                                  // replace=True: randint(0, n_valid, n_pts)
    const uint32_t rng = (uint32_t)n_valid - 1u, mask = gen_mask(rng);
    for (long r = 0; r < count; ++r) {
      int* o = h_out + r * n_pts;
      for (int i = 0; i < n_pts; ++i) o[i] = (int)bounded(s, rng, mask);
    }
  } else {                                                // replace=False: permutation(n_valid)[:n_pts]
    if (!h_scratch) return CG_ERR_ARG;                    // n_valid ints
    for (long r = 0; r < count; ++r) {
      for (int i = 0; i < n_valid; ++i) h_scratch[i] = i;
      uint32_t mask = gen_mask((uint32_t)(n_valid - 1));
      for (int i = n_valid - 1; i > 0; --i) {
        if ((uint32_t)i <= (mask >> 1)) mask >>= 1;          // = gen_mask(i): the smallest 2^k - 1 >= i
        const uint32_t j = bounded(s, (uint32_t)i, mask);
        const int t = h_scratch[i]; h_scratch[i] = h_scratch[j]; h_scratch[j] = t;
      }
      int* o = h_out + r * n_pts;
      for (int i = 0; i < n_pts; ++i) o[i] = h_scratch[i];
    }
  }
  *h_mt_pos = s.pos;
  return CG_OK;
}

// cg_host_numpy_shuffle_partners (the sequential part of permutation(n_valid) alone: the swap partners, for the device's swap chains) lives in
// nprng_heads.hip with the vectorised generator and rejection walk it shares with the hypothesis draw.
