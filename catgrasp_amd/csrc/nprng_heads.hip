// HOST code (no kernel): the hypothesis draw of the 9-D RANSAC, `np.random.choice(n, size=k, replace=False)` once per
// hypothesis from numpy's GLOBAL generator (aligning.py:89-93: k = 4, n = 8192, 2 x 10,000 draws per object,
// predicter.py:167-170), replayed from numpy's own Mersenne-Twister state so a seeded run of the drop-in consumes the stream
// exactly like a seeded run of the reference.
//
// What numpy does per draw: permutation(n)[:k] = a full Fisher-Yates pass  for i = n-1 .. 1: j = bounded(i); swap(a[i], a[j])
// with bounded(i) = "next 32-bit word & mask(i) until <= i" -- 8,191 dependent steps and ~11,350 generator words for FOUR
// numbers.  Replaying it literally (cg_host_numpy_choice_rows) costs 30-60 us per hypothesis = 0.6-1.2 s per object on one
// core, two orders of magnitude more than the device work of the whole NUNOCS stage.  This file keeps the stream semantics
// and drops everything that is not needed for the k heads:
//   1. generator: blocks of 624 words produced OUT OF PLACE (new[kk] reads old[kk], old[kk+1] and either old[kk+397] or
//      new[kk-227]: three spans without a loop-carried dependence inside a vector) and tempered a block at a time, 16 words per
//      instruction.  (A second thread generating ahead through the ring was measured and dropped: handing 2.5 KB blocks from core
//      to core costs more than producing them -- 164 vs 80 ms per 20,000 draws on the EPYC 9575F host of the GPU box);
//   2. rejection walk: 16 .. 128 words per step.  accept(t) = (v_t <= i - #accepts before t) is resolved as the fixed point of
//      a <- (v + prefix_count(a) <= i) started from the upper bound (v <= i): word 0 is exact at once, word t after t rounds, and
//      the second round almost always repeats the first (an earlier word must have flipped AND this word must sit in the
//      one-wide gap that opens).  The loop-carried dependence is i alone (broadcast -> compare -> mask -> popcount, ~16 cycles):
//      the wider the step, the fewer trips through it, so the large masks -- where the chance of a third round stays ~1 % -- walk
//      128 / 64 words at a time.  The mask only changes when i crosses a power of two; a vector that crosses is cut at the accept
//      that reaches the boundary (pdep/tzcnt on the accept mask).  Accepted partners are compressed (vpcompressd) into a 16-bit
//      row buffer that stays in L1;
//   3. no permutation is ever built.  a[p] after the pass is found by undoing the swaps from the last to the first for the k
//      tracked positions only: for i >= k the tracked position is < i, so the only swap that moves it is one whose partner
//      j_i equals it -- a vector equality search over the partner row (32 partners per compare), ~ln(n) hits per head.
// Result: identical heads and identical generator state afterwards (tests/test_cabi_and_host.py pins both to numpy itself).
// Machines without AVX-512 (F, BW, VL, VPOPCNTDQ) + BMI2 take the scalar twin of the same three steps.
#include <immintrin.h>
#include <stdint.h>
#include <string.h>
#include "../../include/catgrasp_amd.h"

namespace {

constexpr int MT_N = 624, MT_M = 397;
constexpr int RING = 8;                                    // generator blocks kept (2 x 20 KB, L1/L2 resident)
constexpr long RING_WORDS = (long)RING * 624;
constexpr int MIRROR = 256;                                // words of slot 0 repeated behind the ring: a step may read across the wrap

#define CG_T512 __attribute__((target("avx512f,avx512bw,avx512vl,avx512dq,avx512vpopcntdq,bmi,bmi2,lzcnt,popcnt")))

// ---- generator: one block step, out of place -------------------------------------------------------------------------
#define CG_GEN_SPAN(a, src, dst, len)                                                          \
  for (int t = 0; t < (len); ++t) {                                                            \
    const uint32_t y = ((a)[t] & 0x80000000u) | ((a)[t + 1] & 0x7fffffffu);                    \
    (dst)[t] = (src)[t] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1u)) & 0x9908b0dfu);           \
  }

#define CG_BLOCK_BODY                                                                                            \
  {                                                                                                              \
    const uint32_t* __restrict o = old;                                                                          \
    uint32_t* __restrict n0 = neu;                                                                               \
    CG_GEN_SPAN(o, o + MT_M, n0, MT_N - MT_M)                               /* kk =   0..226: mt[kk+397] is old */ \
    { const uint32_t* __restrict s = neu; uint32_t* __restrict d = neu + 227; CG_GEN_SPAN(o + 227, s, d, 227) }  \
    { const uint32_t* __restrict s = neu + 227; uint32_t* __restrict d = neu + 454; CG_GEN_SPAN(o + 454, s, d, 169) } \
    const uint32_t y = (old[MT_N - 1] & 0x80000000u) | (neu[0] & 0x7fffffffu);                                   \
    neu[MT_N - 1] = neu[MT_M - 1] ^ (y >> 1) ^ ((uint32_t)(-(int32_t)(y & 1u)) & 0x9908b0dfu);                   \
    for (int t = 0; t < MT_N; ++t) {                                                                             \
      uint32_t z = neu[t];                                                                                       \
      z ^= (z >> 11); z ^= (z << 7) & 0x9d2c5680u; z ^= (z << 15) & 0xefc60000u; z ^= (z >> 18);                  \
      tmp[t] = z;                                                                                                \
    }                                                                                                            \
  }

// neu = the state block after `old` (numpy's mt19937_gen), tmp = neu tempered
CG_T512 void next_block_512(const uint32_t* old, uint32_t* neu, uint32_t* __restrict tmp) CG_BLOCK_BODY
void next_block_scalar(const uint32_t* old, uint32_t* neu, uint32_t* __restrict tmp) CG_BLOCK_BODY

inline void temper_block(const uint32_t* key, uint32_t* tmp) {
  for (int t = 0; t < MT_N; ++t) {
    uint32_t z = key[t];
    z ^= (z >> 11); z ^= (z << 7) & 0x9d2c5680u; z ^= (z << 15) & 0xefc60000u; z ^= (z >> 18);
    tmp[t] = z;
  }
}

// ---- the word source: a ring of raw + tempered generator blocks addressed by the GLOBAL word index -------------------
// Word g of the stream (g = 0 is word 0 of the caller's block) lives at tmp[g % RING_WORDS]; block b = words [624 b, 624 b + 624).
struct Stream {
  uint32_t* raw;                         // RING raw state blocks, flat
  uint32_t* tmp;                         // the same words tempered, + MIRROR words of slot 0 repeated behind the ring
  bool wide;                             // AVX-512 block step
  long fpos = 0;                         // global index of the next word
  long have = 1;                         // blocks 0 .. have-1 exist
  long start = 0;

  void step(long b) {                    // block b from block b-1
    uint32_t* t = tmp + (b % RING) * MT_N;
    if (wide) next_block_512(raw + ((b - 1) % RING) * MT_N, raw + (b % RING) * MT_N, t);
    else next_block_scalar(raw + ((b - 1) % RING) * MT_N, raw + (b % RING) * MT_N, t);
    if (b % RING == 0) memcpy(tmp + RING_WORDS, t, sizeof(uint32_t) * MIRROR);
  }
  // words [fpos, upto) are about to be read: make them exist; -> pointer to word fpos
  inline const uint32_t* need(long upto) {
    const long last = (upto - 1) / MT_N;                 // last block touched
    while (__builtin_expect(have <= last, 0)) step(have++);      // at most a block or two ahead of fpos: the block of the last
    return tmp + fpos % RING_WORDS;                              // consumed word (the final state) is never overwritten
  }
};

template <bool WIDE> void perm_rows(struct Stream& S, int n, int n_pts, long count, uint16_t* o, uint16_t* a, int* out);

// ---- per-row work, AVX-512 ------------------------------------------------------------------------------------------
#define CG_LANE_LOW _mm512_setr_epi32(0, 1, 3, 7, 15, 31, 63, 127, 255, 511, 1023, 2047, 4095, 8191, 16383, 32767)

// one round of the accept recurrence over NV vectors: b = (v + #accepts of `a` before the word <= i)
template <int NV>
CG_T512 inline uint32_t accept_round(const __m512i* v, const __mmask16* a, __mmask16* b, __m512i vi, __m512i* sum = nullptr) {
  const __m512i lane_low = CG_LANE_LOW;
  uint32_t base = 0;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    __m512i cnt = _mm512_popcnt_epi32(_mm512_and_si512(_mm512_set1_epi32((int)(uint32_t)a[q]), lane_low));
    if (q) cnt = _mm512_add_epi32(cnt, _mm512_set1_epi32((int)base));
    const __m512i sm = _mm512_add_epi32(v[q], cnt);
    if (sum) sum[q] = sm;
    b[q] = _mm512_cmple_epu32_mask(sm, vi);
    base += (uint32_t)_mm_popcnt_u32((uint32_t)a[q]);
  }
  return base;                                           // number of accepts in `a`
}

// NV x 16 words under one mask, none of which can reach the mask boundary (the caller checks i - lim > 16 NV)
template <int NV>
CG_T512 inline void walk_group(const uint32_t* w, __m512i vmask, uint32_t& i, uint16_t* o, int& s) {
  const __m512i vi = _mm512_set1_epi32((int)i);
  __m512i v[NV], sum[NV];
  __mmask16 a[NV], b[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    v[q] = _mm512_and_si512(_mm512_loadu_si512((const void*)(w + 16 * q)), vmask);
    b[q] = _mm512_cmple_epu32_mask(v[q], vi);            // a0, upper bound: as if nothing had been accepted before the word
  }
  const uint32_t n0 = accept_round<NV>(v, b, a, vi, sum); // a1 = (v + count0 <= i), a subset of a0 and a lower bound of the answer
  // Is a1 the answer?  With f = |a0 \ a1| words flipped, a word's true count lies in [count0 - f, count0]: the answer can differ from
  // a1 only in a word with v + count0 in (i, i + f].  One more compare per vector, at threshold i + f, rules that out (and repeats a1
  // outright when f = 0) -- instead of a whole further round of prefix counts.
  uint32_t n1 = 0;
#pragma unroll
  for (int q = 0; q < NV; ++q) n1 += (uint32_t)_mm_popcnt_u32((uint32_t)a[q]);
  const __m512i vif = _mm512_set1_epi32((int)(i + (n0 - n1)));
  uint32_t diff = 0;
#pragma unroll
  for (int q = 0; q < NV; ++q) diff |= (uint32_t)(_mm512_cmple_epu32_mask(sum[q], vif) ^ a[q]);
  if (__builtin_expect(diff != 0, 0)) {                   // rare: iterate to the fixed point of the recurrence = the sequential answer
    for (;;) {
      accept_round<NV>(v, a, b, vi);
      diff = 0;
#pragma unroll
      for (int q = 0; q < NV; ++q) { diff |= (uint32_t)(a[q] ^ b[q]); a[q] = b[q]; }
      if (diff == 0) break;
    }
  }
  uint32_t nacc = 0;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    _mm256_storeu_si256((__m256i*)(o + s), _mm512_cvtepi32_epi16(_mm512_maskz_compress_epi32(a[q], v[q])));
    const uint32_t c = (uint32_t)_mm_popcnt_u32((uint32_t)a[q]);
    s += (int)c; nacc += c;
  }
  i -= nacc;
}

// Fisher-Yates partners of one permutation(n) pass: o[s] = j_i, s = n-1-i, i = n-1 .. 1.  o needs 16 entries of slack.
CG_T512 inline void partners_512(Stream& S, int n, uint16_t* o) {
  uint32_t i = (uint32_t)(n - 1), mask = i;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  int s = 0;
  while (i > 0) {
    const uint32_t lim = mask >> 1;                       // this mask serves the steps i in (lim, mask]
    if (i <= lim) { mask = lim; continue; }
    const __m512i vmask = _mm512_set1_epi32((int)mask);
    if (mask >= 4095)
      while (i - lim > 128) { walk_group<8>(S.need(S.fpos + 128), vmask, i, o, s); S.fpos += 128; }
    if (mask >= 1023)
      while (i - lim > 64) { walk_group<4>(S.need(S.fpos + 64), vmask, i, o, s); S.fpos += 64; }
    while (i > lim) {                                     // one vector at a time, cut where the mask changes
      const __m512i v = _mm512_and_si512(_mm512_loadu_si512((const void*)S.need(S.fpos + 16)), vmask);
      const __m512i vi = _mm512_set1_epi32((int)i);
      __mmask16 a = _mm512_cmple_epu32_mask(v, vi), a2;
      for (;;) {
        accept_round<1>(&v, &a, &a2, vi);
        if (a2 == a) break;
        a = a2;
      }
      uint32_t nacc = (uint32_t)_mm_popcnt_u32((uint32_t)a);
      int used = 16;
      const uint32_t room = i - lim;                      // accepts left under this mask
      if (nacc >= room) {                                 // cut behind the accept that reaches the boundary
        const uint32_t lane = (uint32_t)_tzcnt_u32(_pdep_u32(1u << (room - 1), (uint32_t)a));
        a &= (__mmask16)((2u << lane) - 1u);
        used = (int)lane + 1;
        nacc = room;
      }
      _mm256_storeu_si256((__m256i*)(o + s), _mm512_cvtepi32_epi16(_mm512_maskz_compress_epi32(a, v)));
      s += (int)nacc; i -= nacc; S.fpos += used;
    }
  }
}
// heads[t] = permutation(n)[t], t < k, from the partner row: undo the swaps i = 1 .. n-1 for the k tracked positions
CG_T512 inline void heads_512(const uint16_t* o, int n, int k, int* heads) {
  uint32_t p[16];
  for (int t = 0; t < k; ++t) p[t] = (uint32_t)t;
  const int kk = k < n ? k : n;
  for (int i = 1; i < kk; ++i) {                          // below k a tracked position can sit ON i: the full swap rule
    const uint32_t j = o[n - 1 - i];
    for (int t = 0; t < k; ++t) p[t] = p[t] == (uint32_t)i ? j : (p[t] == j ? (uint32_t)i : p[t]);
  }
  const int S = n - 1 - k;                                // the partners of i = k .. n-1 sit at s = S .. 0
  if (S >= 0) {
    __m512i bp[16];
    for (int t = 0; t < k; ++t) bp[t] = _mm512_set1_epi16((short)p[t]);
    for (int b = S & ~31; b >= 0; b -= 32) {
      const int top = S - b;                              // highest valid lane of this chunk
      const __mmask32 valid = top >= 31 ? 0xffffffffu : ((2u << top) - 1u);
      const __m512i vec = _mm512_maskz_loadu_epi16(valid, o + b);
      __mmask32 any = 0;
      for (int t = 0; t < k; ++t) any |= _mm512_mask_cmpeq_epi16_mask(valid, vec, bp[t]);
      if (!any) continue;
      for (int t = 0; t < k; ++t) {                       // rare (~ln n chunks per head): walk this head through the chunk, high lane first
        __mmask32 m = _mm512_mask_cmpeq_epi16_mask(valid, vec, bp[t]);
        while (m) {
          const int h = 31 - (int)_lzcnt_u32((uint32_t)m);
          p[t] = (uint32_t)(n - 1 - (b + h));
          bp[t] = _mm512_set1_epi16((short)p[t]);
          m = h ? _mm512_mask_cmpeq_epi16_mask(valid & (((__mmask32)1u << h) - 1u), vec, bp[t]) : 0;
        }
      }
    }
  }
  for (int t = 0; t < k; ++t) heads[t] = (int)p[t];
}

CG_T512 void perm_rows_512(Stream& S, int n, int n_pts, long count, uint16_t* o, uint16_t* a, int* out) { perm_rows<true>(S, n, n_pts, count, o, a, out); }
CG_T512 void partners_rows_512(Stream& S, int n, uint16_t* o) { partners_512(S, n, o); }

CG_T512 void rows_512(Stream& S, int n, int k, long count, uint16_t* o, int* out) {
  for (long r = 0; r < count; ++r) {
    partners_512(S, n, o);
    heads_512(o, n, k, out + r * k);
  }
}

// ---- per-row work, scalar twin ---------------------------------------------------------------------------------------
void partners_scalar(Stream& S, int n, uint16_t* o) {
  const uint32_t n1 = (uint32_t)(n - 1);
  uint32_t i = n1, mask = n1;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  while (i > 0) {
    const uint32_t lim = mask >> 1;
    if (i <= lim) { mask = lim; continue; }
    const int avail = MT_N - (int)(S.fpos % MT_N);      // to the end of this generator block
    const uint32_t* w = S.need(S.fpos + avail);
    int t = 0;
    for (; t < avail && i > lim; ++t) {                 // branch-free rejection: store, advance by the accept bit
      const uint32_t v = w[t] & mask;
      o[n1 - i] = (uint16_t)v;
      i = i - 1 + (i < v);
    }
    S.fpos += t;
  }
}

void rows_scalar(Stream& S, int n, int k, long count, uint16_t* o, int* out) {
  for (long r = 0; r < count; ++r) {
    partners_scalar(S, n, o);
    uint32_t p[16];
    for (int t = 0; t < k; ++t) p[t] = (uint32_t)t;
    for (int i2 = 1; i2 < n; ++i2) {
      const uint32_t j = o[n - 1 - i2], ii = (uint32_t)i2;
      for (int t = 0; t < k; ++t) p[t] = p[t] == ii ? j : (p[t] == j ? ii : p[t]);
    }
    for (int t = 0; t < k; ++t) out[r * k + t] = (int)p[t];
  }
}

// whole rows: permutation(n)[:n_pts] from the partner row (the swap chain itself, 2 ns a step on a row that sits in L1)
template <bool WIDE>
void perm_rows(Stream& S, int n, int n_pts, long count, uint16_t* o, uint16_t* a, int* out) {
  for (long r = 0; r < count; ++r) {
    if (WIDE) partners_512(S, n, o); else partners_scalar(S, n, o);
    for (int i = 0; i < n; ++i) a[i] = (uint16_t)i;
    for (int i = n - 1; i > 0; --i) {
      const uint16_t j = o[n - 1 - i], t = a[i];
      a[i] = a[j]; a[j] = t;
    }
    int* q = out + r * n_pts;
    for (int i = 0; i < n_pts; ++i) q[i] = a[i];
  }
}

bool cpu_has_avx512() {
  return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") &&
         __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vpopcntdq") && __builtin_cpu_supports("bmi2");
}

}  // namespace

// `count` draws of np.random.choice(n, size=k, replace=False) = permutation(n)[:k] from the generator state (key, pos) of
// np.random.get_state(); the state is advanced exactly as numpy would have.  2 <= n <= 65536, 1 <= k <= min(n, 16).
// isa: 0 = pick at run time, 1 = force the scalar twin (tests).
extern "C" int cg_host_numpy_choice_heads(uint32_t* h_mt_key624, int* h_mt_pos, int n, int k, long count, int isa, int* h_out) {
  if (!h_mt_key624 || !h_mt_pos || n < 2 || n > 65536 || k < 1 || k > 16 || k > n || count < 0 || *h_mt_pos < 0 || *h_mt_pos > MT_N)
    return CG_ERR_ARG;
  if (count == 0) return CG_OK;
  if (!h_out) return CG_ERR_ARG;
  Stream S;
  S.raw = new uint32_t[(size_t)RING_WORDS];
  S.tmp = new uint32_t[(size_t)RING_WORDS + MIRROR];
  uint16_t* row = new uint16_t[(size_t)n + 64];
  S.wide = isa != 1 && cpu_has_avx512();
  memcpy(S.raw, h_mt_key624, sizeof(uint32_t) * MT_N);
  temper_block(S.raw, S.tmp);
  S.fpos = S.start = *h_mt_pos;
  if (S.wide) rows_512(S, n, k, count, row, h_out); else rows_scalar(S, n, k, count, row, h_out);
  if (S.fpos > S.start) {                 // numpy regenerates lazily: after the last word of a block the state is (that block, 624)
    const long b = (S.fpos - 1) / MT_N;
    memcpy(h_mt_key624, S.raw + (b % RING) * MT_N, sizeof(uint32_t) * MT_N);
    *h_mt_pos = (int)(S.fpos - b * MT_N);
  }
  delete[] row; delete[] S.tmp; delete[] S.raw;
  return CG_OK;
}

// `count` draws of np.random.choice(n, size=n_pts, replace=False) = permutation(n)[:n_pts] as WHOLE rows on the host, 2 <= n <= 65536,
// 1 <= n_pts <= n: the vectorised partner extraction above + the swap chain on a row that sits in L1 (~6 us per row at n = 2,500
// against ~14 us for cg_host_numpy_choice_rows).  What a small predict_batch call uses instead of shipping partners to the device:
// a swap chain is a dependent sequence wherever it runs, and one lane of the device needs ~250 us for it.
extern "C" int cg_host_numpy_permutation_rows(uint32_t* h_mt_key624, int* h_mt_pos, int n, int n_pts, long count, int isa, int* h_out) {
  if (!h_mt_key624 || !h_mt_pos || n < 2 || n > 65536 || n_pts < 1 || n_pts > n || count < 0 || *h_mt_pos < 0 || *h_mt_pos > MT_N)
    return CG_ERR_ARG;
  if (count == 0) return CG_OK;
  if (!h_out) return CG_ERR_ARG;
  Stream S;
  S.raw = new uint32_t[(size_t)RING_WORDS];
  S.tmp = new uint32_t[(size_t)RING_WORDS + MIRROR];
  uint16_t* row = new uint16_t[(size_t)n + 64];
  uint16_t* perm = new uint16_t[(size_t)n];
  S.wide = isa != 1 && cpu_has_avx512();
  memcpy(S.raw, h_mt_key624, sizeof(uint32_t) * MT_N);
  temper_block(S.raw, S.tmp);
  S.fpos = S.start = *h_mt_pos;
  if (S.wide) perm_rows_512(S, n, n_pts, count, row, perm, h_out); else perm_rows<false>(S, n, n_pts, count, row, perm, h_out);
  if (S.fpos > S.start) {
    const long b = (S.fpos - 1) / MT_N;
    memcpy(h_mt_key624, S.raw + (b % RING) * MT_N, sizeof(uint32_t) * MT_N);
    *h_mt_pos = (int)(S.fpos - b * MT_N);
  }
  delete[] perm; delete[] row; delete[] S.tmp; delete[] S.raw;
  return CG_OK;
}

// The sequential part of `count` calls of permutation(n_valid) alone: the Fisher-Yates swap partners j(i), i = n_valid-1 .. 1, of
// every row, as u16 (row r at h_j + r*row_stride, step s = n_valid-1-i at [s]; the tail of the stride is zero-filled).  Consumes the
// generator exactly like cg_host_numpy_choice_rows' replace=False branch -- the rejection loop is what makes the stream sequential --
// and leaves the swap chain itself to the device (cg_apply_shuffle_rows), which runs one chain per lane.  Same vectorised generator
// and rejection walk as the hypothesis draw above (~1.5 us per row at n_valid = 2,500 on the AVX-512 path; the scalar walk: ~5 us).
extern "C" int cg_host_numpy_shuffle_partners(uint32_t* h_mt_key624, int* h_mt_pos, int n_valid, long count, long row_stride, uint16_t* h_j) {
  if (!h_mt_key624 || !h_mt_pos || n_valid < 2 || n_valid > 65536 || count < 0 || row_stride < n_valid - 1 || *h_mt_pos < 0 || *h_mt_pos > MT_N)
    return CG_ERR_ARG;
  if (count == 0) return CG_OK;
  if (!h_j) return CG_ERR_ARG;
  Stream S;
  S.raw = new uint32_t[(size_t)RING_WORDS];
  S.tmp = new uint32_t[(size_t)RING_WORDS + MIRROR];
  uint16_t* row = new uint16_t[(size_t)n_valid + 64];      // the walk stores whole vectors: a row is produced here and copied out
  S.wide = cpu_has_avx512();
  memcpy(S.raw, h_mt_key624, sizeof(uint32_t) * MT_N);
  temper_block(S.raw, S.tmp);
  S.fpos = S.start = *h_mt_pos;
  for (long r = 0; r < count; ++r) {
    if (S.wide) partners_rows_512(S, n_valid, row); else partners_scalar(S, n_valid, row);
    uint16_t* o = h_j + r * row_stride;
    memcpy(o, row, sizeof(uint16_t) * (size_t)(n_valid - 1));
    for (long k = n_valid - 1; k < row_stride; ++k) o[k] = 0;
  }
  if (S.fpos > S.start) {
    const long b = (S.fpos - 1) / MT_N;
    memcpy(h_mt_key624, S.raw + (b % RING) * MT_N, sizeof(uint32_t) * MT_N);
    *h_mt_pos = (int)(S.fpos - b * MT_N);
  }
  delete[] row; delete[] S.tmp; delete[] S.raw;
  return CG_OK;
}
