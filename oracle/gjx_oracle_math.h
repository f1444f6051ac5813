/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load anything under oracle/.
 *
 * Plain-C restatement of the arithmetic the SMC / ImportanceK hot path performs.  In the
 * reference that arithmetic lives in third-party code that is NOT vendored under /root/reference
 * (SURVEY F4): jax 0.5.2 / jaxlib 0.5.1 (jax.random threefry2x32, jax.scipy.special.logsumexp)
 * and tensorflow-probability 0.23.0 (tfd.Normal/Gamma/Beta/Bernoulli/Categorical sample/log_prob),
 * called from generative_functions/distributions/tensorflow_probability/__init__.py:52-62.
 * We restate the published algorithms (Salmon et al. SC'11 for Threefry/Philox; Giles 2010 erfinv
 * as used by XLA; Marsaglia-Tsang 2000 gamma; TFP log_prob formulas) with a fully
 * specified f32 operation order so that the HIP kernels can be bit-compared.
 *
 * PARITY PINNING: the reference cannot be imported here (missing jax/tfp/beartype,
 * Python 3.10 < 3.11) and its tests hold no random golden vectors.  This oracle is
 * pinned by: Random123 known-answer vectors (Threefry2x32-20, Philox4x32-10), scipy float64
 * log-densities, the reference tests' closed-form answers (tests/inference/test_smc.py:32-87,
 * tests/generative_functions/test_static_gen_fn.py:317-318) and analytic log-Z (Kalman / HMM
 * forward / conjugate Gaussian).  Bit-level parity with real JAX output is UNPINNED.
 */
#ifndef GJX_ORACLE_MATH_H
#define GJX_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline uint32_t o_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float o_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ---------------- ciphers (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as
 * 1, 2, 3", SC'11).  jax.random's default implementation is threefry2x32. -------- */

static inline uint32_t o_rotl(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

static inline void o_threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1,
                                  uint32_t* o0, uint32_t* o1) {
  static const int R[8] = {13, 15, 26, 6, 17, 29, 16, 24};
  uint32_t ks[3] = {k0, k1, 0x1BD11BDAu ^ k0 ^ k1};
  uint32_t x0 = c0 + ks[0], x1 = c1 + ks[1];
  for (int blk = 0; blk < 5; ++blk) {
    for (int r = 0; r < 4; ++r) {
      x0 += x1;
      x1 = o_rotl(x1, R[(blk & 1) * 4 + r]);
      x1 ^= x0;
    }
    x0 += ks[(blk + 1) % 3];
    x1 += ks[(blk + 2) % 3] + (uint32_t)(blk + 1);
  }
  *o0 = x0;
  *o1 = x1;
}

static inline void o_philox4x32(uint32_t k0, uint32_t k1, const uint32_t c[4], uint32_t out[4]) {
  uint32_t c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ---------------- key derivation ----------------------------------------- *
 * A key is 4 words (k0, k1, lane_lo, lane_hi); THREEFRY uses k0,k1 only (lane 0).
 * THREEFRY = jax.random semantics with jax_threefry_partitionable (default from jax 0.5.0,
 * SURVEY App. A): split(k,n)[i] = TF(k,(hi(i),lo(i))); fold_in(k,d) = TF(k,(0,d)).
 * PHILOX (native): every block is PH(ctr = (lane_lo, lane_hi, b, (a << 8) | tag), key = (k0,k1)).
 *   split of a lane-0 key: child i = (k0, k1, lane i+1)             — no cipher block
 *   split of a laned key:  child i = words 0,1 of block (b = i_lo, a = i_hi, 'S'), lane 0
 *   fold_in(k, d)        = words 0,1 of block (b = d, 'F'), lane 0
 *   packed single-word draw number f = word f&3 of block (b = f>>2, 'D')
 *   sub-stream s of a stream with optional fold f = words 0,1 of block (b = s, a = has_fold ? f+1 : 0, 'R') */
#define O_TAG_SPLIT 0x53u
#define O_TAG_FOLD 0x46u
#define O_TAG_DRAW 0x44u
#define O_TAG_STREAM 0x52u

static inline void o_key_copy(uint32_t dst[4], const uint32_t src[4]) { memcpy(dst, src, 16); }
static inline void o_philox_lane(const uint32_t k[4], uint32_t b, uint32_t a_tag, uint32_t o[4]) {
  uint32_t c[4] = {k[2], k[3], b, a_tag};
  o_philox4x32(k[0], k[1], c, o);
}
static inline void o_split_at(int impl, const uint32_t k[4], uint64_t i, uint32_t out[4]) {
  uint32_t r[4] = {0u, 0u, 0u, 0u};
  if (impl == 0) {
    o_threefry2x32(k[0], k[1], (uint32_t)(i >> 32), (uint32_t)i, &r[0], &r[1]);
  } else if ((k[2] | k[3]) == 0u) {
    const uint64_t lane = i + 1u;
    r[0] = k[0]; r[1] = k[1]; r[2] = (uint32_t)lane; r[3] = (uint32_t)(lane >> 32);
  } else {
    uint32_t o[4];
    o_philox_lane(k, (uint32_t)i, ((uint32_t)(i >> 32) << 8) | O_TAG_SPLIT, o);
    r[0] = o[0]; r[1] = o[1];
  }
  o_key_copy(out, r);
}
static inline void o_fold_in(int impl, const uint32_t k[4], uint32_t d, uint32_t out[4]) {
  uint32_t r[4] = {0u, 0u, 0u, 0u};
  if (impl == 0) {
    o_threefry2x32(k[0], k[1], 0u, d, &r[0], &r[1]);
  } else {
    uint32_t o[4];
    o_philox_lane(k, d, O_TAG_FOLD, o);
    r[0] = o[0]; r[1] = o[1];
  }
  o_key_copy(out, r);
}
/* A draw stream = (key, optional leaf-site fold).  THREEFRY: the stream key is
 * fold_in(key, fold) with fold = the site counter (static.py:349-352) and sub-stream `sub` is
 * TF(stream key, (0, sub)).  PHILOX: the fold (0-based index among the sampled sites) is carried in
 * the counter instead of costing a block. */
typedef struct { int impl; uint32_t k[4]; uint32_t f; uint32_t hf; } o_stream;

static inline o_stream o_stream_make(int impl, const uint32_t key[4], int has_fold, uint32_t fold) {
  o_stream s;
  s.impl = impl; s.f = 0u; s.hf = 0u;
  o_key_copy(s.k, key);
  if (has_fold) {
    if (impl == 0) o_fold_in(0, key, fold, s.k);
    else { s.f = fold; s.hf = 1u; }
  }
  return s;
}
/* Two raw words of sub-stream `sub`. */
static inline void o_words(const o_stream* s, uint32_t sub, uint32_t* w0, uint32_t* w1) {
  if (s->impl == 0) {
    o_threefry2x32(s->k[0], s->k[1], 0u, sub, w0, w1);
  } else {
    uint32_t o[4];
    o_philox_lane(s->k, sub, ((s->hf ? s->f + 1u : 0u) << 8) | O_TAG_STREAM, o);
    *w0 = o[0]; *w1 = o[1];
  }
}
/* 32 random bits of element `sub` — THREEFRY: jax _threefry_random_bits_partitionable, hi ^ lo.
 * PHILOX: word 0 of the sub-stream, except the single-word draw (sub 0) of a folded stream, which
 * is word (f & 3) of the block shared by folds 4(f>>2)..4(f>>2)+3 — one cipher block serves four
 * scalar sites. */
/* PHILOX single-word draw number f of key k.  A lane-0 key packs four draws per block of its
 * own: word f & 3 of PH(ctr = (0, 0, f >> 2, 'D')).  A laned key (lane L >= 1: particle i = L - 1 of its parent)
 * shares blocks with its pair partner i ^ 1: word ((f & 1) << 1) | (i & 1) of PH(ctr = (p_lo, p_hi, f >> 1, 'P')),
 * p = i >> 1. */
#define O_TAG_PAIR 0x50u
static inline uint32_t o_single_draw(const uint32_t k[4], uint32_t f) {
  uint32_t o[4];
  if ((k[2] | k[3]) == 0u) {
    const uint32_t c[4] = {0u, 0u, f >> 2, O_TAG_DRAW};
    o_philox4x32(k[0], k[1], c, o);
    return o[f & 3u];
  }
  const uint64_t i = (((uint64_t)k[3] << 32) | k[2]) - 1u, p = i >> 1;
  const uint32_t c[4] = {(uint32_t)p, (uint32_t)(p >> 32), f >> 1, O_TAG_PAIR};
  o_philox4x32(k[0], k[1], c, o);
  return o[((f & 1u) << 1) | (uint32_t)(i & 1u)];
}
static inline uint32_t o_bits32_at(const o_stream* s, uint32_t sub) {
  uint32_t w0, w1;
  if (s->impl == 1 && s->hf && sub == 0u) return o_single_draw(s->k, s->f);
  o_words(s, sub, &w0, &w1);
  return s->impl == 0 ? (w0 ^ w1) : w0;
}
static inline uint64_t o_bits64_at(const o_stream* s, uint32_t sub) {
  uint32_t w0, w1;
  o_words(s, sub, &w0, &w1);
  return ((uint64_t)w0 << 32) | w1;
}

/* The 32-bit draw of SMC slot j at one step of the fixed-model filters.  THREEFRY: the first
 * single-word draw of the slot key split(step_key)[j] (site counter 1).  PHILOX: slots 4g .. 4g+3 share the block
 * PH(ctr = (g_lo, g_hi, 0, 'Q'), key = step_key) and slot j takes word j & 3. */
#define O_TAG_QUAD 0x51u
/* One-word draw number f of SMC slot j under PHILOX: word j & 3 of PH(ctr = (g_lo, g_hi, f, 'Q'), key = step key),
 * g = j / 4 (the fixed-model filters have one draw per step: f = 0). */
static inline uint32_t o_smc_quad_word(const uint32_t step_key[4], uint64_t j, uint32_t f) {
  const uint64_t g = j >> 2;
  const uint32_t c[4] = {(uint32_t)g, (uint32_t)(g >> 32), f, O_TAG_QUAD};
  uint32_t o[4];
  o_philox4x32(step_key[0], step_key[1], c, o);
  return o[j & 3u];
}
static inline uint32_t o_smc_slot_bits(int impl, const uint32_t step_key[4], uint64_t j) {
  if (impl == 0) {
    uint32_t pk[4];
    o_split_at(impl, step_key, j, pk);
    o_stream st = o_stream_make(impl, pk, 1, 1u);
    return o_bits32_at(&st, 0);
  }
  return o_smc_quad_word(step_key, j, 0u);
}

/* ---------------- f32 math spec: only IEEE-exact primitives -------------- *
 * (+, -, *, fmaf, /, sqrtf, rintf, integer ops).  Coefficients: Cephes logf/expf
 * (Moshier), Giles' single-precision erfinv (the polynomial XLA's ErfInv32 uses). */

static inline float o_log(float x) {
  uint32_t ix = o_f2u(x);
  int32_t e = 0;
  if (ix - 1u >= 0x7f7fffffu) { /* +-0, +inf, NaN, every negative: log's own values */
    if ((ix << 1) == 0u) return -INFINITY;
    return ix == 0x7f800000u ? x : o_u2f(0x7fc00000u);
  }
  if (ix < 0x00800000u) { x = x * 8388608.0f; ix = o_f2u(x); e = -23; }
  uint32_t t = ix - 0x3f3504f3u;
  e += (int32_t)t >> 23;
  float m = o_u2f((t & 0x007fffffu) + 0x3f3504f3u);
  float f = m - 1.0f;
  float z = f * f;
  float p = 7.0376836292E-2f;
  p = fmaf(p, f, -1.1514610310E-1f);
  p = fmaf(p, f, 1.1676998740E-1f);
  p = fmaf(p, f, -1.2420140846E-1f);
  p = fmaf(p, f, 1.4249322787E-1f);
  p = fmaf(p, f, -1.6668057665E-1f);
  p = fmaf(p, f, 2.0000714765E-1f);
  p = fmaf(p, f, -2.4999993993E-1f);
  p = fmaf(p, f, 3.3333331174E-1f);
  float y = (p * f) * z;
  float fe = (float)e;
  y = fmaf(fe, -2.12194440e-4f, y);
  y = fmaf(-0.5f, z, y);
  float r = f + y;
  r = fmaf(fe, 0.693359375f, r);
  return r;
}

static inline float o_exp(float x) {
  if (!(x >= -86.0f)) return 0.0f; /* also NaN -> 0 */
  if (x > 88.0f) x = 88.0f;
  float fx = rintf(x * 1.44269504088896341f);
  x = fmaf(fx, -0.693359375f, x);
  x = fmaf(fx, 2.12194440e-4f, x);
  float z = x * x;
  float p = 1.9875691500E-4f;
  p = fmaf(p, x, 1.3981999507E-3f);
  p = fmaf(p, x, 8.3334519073E-3f);
  p = fmaf(p, x, 4.1665795894E-2f);
  p = fmaf(p, x, 1.6666665459E-1f);
  p = fmaf(p, x, 5.0000001201E-1f);
  float y = fmaf(p, z, x) + 1.0f;
  int32_t n = (int32_t)fx;
  return o_u2f(o_f2u(y) + ((uint32_t)n << 23));
}

/* exp as a model body writes it (GJX_EXPR_EXP, gjx_map_f32) */
static inline float o_e_exp(float x) {
  if (x != x) return x;
  if (x > 88.0f || x < -86.0f) {
    const float h = o_exp(x * 0.5f);
    return h * h;
  }
  return o_exp(x);
}

/* max / min as a model body writes them (GJX_EXPR_MAX / _MIN): a NaN if either argument is one */
static inline float o_e_max(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }
static inline float o_e_min(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }

static inline float o_erfinv(float x) {
  float w = -o_log((1.0f - x) * (1.0f + x));
  float p;
  if (w < 5.0f) {
    w = w - 2.5f;
    p = 2.81022636e-08f;
    p = fmaf(p, w, 3.43273939e-07f);
    p = fmaf(p, w, -3.5233877e-06f);
    p = fmaf(p, w, -4.39150654e-06f);
    p = fmaf(p, w, 0.00021858087f);
    p = fmaf(p, w, -0.00125372503f);
    p = fmaf(p, w, -0.00417768164f);
    p = fmaf(p, w, 0.246640727f);
    p = fmaf(p, w, 1.50140941f);
  } else {
    w = sqrtf(w) - 3.0f;
    p = -0.000200214257f;
    p = fmaf(p, w, 0.000100950558f);
    p = fmaf(p, w, 0.00134934322f);
    p = fmaf(p, w, -0.00367342844f);
    p = fmaf(p, w, 0.00573950773f);
    p = fmaf(p, w, -0.0076224613f);
    p = fmaf(p, w, 0.00943887047f);
    p = fmaf(p, w, 1.00167406f);
    p = fmaf(p, w, 2.83297682f);
  }
  return p * x;
}

/* lgamma for x > 0: recurrence up to x >= 8, then Stirling with three correction terms. */
static inline float o_lgamma(float x) {
  float p = 1.0f;
  for (int i = 0; i < 8; ++i) {
    if (x < 8.0f) { p = p * x; x = x + 1.0f; }
  }
  float xi = 1.0f / x;
  float xi2 = xi * xi;
  float s = fmaf(xi2, 7.9365079365e-4f, -2.7777777778e-3f);
  s = fmaf(s, xi2, 8.3333333333e-2f);
  s = s * xi;
  float r = (x - 0.5f) * o_log(x);
  r = r - x;
  r = r + 0.91893853320467f;
  r = r + s;
  r = r - o_log(p);
  return r;
}

/* jax.random.uniform bit trick: 23 mantissa bits -> [0,1). */
static inline float o_uniform01(uint32_t bits) { return o_u2f((bits >> 9) | 0x3F800000u) - 1.0f; }

/* jax.random.normal: u ~ U(nextafter(-1,0), 1); sqrt(2) * erfinv(u)  (SURVEY App. A). */
static inline float o_std_normal(uint32_t bits) {
  const float lo = -0.99999994f;
  float u = o_uniform01(bits) * 2.0f + lo;
  u = u > lo ? u : lo;
  return 1.41421356237309505f * o_erfinv(u);
}

/* ---- PHILOX Normal sites: Box-Muller over pairs of particles ----------------------
 * Particles j0 = j & ~1 and j1 = j0 + 1 of a key batch (key lanes j0+1, j0+2) share one transform at every
 * Normal site: radius from j0's draw word, angle from j1's; j0 takes the cosine, j1 the sine.  Every
 * operation is IEEE-exact.  A lane-0 key has no partner: its angle word comes from the same key under tag 'T'. */
#define O_TAG_TWIN 0x54u
/* The transform is table-driven (gjx_device.hpp bm_pair is the statement of the spec; tools/gen_bm_tables.py writes the
 * same constants into both files): log u through Cephes' mantissa reduction, a 64-entry (1/c, log c) table and the
 * log1p series to the 4th power; the angle through a 256-entry (cos, sin) table and the angle-sum formulas with
 * short series in the remainder.  Only fmaf / * / + on floats and one correctly rounded sqrtf. */
typedef struct { uint32_t a, b; } o_bm_ent;
/* BEGIN BM TABLES (generated: tools/gen_bm_tables.py) */
#define GJX_BM_LG_INIT { \
  {0x3fb4065bu, 0xbeaea002u}, {0x3fb21179u, 0xbea907acu}, {0x3fb0275bu, 0xbea37eceu}, {0x3fae47abu, 0xbe9e051au}, \
  {0x3fac7214u, 0xbe989a3au}, {0x3faaa645u, 0xbe933ddfu}, {0x3fa8e3f0u, 0xbe8defbau}, {0x3fa72acdu, 0xbe88af88u}, \
  {0x3fa57a92u, 0xbe837cf7u}, {0x3fa3d2fcu, 0xbe7caf8au}, {0x3fa233cau, 0xbe727f63u}, {0x3fa09cbbu, 0xbe6868e7u}, \
  {0x3f9f0d94u, 0xbe5e6ba2u}, {0x3f9d8619u, 0xbe54870cu}, {0x3f9c0613u, 0xbe4abab3u}, {0x3f9a8d4cu, 0xbe410622u}, \
  {0x3f991b90u, 0xbe3768e7u}, {0x3f97b0acu, 0xbe2de28cu}, {0x3f964c71u, 0xbe2472aeu}, {0x3f94eeb0u, 0xbe1b18dfu}, \
  {0x3f93973bu, 0xbe11d4b2u}, {0x3f9245e9u, 0xbe08a5d3u}, {0x3f90fa8fu, 0xbdff17adu}, {0x3f8fb505u, 0xbded0cc2u}, \
  {0x3f8e7525u, 0xbddb2a38u}, {0x3f8d3ac8u, 0xbdc96f46u}, {0x3f8c05cau, 0xbdb7db42u}, {0x3f8ad60au, 0xbda66d9eu}, \
  {0x3f89ab64u, 0xbd952594u}, {0x3f8885b8u, 0xbd84028du}, {0x3f8764e7u, 0xbd6607e9u}, {0x3f8648d1u, 0xbd445236u}, \
  {0x3f85315au, 0xbd22e305u}, {0x3f841e65u, 0xbd01b926u}, {0x3f830fd6u, 0xbcc1a6e7u}, {0x3f820593u, 0xbc8061deu}, \
  {0x3f80ff81u, 0xbbfe834fu}, {0x3f800000u, 0x80000000u}, {0x3f7c0629u, 0x3c803a74u}, {0x3f7834c1u, 0x3cfd47fdu}, \
  {0x3f748086u, 0x3d3c3a06u}, {0x3f70e82du, 0x3d78e68cu}, {0x3f6d6a80u, 0x3d9a582eu}, {0x3f6a065bu, 0x3db7cef9u}, \
  {0x3f66baaau, 0x3dd4dac6u}, {0x3f638667u, 0x3df17e9eu}, {0x3f60689du, 0x3e06dea8u}, {0x3f5d6061u, 0x3e14ccd8u}, \
  {0x3f5a6cd9u, 0x3e228b23u}, {0x3f578d31u, 0x3e301adbu}, {0x3f54c0a5u, 0x3e3d7d2fu}, {0x3f520677u, 0x3e4ab351u}, \
  {0x3f4f5df6u, 0x3e57be5cu}, {0x3f4cc677u, 0x3e649f6cu}, {0x3f4a3f5au, 0x3e715784u}, {0x3f47c805u, 0x3e7de7aau}, \
  {0x3f455fe6u, 0x3e85286bu}, {0x3f430672u, 0x3e8b49fau}, {0x3f40bb24u, 0x3e9158f8u}, {0x3f3e7d7fu, 0x3e9755d1u}, \
  {0x3f3c4d09u, 0x3e9d40f2u}, {0x3f3a294fu, 0x3ea31ac5u}, {0x3f3811e5u, 0x3ea8e3a8u}, {0x3f36065fu, 0x3eae9c03u}, \
}
#define GJX_BM_CS_INIT { \
  {0x3f7ffb11u, 0x3c490e90u}, {0x3f7fd397u, 0x3d16c32cu}, {0x3f7f84abu, 0x3d7b2b74u}, {0x3f7f0e58u, 0x3dafb680u}, \
  {0x3f7e70b0u, 0x3de1bc2eu}, {0x3f7dabccu, 0x3e09cf86u}, {0x3f7cbfc9u, 0x3e22abb6u}, {0x3f7baccdu, 0x3e3b6ecfu}, \
  {0x3f7a7302u, 0x3e541501u}, {0x3f791298u, 0x3e6c9a7fu}, {0x3f778bc5u, 0x3e827dc0u}, {0x3f75dec6u, 0x3e8e9a22u}, \
  {0x3f740bddu, 0x3e9aa086u}, {0x3f721352u, 0x3ea68f12u}, {0x3f6ff573u, 0x3eb263efu}, {0x3f6db293u, 0x3ebe1d4au}, \
  {0x3f6b4b0cu, 0x3ec9b953u}, {0x3f68bf3cu, 0x3ed53641u}, {0x3f660f88u, 0x3ee0924fu}, {0x3f633c5au, 0x3eebcbbbu}, \
  {0x3f604621u, 0x3ef6e0cbu}, {0x3f5d2d53u, 0x3f00e7e4u}, {0x3f59f26au, 0x3f064b82u}, {0x3f5695e5u, 0x3f0b9a6bu}, \
  {0x3f531849u, 0x3f10d3cdu}, {0x3f4f7a1fu, 0x3f15f6d9u}, {0x3f4bbbf8u, 0x3f1b02c6u}, {0x3f47de65u, 0x3f1ff6cbu}, \
  {0x3f43e200u, 0x3f24d225u}, {0x3f3fc767u, 0x3f299415u}, {0x3f3b8f3bu, 0x3f2e3bdeu}, {0x3f373a23u, 0x3f32c8c9u}, \
  {0x3f32c8c9u, 0x3f373a23u}, {0x3f2e3bdeu, 0x3f3b8f3bu}, {0x3f299415u, 0x3f3fc767u}, {0x3f24d225u, 0x3f43e200u}, \
  {0x3f1ff6cbu, 0x3f47de65u}, {0x3f1b02c6u, 0x3f4bbbf8u}, {0x3f15f6d9u, 0x3f4f7a1fu}, {0x3f10d3cdu, 0x3f531849u}, \
  {0x3f0b9a6bu, 0x3f5695e5u}, {0x3f064b82u, 0x3f59f26au}, {0x3f00e7e4u, 0x3f5d2d53u}, {0x3ef6e0cbu, 0x3f604621u}, \
  {0x3eebcbbbu, 0x3f633c5au}, {0x3ee0924fu, 0x3f660f88u}, {0x3ed53641u, 0x3f68bf3cu}, {0x3ec9b953u, 0x3f6b4b0cu}, \
  {0x3ebe1d4au, 0x3f6db293u}, {0x3eb263efu, 0x3f6ff573u}, {0x3ea68f12u, 0x3f721352u}, {0x3e9aa086u, 0x3f740bddu}, \
  {0x3e8e9a22u, 0x3f75dec6u}, {0x3e827dc0u, 0x3f778bc5u}, {0x3e6c9a7fu, 0x3f791298u}, {0x3e541501u, 0x3f7a7302u}, \
  {0x3e3b6ecfu, 0x3f7baccdu}, {0x3e22abb6u, 0x3f7cbfc9u}, {0x3e09cf86u, 0x3f7dabccu}, {0x3de1bc2eu, 0x3f7e70b0u}, \
  {0x3dafb680u, 0x3f7f0e58u}, {0x3d7b2b74u, 0x3f7f84abu}, {0x3d16c32cu, 0x3f7fd397u}, {0x3c490e90u, 0x3f7ffb11u}, \
  {0xbc490e90u, 0x3f7ffb11u}, {0xbd16c32cu, 0x3f7fd397u}, {0xbd7b2b74u, 0x3f7f84abu}, {0xbdafb680u, 0x3f7f0e58u}, \
  {0xbde1bc2eu, 0x3f7e70b0u}, {0xbe09cf86u, 0x3f7dabccu}, {0xbe22abb6u, 0x3f7cbfc9u}, {0xbe3b6ecfu, 0x3f7baccdu}, \
  {0xbe541501u, 0x3f7a7302u}, {0xbe6c9a7fu, 0x3f791298u}, {0xbe827dc0u, 0x3f778bc5u}, {0xbe8e9a22u, 0x3f75dec6u}, \
  {0xbe9aa086u, 0x3f740bddu}, {0xbea68f12u, 0x3f721352u}, {0xbeb263efu, 0x3f6ff573u}, {0xbebe1d4au, 0x3f6db293u}, \
  {0xbec9b953u, 0x3f6b4b0cu}, {0xbed53641u, 0x3f68bf3cu}, {0xbee0924fu, 0x3f660f88u}, {0xbeebcbbbu, 0x3f633c5au}, \
  {0xbef6e0cbu, 0x3f604621u}, {0xbf00e7e4u, 0x3f5d2d53u}, {0xbf064b82u, 0x3f59f26au}, {0xbf0b9a6bu, 0x3f5695e5u}, \
  {0xbf10d3cdu, 0x3f531849u}, {0xbf15f6d9u, 0x3f4f7a1fu}, {0xbf1b02c6u, 0x3f4bbbf8u}, {0xbf1ff6cbu, 0x3f47de65u}, \
  {0xbf24d225u, 0x3f43e200u}, {0xbf299415u, 0x3f3fc767u}, {0xbf2e3bdeu, 0x3f3b8f3bu}, {0xbf32c8c9u, 0x3f373a23u}, \
  {0xbf373a23u, 0x3f32c8c9u}, {0xbf3b8f3bu, 0x3f2e3bdeu}, {0xbf3fc767u, 0x3f299415u}, {0xbf43e200u, 0x3f24d225u}, \
  {0xbf47de65u, 0x3f1ff6cbu}, {0xbf4bbbf8u, 0x3f1b02c6u}, {0xbf4f7a1fu, 0x3f15f6d9u}, {0xbf531849u, 0x3f10d3cdu}, \
  {0xbf5695e5u, 0x3f0b9a6bu}, {0xbf59f26au, 0x3f064b82u}, {0xbf5d2d53u, 0x3f00e7e4u}, {0xbf604621u, 0x3ef6e0cbu}, \
  {0xbf633c5au, 0x3eebcbbbu}, {0xbf660f88u, 0x3ee0924fu}, {0xbf68bf3cu, 0x3ed53641u}, {0xbf6b4b0cu, 0x3ec9b953u}, \
  {0xbf6db293u, 0x3ebe1d4au}, {0xbf6ff573u, 0x3eb263efu}, {0xbf721352u, 0x3ea68f12u}, {0xbf740bddu, 0x3e9aa086u}, \
  {0xbf75dec6u, 0x3e8e9a22u}, {0xbf778bc5u, 0x3e827dc0u}, {0xbf791298u, 0x3e6c9a7fu}, {0xbf7a7302u, 0x3e541501u}, \
  {0xbf7baccdu, 0x3e3b6ecfu}, {0xbf7cbfc9u, 0x3e22abb6u}, {0xbf7dabccu, 0x3e09cf86u}, {0xbf7e70b0u, 0x3de1bc2eu}, \
  {0xbf7f0e58u, 0x3dafb680u}, {0xbf7f84abu, 0x3d7b2b74u}, {0xbf7fd397u, 0x3d16c32cu}, {0xbf7ffb11u, 0x3c490e90u}, \
  {0xbf7ffb11u, 0xbc490e90u}, {0xbf7fd397u, 0xbd16c32cu}, {0xbf7f84abu, 0xbd7b2b74u}, {0xbf7f0e58u, 0xbdafb680u}, \
  {0xbf7e70b0u, 0xbde1bc2eu}, {0xbf7dabccu, 0xbe09cf86u}, {0xbf7cbfc9u, 0xbe22abb6u}, {0xbf7baccdu, 0xbe3b6ecfu}, \
  {0xbf7a7302u, 0xbe541501u}, {0xbf791298u, 0xbe6c9a7fu}, {0xbf778bc5u, 0xbe827dc0u}, {0xbf75dec6u, 0xbe8e9a22u}, \
  {0xbf740bddu, 0xbe9aa086u}, {0xbf721352u, 0xbea68f12u}, {0xbf6ff573u, 0xbeb263efu}, {0xbf6db293u, 0xbebe1d4au}, \
  {0xbf6b4b0cu, 0xbec9b953u}, {0xbf68bf3cu, 0xbed53641u}, {0xbf660f88u, 0xbee0924fu}, {0xbf633c5au, 0xbeebcbbbu}, \
  {0xbf604621u, 0xbef6e0cbu}, {0xbf5d2d53u, 0xbf00e7e4u}, {0xbf59f26au, 0xbf064b82u}, {0xbf5695e5u, 0xbf0b9a6bu}, \
  {0xbf531849u, 0xbf10d3cdu}, {0xbf4f7a1fu, 0xbf15f6d9u}, {0xbf4bbbf8u, 0xbf1b02c6u}, {0xbf47de65u, 0xbf1ff6cbu}, \
  {0xbf43e200u, 0xbf24d225u}, {0xbf3fc767u, 0xbf299415u}, {0xbf3b8f3bu, 0xbf2e3bdeu}, {0xbf373a23u, 0xbf32c8c9u}, \
  {0xbf32c8c9u, 0xbf373a23u}, {0xbf2e3bdeu, 0xbf3b8f3bu}, {0xbf299415u, 0xbf3fc767u}, {0xbf24d225u, 0xbf43e200u}, \
  {0xbf1ff6cbu, 0xbf47de65u}, {0xbf1b02c6u, 0xbf4bbbf8u}, {0xbf15f6d9u, 0xbf4f7a1fu}, {0xbf10d3cdu, 0xbf531849u}, \
  {0xbf0b9a6bu, 0xbf5695e5u}, {0xbf064b82u, 0xbf59f26au}, {0xbf00e7e4u, 0xbf5d2d53u}, {0xbef6e0cbu, 0xbf604621u}, \
  {0xbeebcbbbu, 0xbf633c5au}, {0xbee0924fu, 0xbf660f88u}, {0xbed53641u, 0xbf68bf3cu}, {0xbec9b953u, 0xbf6b4b0cu}, \
  {0xbebe1d4au, 0xbf6db293u}, {0xbeb263efu, 0xbf6ff573u}, {0xbea68f12u, 0xbf721352u}, {0xbe9aa086u, 0xbf740bddu}, \
  {0xbe8e9a22u, 0xbf75dec6u}, {0xbe827dc0u, 0xbf778bc5u}, {0xbe6c9a7fu, 0xbf791298u}, {0xbe541501u, 0xbf7a7302u}, \
  {0xbe3b6ecfu, 0xbf7baccdu}, {0xbe22abb6u, 0xbf7cbfc9u}, {0xbe09cf86u, 0xbf7dabccu}, {0xbde1bc2eu, 0xbf7e70b0u}, \
  {0xbdafb680u, 0xbf7f0e58u}, {0xbd7b2b74u, 0xbf7f84abu}, {0xbd16c32cu, 0xbf7fd397u}, {0xbc490e90u, 0xbf7ffb11u}, \
  {0x3c490e90u, 0xbf7ffb11u}, {0x3d16c32cu, 0xbf7fd397u}, {0x3d7b2b74u, 0xbf7f84abu}, {0x3dafb680u, 0xbf7f0e58u}, \
  {0x3de1bc2eu, 0xbf7e70b0u}, {0x3e09cf86u, 0xbf7dabccu}, {0x3e22abb6u, 0xbf7cbfc9u}, {0x3e3b6ecfu, 0xbf7baccdu}, \
  {0x3e541501u, 0xbf7a7302u}, {0x3e6c9a7fu, 0xbf791298u}, {0x3e827dc0u, 0xbf778bc5u}, {0x3e8e9a22u, 0xbf75dec6u}, \
  {0x3e9aa086u, 0xbf740bddu}, {0x3ea68f12u, 0xbf721352u}, {0x3eb263efu, 0xbf6ff573u}, {0x3ebe1d4au, 0xbf6db293u}, \
  {0x3ec9b953u, 0xbf6b4b0cu}, {0x3ed53641u, 0xbf68bf3cu}, {0x3ee0924fu, 0xbf660f88u}, {0x3eebcbbbu, 0xbf633c5au}, \
  {0x3ef6e0cbu, 0xbf604621u}, {0x3f00e7e4u, 0xbf5d2d53u}, {0x3f064b82u, 0xbf59f26au}, {0x3f0b9a6bu, 0xbf5695e5u}, \
  {0x3f10d3cdu, 0xbf531849u}, {0x3f15f6d9u, 0xbf4f7a1fu}, {0x3f1b02c6u, 0xbf4bbbf8u}, {0x3f1ff6cbu, 0xbf47de65u}, \
  {0x3f24d225u, 0xbf43e200u}, {0x3f299415u, 0xbf3fc767u}, {0x3f2e3bdeu, 0xbf3b8f3bu}, {0x3f32c8c9u, 0xbf373a23u}, \
  {0x3f373a23u, 0xbf32c8c9u}, {0x3f3b8f3bu, 0xbf2e3bdeu}, {0x3f3fc767u, 0xbf299415u}, {0x3f43e200u, 0xbf24d225u}, \
  {0x3f47de65u, 0xbf1ff6cbu}, {0x3f4bbbf8u, 0xbf1b02c6u}, {0x3f4f7a1fu, 0xbf15f6d9u}, {0x3f531849u, 0xbf10d3cdu}, \
  {0x3f5695e5u, 0xbf0b9a6bu}, {0x3f59f26au, 0xbf064b82u}, {0x3f5d2d53u, 0xbf00e7e4u}, {0x3f604621u, 0xbef6e0cbu}, \
  {0x3f633c5au, 0xbeebcbbbu}, {0x3f660f88u, 0xbee0924fu}, {0x3f68bf3cu, 0xbed53641u}, {0x3f6b4b0cu, 0xbec9b953u}, \
  {0x3f6db293u, 0xbebe1d4au}, {0x3f6ff573u, 0xbeb263efu}, {0x3f721352u, 0xbea68f12u}, {0x3f740bddu, 0xbe9aa086u}, \
  {0x3f75dec6u, 0xbe8e9a22u}, {0x3f778bc5u, 0xbe827dc0u}, {0x3f791298u, 0xbe6c9a7fu}, {0x3f7a7302u, 0xbe541501u}, \
  {0x3f7baccdu, 0xbe3b6ecfu}, {0x3f7cbfc9u, 0xbe22abb6u}, {0x3f7dabccu, 0xbe09cf86u}, {0x3f7e70b0u, 0xbde1bc2eu}, \
  {0x3f7f0e58u, 0xbdafb680u}, {0x3f7f84abu, 0xbd7b2b74u}, {0x3f7fd397u, 0xbd16c32cu}, {0x3f7ffb11u, 0xbc490e90u}, \
}
/* END BM TABLES */
static const o_bm_ent o_bm_lg[64] = GJX_BM_LG_INIT;
static const o_bm_ent o_bm_cs[256] = GJX_BM_CS_INIT;
static inline void o_bm_pair(uint32_t w_radius, uint32_t w_angle, float* z_cos, float* z_sin) {
  float u = ((float)w_radius + 1.0f) * 2.3283064365386963e-10f;
  uint32_t t = o_f2u(u) - 0x3f3504f3u;
  o_bm_ent lg = o_bm_lg[(t >> 17) & 63u];
  float m = o_u2f((t & 0x007fffffu) + 0x3f3504f3u);
  float fe = (float)((int32_t)t >> 23);
  float q = fmaf(m, o_u2f(lg.a), -1.0f);
  float p = fmaf(q, -0.25f, 0.333333343f);
  p = fmaf(p, q, -0.5f);
  float qq = q * q;
  float l1 = fmaf(p, qq, q);
  float lu = fmaf(fe, 0.693147182f, o_u2f(lg.b)) + l1;
  float r = sqrtf(-2.0f * lu);
  o_bm_ent cs = o_bm_cs[w_angle >> 24];
  float d = fmaf((float)((w_angle >> 8) & 0xffffu), 3.74507035e-07f, -0.0122718466f);
  float d2 = d * d;
  float d3 = d2 * d;
  float sd = fmaf(d3, -0.166666672f, d);
  float cd = fmaf(d2, fmaf(d2, 0.0416666679f, -0.5f), 1.0f);
  float ca = o_u2f(cs.a), sa = o_u2f(cs.b);
  float t1 = sa * sd, t2 = ca * sd;
  *z_cos = r * fmaf(ca, cd, -t1);
  *z_sin = r * fmaf(sa, cd, t2);
}
/* The standard normal of a Normal site for stream `s` (folded, PHILOX): Box-Muller over the particle pair — the
 * even particle's draw is the radius word, the odd one's the angle word, the even particle takes the cosine
 * branch; a lane-0 key has no partner and takes its angle word from its twin block ('T').  THREEFRY / unfolded
 * streams keep jax's erfinv form. */
static inline float o_site_normal(const o_stream* s) {
  if (s->impl == 0 || !s->hf) return o_std_normal(o_bits32_at(s, 0));
  float zc, zs;
  if ((s->k[2] | s->k[3]) == 0u) {
    uint32_t o[4];
    o_philox_lane(s->k, s->f >> 2, O_TAG_TWIN, o);
    o_bm_pair(o_single_draw(s->k, s->f), o[s->f & 3u], &zc, &zs);
    return zc;
  }
  uint64_t j = (((uint64_t)s->k[3] << 32) | s->k[2]) - 1u;
  uint64_t la = (j & ~(uint64_t)1) + 1u, lb = la + 1u;
  uint32_t ka[4] = {s->k[0], s->k[1], (uint32_t)la, (uint32_t)(la >> 32)};
  uint32_t kb[4] = {s->k[0], s->k[1], (uint32_t)lb, (uint32_t)(lb >> 32)};
  o_bm_pair(o_single_draw(ka, s->f), o_single_draw(kb, s->f), &zc, &zs);
  return (j & 1u) ? zs : zc;
}

/* The standard normal of SMC slot j (LGSSM filter).  THREEFRY: erfinv of the slot's draw.  PHILOX: Box-Muller
 * over the slot's pair inside its quad — slots (4g, 4g+1) from words (0, 1), slots (4g+2, 4g+3) from words
 * (2, 3); the even slot takes the cosine branch. */
static inline float o_smc_quad_normal(const uint32_t step_key[4], uint64_t j, uint32_t f) { /* PHILOX */
  const uint64_t even = j & ~(uint64_t)1;
  float zc, zs;
  o_bm_pair(o_smc_quad_word(step_key, even, f), o_smc_quad_word(step_key, even + 1u, f), &zc, &zs);
  return (j & 1u) ? zs : zc;
}
static inline float o_smc_slot_normal(int impl, const uint32_t step_key[4], uint64_t j) {
  if (impl == 0) return o_std_normal(o_smc_slot_bits(impl, step_key, j));
  return o_smc_quad_normal(step_key, j, 0u);
}

/* ---------------- log-densities (TFP formulas) ----------------------------- */

static inline float o_logpdf_normal(float x, float loc, float scale) {
  float rs = 1.0f / scale;
  float lognorm = 0.91893853320467f + o_log(scale);
  float d = x * rs - loc * rs;
  return (-0.5f * d) * d - lognorm;
}
static inline float o_xlogy(float a, float y) { return a == 0.0f ? 0.0f : a * o_log(y); }
static inline float o_logpdf_gamma(float x, float conc, float rate) {
  float lognorm = o_lgamma(conc) - conc * o_log(rate);
  return (o_xlogy(conc - 1.0f, x) - rate * x) - lognorm;
}
static inline float o_logpdf_beta(float x, float a, float b) {
  float lbeta = (o_lgamma(a) + o_lgamma(b)) - o_lgamma(a + b);
  return (o_xlogy(a - 1.0f, x) + o_xlogy(b - 1.0f, 1.0f - x)) - lbeta;
}
static inline float o_logpdf_bernoulli(int e, float p) {
  return e ? o_log(p) : o_log(1.0f - p);
}

/* ---------------- samplers ------------------------------------------------------------------ */

/* Marsaglia & Tsang (2000) Gamma(conc, 1) on sub-streams of `key`: attempt a of gamma `which`
 * (0/1; Beta draws two) uses words(sub = 1 + 2a + which): w0 -> normal, w1 -> uniform.  The
 * conc < 1 boost uniform is word `which` of sub 0.  At most 64 attempts (then accept). */
static inline float o_std_gamma(const o_stream* st, int which, float conc) {
  int boost = conc < 1.0f;
  float a = boost ? conc + 1.0f : conc;
  float d = a - 0.33333334f;
  float c = 1.0f / sqrtf(9.0f * d);
  float v = 1.0f;
  for (int att = 0; att < 64; ++att) {
    uint32_t w0, w1;
    o_words(st, (uint32_t)(1 + 2 * att + which), &w0, &w1);
    float x = o_std_normal(w0);
    float t = 1.0f + c * x;
    if (t <= 0.0f) continue;
    v = (t * t) * t;
    float u = o_uniform01(w1);
    float rhs = (0.5f * x) * x + d;
    rhs = rhs - d * v;
    rhs = rhs + d * o_log(v);
    if (o_log(u) < rhs) break;
  }
  float g = d * v;
  if (boost) {
    uint32_t w0, w1;
    o_words(st, 0u, &w0, &w1);
    float ub = o_uniform01(which ? w1 : w0);
    g = g * o_exp(o_log(ub) / conc);
  }
  return g;
}

/* ---------------- fixed-point weights ------------------------------------ */
static inline int o_frac_bits(uint64_t n_total) {
  int lg = 0;
  while (lg < 63 && ((uint64_t)1 << lg) < n_total) ++lg;
  int f = 62 - lg;
  return f > 40 ? 40 : (f < 8 ? 8 : f);
}
static inline uint64_t o_fixw(float lw, float m, int frac) {
  if (lw == m) return (uint64_t)1 << frac;
  float d = lw - m;
  if (!(d >= -80.0f)) return 0;
  float s = o_exp(d);
  float t = s * o_u2f((uint32_t)(127 + frac) << 23); /* * 2^frac, exact */
  return (uint64_t)rintf(t);
}

/* ---------------- row-anchored fixed point ------------------------------- */
#define O_ROW_FRAC 30
#define O_ROW_EMPTY (-(1 << 30))
static inline int32_t o_row_anchor(float m) {
  if (!(m > -INFINITY)) return O_ROW_EMPTY;
  float t = m * 1.44269504088896341f;
  t = t > 16777216.0f ? 16777216.0f : (t < -16777216.0f ? -16777216.0f : t);
  return (int32_t)ceilf(t);
}
static inline uint64_t o_rowfix(float lw, int32_t e) {
  if (e == O_ROW_EMPTY || !(lw > -INFINITY)) return 0;
  float fe = (float)e;
  float d = fmaf(-fe, 0.693359375f, lw);
  d = fmaf(-fe, -2.12194440e-4f, d);
  if (d > 1.0f) d = 1.0f; /* only a weight beyond the clamped anchor (+inf, > 1.1e7): keeps the conversion defined */
  return (uint64_t)rintf(o_exp(d) * 1073741824.0f);
}

#endif
