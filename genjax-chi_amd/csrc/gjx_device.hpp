// gjx_device.hpp — gfx950 device-side building blocks: counter-based RNG, the bit-exact f32 math
// specification, samplers / log-densities, and 64-wide wavefront reductions/scans.
//
// Everything numeric here uses only IEEE-exact primitives (+ - * fma / sqrt rint, integer ops)
// in a fixed order; the translation unit is compiled with -ffp-contract=off so nothing is
// re-associated or fused behind our back.  That is what makes ancestor indices and log-weights
// bit-identical to the CPU oracle on identical counters.
//
// Reference call sites this arithmetic replaces (relative to /root/reference/src/genjax/_src):
//   jax.random.split/fold_in ........ inference/smc.py:299-300, generative_functions/static.py:349-352
//   tfd.X(...).sample / .log_prob ... generative_functions/distributions/tensorflow_probability/__init__.py:52-62
//   logsumexp ....................... inference/smc.py:97,107
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdint.h>
#else  // hiprtc (plan specialisation): HIP built-ins are pre-included, fixed-width types are not
typedef unsigned char uint8_t;
typedef unsigned short uint16_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
typedef unsigned long uintptr_t;
typedef unsigned long size_t;
#endif

#define GJX_DEV __device__ __forceinline__
#define GJX_HD __host__ __device__ __forceinline__

namespace gjx {

constexpr int kWave = 64;         // CDNA wavefront
constexpr int kBlock = 256;       // threads per workgroup (4 waves, one per SIMD)
#ifndef GJX_TILE
#define GJX_TILE 1024
#endif
constexpr int kTile = GJX_TILE;         // particles per workgroup tile
constexpr int kPer = kTile / kBlock;    // ... = particles per thread
constexpr int kCatFrac = 23;      // fixed-point bits of per-row categorical CDFs

GJX_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
GJX_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// ------------------------------------------------------------------------------------------------
// Ciphers (Salmon et al., SC'11).  Threefry2x32-20 is jax.random's default; Philox4x32-10 is the
// native scheme named by the north star.
// ------------------------------------------------------------------------------------------------
GJX_HD uint32_t rotl32(uint32_t x, uint32_t r) { return (x << r) | (x >> (32u - r)); }

GJX_HD void threefry2x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t& o0,
                         uint32_t& o1) {
  const uint32_t k2 = 0x1BD11BDAu ^ k0 ^ k1;
  uint32_t x0 = c0 + k0, x1 = c1 + k1;
#define GJX_TF_R(r) x0 += x1; x1 = rotl32(x1, r); x1 ^= x0;
  GJX_TF_R(13) GJX_TF_R(15) GJX_TF_R(26) GJX_TF_R(6)
  x0 += k1; x1 += k2 + 1u;
  GJX_TF_R(17) GJX_TF_R(29) GJX_TF_R(16) GJX_TF_R(24)
  x0 += k2; x1 += k0 + 2u;
  GJX_TF_R(13) GJX_TF_R(15) GJX_TF_R(26) GJX_TF_R(6)
  x0 += k0; x1 += k1 + 3u;
  GJX_TF_R(17) GJX_TF_R(29) GJX_TF_R(16) GJX_TF_R(24)
  x0 += k1; x1 += k2 + 4u;
  GJX_TF_R(13) GJX_TF_R(15) GJX_TF_R(26) GJX_TF_R(6)
  x0 += k2; x1 += k0 + 5u;
#undef GJX_TF_R
  o0 = x0;
  o1 = x1;
}

GJX_HD void philox4x32(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2,
                       uint32_t c3, uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;  // v_mad_u64_u32: hi and lo in one op
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

// Philox domain tags (low byte of counter word 3; the upper 24 bits carry the site fold of a stream).
constexpr uint32_t kTagSplit = 0x53u;   // 'S'  split of a laned key
constexpr uint32_t kTagFold = 0x46u;    // 'F'  fold_in
constexpr uint32_t kTagDraw = 0x44u;    // 'D'  packed single-word draws
constexpr uint32_t kTagStream = 0x52u;  // 'R'  sub-streams of multi-word samplers

// A key.  THREEFRY (jax semantics) uses k0,k1 only.  PHILOX keys carry a 64-bit LANE next to the
// 64-bit cipher key: counter words 0,1 of every block are the lane, so the n children of a lane-0
// key are (same cipher key, lane i+1) — a population's keys cost no cipher block and share one
// wave-uniform cipher key (the round keys live in scalar registers).
struct Key {
  uint32_t k0, k1;
  uint32_t l0, l1;  // PHILOX lane (0 = none)
};
GJX_HD Key make_key(uint32_t k0, uint32_t k1) { return Key{k0, k1, 0u, 0u}; }

// split(parent, *)[i]
template <int IMPL>
GJX_HD Key split_at(Key parent, uint64_t i) {
  Key out;
  out.l0 = 0u;
  out.l1 = 0u;
  if (IMPL == 0) {
    threefry2x32(parent.k0, parent.k1, (uint32_t)(i >> 32), (uint32_t)i, out.k0, out.k1);
  } else if ((parent.l0 | parent.l1) == 0u) {
    const uint64_t lane = i + 1u;
    out = Key{parent.k0, parent.k1, (uint32_t)lane, (uint32_t)(lane >> 32)};
  } else {
    uint32_t o2, o3;
    philox4x32(parent.k0, parent.k1, parent.l0, parent.l1, (uint32_t)i,
               ((uint32_t)(i >> 32) << 8) | kTagSplit, out.k0, out.k1, o2, o3);
  }
  return out;
}
template <int IMPL>
GJX_HD Key fold_in(Key k, uint32_t d) {
  Key out;
  out.l0 = 0u;
  out.l1 = 0u;
  if (IMPL == 0) {
    threefry2x32(k.k0, k.k1, 0u, d, out.k0, out.k1);
  } else {
    uint32_t o2, o3;
    philox4x32(k.k0, k.k1, k.l0, k.l1, d, kTagFold, out.k0, out.k1, o2, o3);
  }
  return out;
}

// The key of step t of a Scan (gjx_scan_run; combinators.py Scan._run).  THREEFRY keeps jax's chain, key_t =
// fold_in(key_{t-1}, t) (scan.py:267-268, 276).  PHILOX needs no cipher block for a key: step t of particle lane L draws
// under (the same cipher key, lane L + (t + 1) 2^40) — lanes below 2^40, fewer than 2^24 - 1 steps — so a step's keys
// are laned keys like a population's: particle pairs share their blocks and the Box-Muller transform of a Normal site.
GJX_HD Key scan_step_key_philox(Key k, uint32_t t) { return Key{k.k0, k.k1, k.l0, k.l1 + ((t + 1u) << 8)}; }

// PHILOX single-word draw number f of a key (f = 0-based index of the site among the body's sampled sites).
//  * lane-0 key (a lone key): four draws per block of its own, word f & 3 of PH(ctr = (0, 0, f >> 2, 'D'), key);
//  * laned key (lane L >= 1 = particle i = L - 1 among its parent's children): the PAIR (i, i ^ 1) shares its
//    blocks — PH(ctr = (p_lo, p_hi, f >> 1, 'P'), key), p = i >> 1, holds draws 2 (f >> 1) and 2 (f >> 1) + 1 of
//    both particles, particle i taking word ((f & 1) << 1) | (i & 1).  A kernel that owns both particles of a
//    pair computes S / 2 blocks per particle for S draws, and a Normal site's Box-Muller pair (even particle:
//    radius word, odd: angle word) sits in one block.
constexpr uint32_t kTagPair = 0x50u;  // 'P'
GJX_HD void philox_pair_block(Key k, uint32_t blk, uint32_t (&o)[4]) {  // lane >= 1
  const uint64_t p = ((((uint64_t)k.l1 << 32) | k.l0) - 1u) >> 1;
  philox4x32(k.k0, k.k1, (uint32_t)p, (uint32_t)(p >> 32), blk, kTagPair, o[0], o[1], o[2], o[3]);
}
GJX_HD uint32_t philox_single_draw(Key k, uint32_t f) {
  uint32_t o[4];
  uint32_t sel;
  if ((k.l0 | k.l1) == 0u) {
    philox4x32(k.k0, k.k1, 0u, 0u, f >> 2, kTagDraw, o[0], o[1], o[2], o[3]);
    sel = f & 3u;
  } else {
    philox_pair_block(k, f >> 1, o);
    sel = ((f & 1u) << 1) | ((k.l0 - 1u) & 1u);
  }
  return sel == 0 ? o[0] : (sel == 1 ? o[1] : (sel == 2 ? o[2] : o[3]));
}

// A draw stream: key plus optional leaf-site fold.  THREEFRY folds the counter into the key (one
// block, jax semantics: fold = site counter from 1).  PHILOX carries it in the 128-bit counter (no
// extra block; fold = 0-based index of the site among the body's randomness-consuming sites).
template <int IMPL>
struct Stream {
  Key k;
  uint32_t f, hf;
  GJX_HD Stream(Key key, bool has_fold, uint32_t fold) {
    if (IMPL == 0) {
      k = has_fold ? fold_in<0>(key, fold) : key;
      f = 0u;
      hf = 0u;
    } else {
      k = key;
      f = has_fold ? fold : 0u;
      hf = has_fold ? 1u : 0u;
    }
  }
  // words 0,1 of sub-stream `sub` (multi-word samplers)
  GJX_HD void words(uint32_t sub, uint32_t& w0, uint32_t& w1) const {
    if (IMPL == 0) {
      threefry2x32(k.k0, k.k1, 0u, sub, w0, w1);
    } else {
      uint32_t o2, o3;
      philox4x32(k.k0, k.k1, k.l0, k.l1, sub, ((hf ? f + 1u : 0u) << 8) | kTagStream, w0, w1, o2, o3);
    }
  }
  // 32 bits of sub-stream `sub`.  PHILOX packs the single-word draws (sub 0) of a folded stream several to a
  // block (philox_single_draw).
  GJX_HD uint32_t bits32(uint32_t sub) const {
    uint32_t w0, w1;
    if (IMPL == 1 && hf && sub == 0u) return philox_single_draw(k, f);
    words(sub, w0, w1);
    return IMPL == 0 ? (w0 ^ w1) : w0;
  }
  GJX_HD uint64_t bits64(uint32_t sub) const {
    uint32_t w0, w1;
    words(sub, w0, w1);
    return ((uint64_t)w0 << 32) | w1;
  }
};

// Several independent passes in one launch (the algorithm vmapped over keys): pass p draws from the lazy
// children of its own parent key and writes p * pass_stride elements further in every output column
// (p * row_stride entries further in the per-row arrays).  One 1e6-particle pass is 7.8k waves — less than two
// rounds of the machine — so a launch of several passes keeps the SIMDs full through what would otherwise
// be each pass's fill and drain.
constexpr int kMaxPasses = 32;
struct PassBatch {
  uint32_t n_pass;         // >= 1
  uint32_t rows_per_pass;  // ceil(n / 256)
  uint64_t pass_stride;
  uint64_t row_stride;
  uint32_t parent[kMaxPasses][2];  // used when n_pass > 1 (lazy batches of lane-0 parents)
};

// Device tables a generated kernel reads (categorical logits, the per-row tables derived from them, ARG_TABLE columns):
// a kernel argument, not literals in the source — the source of a plan is keyed by its STRUCTURE, so another tensor of the
// same shape (a new transition matrix) costs no compilation (gjx_plan_jit.hpp: TableReg).
constexpr int kMaxPlanTables = 48;
struct PlanTables {
  const void* p[kMaxPlanTables];
};

// Launch-uniform parameters of an importance plan (GJX_ARG_PARAM; kernel argument, by value: they live in scalar
// registers): p[] the caller's values (gjx_plan_set_params), d[2 q], d[2 q + 1] the per-site constants derived from them
// on the host by the spec functions (normal: 1 / scale, log normaliser; gamma / beta: -, log normaliser).
constexpr int kMaxParams = 64;
struct PlanParams {
  float p[kMaxParams];
  float d[2 * 64];
};

// Launch arguments of an importance run over a Scan model (gjx_scan_run; kernel argument, by value).
struct ScanArgs {
  const float* obs;  // [T, n_obs]
  uint64_t n;
  uint64_t col_stride;
  int32_t n_steps;
  float carry0[4];
  const float* carry0_cols[4];
  float* carry_out[4];
};

// Column pointer table of one importance run (kernel argument, by value).
struct RunCols {
  const float* in[16];
  void* out[64];
};

// The 32-bit draw of SMC slot j at one step of the fixed-model filters (one latent site per step).
// THREEFRY keeps the jax shape: the first single-word draw of the slot's key split(step_key)[j] (split,
// fold_in(., 1), bits: 3 blocks per slot).  PHILOX: step keys have lane 0, and output slots 4g .. 4g+3 share
// ONE block, PH(ctr = (g_lo, g_hi, 0, 'Q'), key = step_key): slot j takes word j & 3 — a quarter of a block per
// particle-step, the cipher key uniform over the launch.  The resampling kernel gives each lane four
// consecutive slots, i.e. exactly one block.
template <int IMPL>
GJX_HD Key slot_key(Key step_key, uint64_t j) {  // split(step_key, *)[j]; step keys have lane 0
  if (IMPL == 0) return split_at<0>(step_key, j);
  const uint64_t lane = j + 1u;
  return Key{step_key.k0, step_key.k1, (uint32_t)lane, (uint32_t)(lane >> 32)};
}
constexpr uint32_t kTagQuad = 0x51u;  // 'Q': the shared block of four SMC slots
template <int IMPL>
GJX_HD void smc_quad_bits(Key step_key, uint64_t g, uint32_t (&w)[4]) {  // the draws of slots 4g .. 4g+3
  if (IMPL == 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const Stream<IMPL> st(slot_key<IMPL>(step_key, 4u * g + (uint64_t)u), true, 1u);
      w[u] = st.bits32(0);
    }
  } else {
    philox4x32(step_key.k0, step_key.k1, (uint32_t)g, (uint32_t)(g >> 32), 0u, kTagQuad, w[0], w[1], w[2], w[3]);
  }
}
template <int IMPL>
GJX_HD uint32_t smc_slot_bits(Key step_key, uint64_t j) {  // one slot on its own (init kernels' ragged edges, oracle)
  if (IMPL == 0) {
    const Stream<IMPL> st(slot_key<IMPL>(step_key, j), true, 1u);
    return st.bits32(0);
  }
  uint32_t w[4];
  smc_quad_bits<IMPL>(step_key, j >> 2, w);
  const uint32_t sel = (uint32_t)j & 3u;
  return sel == 0 ? w[0] : (sel == 1 ? w[1] : (sel == 2 ? w[2] : w[3]));
}

// Host-visible description of a key batch (mirrors gjx_keys).
struct KeySrc {
  const uint32_t* keys;  // mode 0: [n, 2] (threefry) / [n, 4] (philox)
  Key parent;            // modes 1, 2 (with its lane)
  uint64_t first;
  int mode;
  int has_fold;
  uint32_t fold;
};
template <int IMPL>
GJX_DEV Key key_at(const KeySrc& s, uint64_t i) {
  if (s.mode == 0) {
    if (IMPL == 0) {
      const uint2 v = reinterpret_cast<const uint2*>(s.keys)[i];
      return make_key(v.x, v.y);
    }
    const uint4 v = reinterpret_cast<const uint4*>(s.keys)[i];
    return Key{v.x, v.y, v.z, v.w};
  }
  if (s.mode == 2) return s.parent;
  return split_at<IMPL>(s.parent, s.first + i);
}
template <int IMPL>
GJX_DEV void store_key(uint32_t* out, uint64_t i, Key k) {
  if (IMPL == 0) reinterpret_cast<uint2*>(out)[i] = make_uint2(k.k0, k.k1);
  else reinterpret_cast<uint4*>(out)[i] = make_uint4(k.k0, k.k1, k.l0, k.l1);
}

// A value the optimiser must treat as unknown (generated kernels: constants whose folding would put a NaN / +inf
// CONSTANT log-weight into the kernel — invalid parameters, observations outside the support — are kept out of constant
// propagation: this toolchain's backend dies on such kernels, "SmallVector unable to grow", taking the process with it
// when the compiler runs in-process).  No instruction: an empty asm with a register constraint.
#if defined(__HIP_DEVICE_COMPILE__)
GJX_DEV float opq(float x) {
  asm volatile("" : "+v"(x));
  return x;
}
#else
GJX_HD float opq(float x) { return x; }
#endif

// ------------------------------------------------------------------------------------------------
// f32 math spec.  Cephes logf/expf coefficients; Giles' single-precision erfinv.
// ------------------------------------------------------------------------------------------------
GJX_HD float m_log(float x) {
  uint32_t ix = f2u(x);
  int32_t e = 0;
  if (ix - 1u >= 0x7f7fffffu) {  // +-0, +inf, NaN, every negative: log's own values (one compare on the common path)
    if ((ix << 1) == 0u) return -__builtin_inff();
    return ix == 0x7f800000u ? x : u2f(0x7fc00000u);
  }
  if (ix < 0x00800000u) {
    x = x * 8388608.0f;
    ix = f2u(x);
    e = -23;
  }
  const uint32_t t = ix - 0x3f3504f3u;
  e += (int32_t)t >> 23;
  const float m = u2f((t & 0x007fffffu) + 0x3f3504f3u);
  const float f = m - 1.0f;
  const float z = f * f;
  float p = 7.0376836292E-2f;
  p = __builtin_fmaf(p, f, -1.1514610310E-1f);
  p = __builtin_fmaf(p, f, 1.1676998740E-1f);
  p = __builtin_fmaf(p, f, -1.2420140846E-1f);
  p = __builtin_fmaf(p, f, 1.4249322787E-1f);
  p = __builtin_fmaf(p, f, -1.6668057665E-1f);
  p = __builtin_fmaf(p, f, 2.0000714765E-1f);
  p = __builtin_fmaf(p, f, -2.4999993993E-1f);
  p = __builtin_fmaf(p, f, 3.3333331174E-1f);
  float y = (p * f) * z;
  const float fe = (float)e;
  y = __builtin_fmaf(fe, -2.12194440e-4f, y);
  y = __builtin_fmaf(-0.5f, z, y);
  float r = f + y;
  r = __builtin_fmaf(fe, 0.693359375f, r);
  return r;
}

// m_log for positive NORMAL floats: the same operations minus the zero / subnormal pre-scaling, hence
// the same bits.  erfinv's argument (1-x)(1+x) lies in [2^-24, 1].
GJX_HD float m_log_normal(float x) {
  const uint32_t t = f2u(x) - 0x3f3504f3u;
  const int32_t e = (int32_t)t >> 23;
  const float m = u2f((t & 0x007fffffu) + 0x3f3504f3u);
  const float f = m - 1.0f;
  const float z = f * f;
  float p = 7.0376836292E-2f;
  p = __builtin_fmaf(p, f, -1.1514610310E-1f);
  p = __builtin_fmaf(p, f, 1.1676998740E-1f);
  p = __builtin_fmaf(p, f, -1.2420140846E-1f);
  p = __builtin_fmaf(p, f, 1.4249322787E-1f);
  p = __builtin_fmaf(p, f, -1.6668057665E-1f);
  p = __builtin_fmaf(p, f, 2.0000714765E-1f);
  p = __builtin_fmaf(p, f, -2.4999993993E-1f);
  p = __builtin_fmaf(p, f, 3.3333331174E-1f);
  float y = (p * f) * z;
  const float fe = (float)e;
  y = __builtin_fmaf(fe, -2.12194440e-4f, y);
  y = __builtin_fmaf(-0.5f, z, y);
  float r = f + y;
  r = __builtin_fmaf(fe, 0.693359375f, r);
  return r;
}

GJX_HD float m_exp_core(float x);
GJX_HD float m_exp(float x) {
  if (!(x >= -86.0f)) return 0.0f;
  if (x > 88.0f) x = 88.0f;
  return m_exp_core(x);
}
// (the arithmetic of m_exp for -86 <= x <= 88: callers that have already clamped select around it without a branch)
GJX_HD float m_exp_core(float x) {
  const float fx = __builtin_rintf(x * 1.44269504088896341f);
  x = __builtin_fmaf(fx, -0.693359375f, x);
  x = __builtin_fmaf(fx, 2.12194440e-4f, x);
  const float z = x * x;
  float p = 1.9875691500E-4f;
  p = __builtin_fmaf(p, x, 1.3981999507E-3f);
  p = __builtin_fmaf(p, x, 8.3334519073E-3f);
  p = __builtin_fmaf(p, x, 4.1665795894E-2f);
  p = __builtin_fmaf(p, x, 1.6666665459E-1f);
  p = __builtin_fmaf(p, x, 5.0000001201E-1f);
  const float y = __builtin_fmaf(p, z, x) + 1.0f;
  const int32_t n = (int32_t)fx;
  return u2f(f2u(y) + ((uint32_t)n << 23));
}

// exp as a model body writes it (GJX_EXPR_EXP, gjx_map_f32): m_exp over its whole domain — NaN stays NaN, the top of the
// range (m_exp clamps at 88, flushes below -86) goes through exp(x / 2)^2: +inf / subnormals / 0 where exp has them.
GJX_HD float e_exp(float x) {
  if (x != x) return x;
  if (x > 88.0f || x < -86.0f) {  // (below -86 m_exp flushes to 0: the square reaches the subnormals and then 0 as exp does)
    const float h = m_exp(x * 0.5f);
    return h * h;
  }
  return m_exp(x);
}

// max / min as a model body writes them (GJX_EXPR_MAX / _MIN; torch.maximum / jnp.maximum): a NaN if either argument is one.
GJX_HD float e_max(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }
GJX_HD float e_min(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }

// --- opt-in FAST math for importance plans (gjx.h: GJX_PLAN_FAST_MATH).  The north star asks for log-weights within
// 1e-5 relative of the reference on a path WITHOUT resampling, so an importance plan may trade the bit-exact
// polynomials for the hardware transcendentals (v_log_f32 / v_exp_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32: ~1 ulp,
// an 8-cycle issue each against 24-40 dependent fma's) wherever the result is a CONTINUOUS function of its input: the
// Box-Muller transform of Normal sites, the transcendental terms of log-densities, and the row-anchored weight sums.
// Everything that DECIDES something (rejection tests of the gamma sampler, categorical CDFs, Bernoulli thresholds,
// resampling weights) keeps the exact functions, so a fast plan draws the same particles as the exact plan up to
// rounding.  d_log / d_exp are the density-side functions; the exact build maps them to m_log / m_exp.
#if defined(GJX_FAST_MATH) && defined(__HIP_DEVICE_COMPILE__)
#define GJX_FAST_MATH_DEVICE 1
GJX_HD float d_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994531f; }
GJX_HD float d_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
GJX_HD float d_exp_core(float x) { return d_exp(x); }
#else
GJX_HD float d_log(float x) { return m_log(x); }
GJX_HD float d_exp(float x) { return m_exp(x); }
GJX_HD float d_exp_core(float x) { return m_exp_core(x); }
#endif

GJX_HD float m_erfinv(float x) {
  float w = -m_log_normal((1.0f - x) * (1.0f + x));
  float p;
  if (w < 5.0f) {
    w = w - 2.5f;
    p = 2.81022636e-08f;
    p = __builtin_fmaf(p, w, 3.43273939e-07f);
    p = __builtin_fmaf(p, w, -3.5233877e-06f);
    p = __builtin_fmaf(p, w, -4.39150654e-06f);
    p = __builtin_fmaf(p, w, 0.00021858087f);
    p = __builtin_fmaf(p, w, -0.00125372503f);
    p = __builtin_fmaf(p, w, -0.00417768164f);
    p = __builtin_fmaf(p, w, 0.246640727f);
    p = __builtin_fmaf(p, w, 1.50140941f);
  } else {
    w = __builtin_sqrtf(w) - 3.0f;
    p = -0.000200214257f;
    p = __builtin_fmaf(p, w, 0.000100950558f);
    p = __builtin_fmaf(p, w, 0.00134934322f);
    p = __builtin_fmaf(p, w, -0.00367342844f);
    p = __builtin_fmaf(p, w, 0.00573950773f);
    p = __builtin_fmaf(p, w, -0.0076224613f);
    p = __builtin_fmaf(p, w, 0.00943887047f);
    p = __builtin_fmaf(p, w, 1.00167406f);
    p = __builtin_fmaf(p, w, 2.83297682f);
  }
  return p * x;
}

GJX_HD float m_lgamma(float x) {
  float p = 1.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (x < 8.0f) {
      p = p * x;
      x = x + 1.0f;
    }
  }
  const float xi = 1.0f / x;
  const float xi2 = xi * xi;
  float s = __builtin_fmaf(xi2, 7.9365079365e-4f, -2.7777777778e-3f);
  s = __builtin_fmaf(s, xi2, 8.3333333333e-2f);
  s = s * xi;
  float r = (x - 0.5f) * m_log(x);
  r = r - x;
  r = r + 0.91893853320467f;
  r = r + s;
  r = r - m_log(p);
  return r;
}

GJX_HD float uniform01(uint32_t bits) { return u2f((bits >> 9) | 0x3F800000u) - 1.0f; }

GJX_HD float std_normal(uint32_t bits) {
  const float lo = -0.99999994f;
  float u = uniform01(bits) * 2.0f + lo;
  u = u > lo ? u : lo;
  return 1.41421356237309505f * m_erfinv(u);
}

// --- PHILOX normal sites: Box-Muller over PAIRS OF PARTICLES.  Particles j0 = j & ~1 and j1 = j0 + 1 of a
// key batch (key lanes j0+1, j0+2) share one transform at every Normal site: the radius comes from j0's
// draw word, the angle from j1's, j0 takes the cosine and j1 the sine — two independent standard normals
// for ONE log, one sqrt and one sine/cosine pair, against a log + erfinv polynomial per normal.  Each particle
// still consumes exactly one word per site, so keys, blocks and draw indices are unchanged.
// The transform is TABLE-DRIVEN (tools/gen_bm_tables.py writes the constants; the oracle holds the same bits):
//   log u:  Cephes' mantissa reduction to [sqrt(1/2), sqrt(2)), then the 64-entry table over the offset mantissa:
//           log m = log c_k + log1p(q), q = fma(m, 1/c_k, -1) (|q| <= 2^-6), log1p by its series to q^4; the entry
//           holds fl(1/c_k) and -log of THAT float, and the interval around 1 has c = 1, so u -> 1 keeps full
//           relative accuracy;
//   angle 2 pi a / 2^24:  the top 8 bits pick (cos A_k, sin A_k), A_k = 2 pi (k + 1/2) / 256; the low 16 bits are
//           d in [-pi/256, pi/256): sin d = d - d^3/6, cos d = 1 - d^2/2 + d^4/24 (truncation < 3e-12), and the
//           angle-sum formulas give the pair.
// 40 vector instructions per transform against 69 for the polynomial form (Cephes log to degree 9, octant-reduced
// sinf / cosf kernels) it replaces; two 8-byte table reads, from LDS in the kernels that stage the tables
// (GJX_BM_LDS: bm_stage() at kernel entry — the compiled importance and scan kernels), from the constant arrays
// elsewhere.  Every operation is an IEEE fma / mul / add on f32, so HIP == oracle bit for bit; against float64 the
// normals are within 2.5e-7 of the radius (tests/test_oracle_pinning.py: test_box_muller_against_float64).  THREEFRY keeps jax's erfinv form.
struct BmEnt {
  uint32_t a, b;
};
// BEGIN BM TABLES (generated: tools/gen_bm_tables.py)
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
// END BM TABLES
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ const BmEnt g_bm_lg[64] = GJX_BM_LG_INIT;
static __device__ const BmEnt g_bm_cs[256] = GJX_BM_CS_INIT;
#else
static const BmEnt g_bm_lg[64] = GJX_BM_LG_INIT;
static const BmEnt g_bm_cs[256] = GJX_BM_CS_INIT;
#endif
#if defined(GJX_BM_LDS) && defined(__HIP_DEVICE_COMPILE__)
__shared__ BmEnt s_bm_lg[64];
__shared__ BmEnt s_bm_cs[256];
GJX_DEV void bm_stage() {  // every thread of the workgroup, before its first Normal site
  // (r04: every load of the thread issued before the first is waited for — as two copy loops a 64-thread workgroup went
  // through FIVE dependent memory trips at kernel entry, load / wait / write each; workgroups have 64, 128 or 256 threads)
  const uint32_t t = threadIdx.x, bd = blockDim.x;
  BmEnt c[4], l{0u, 0u};
#pragma unroll
  for (int k = 0; k < 4; ++k) c[k] = t + (uint32_t)k * bd < 256u ? g_bm_cs[t + (uint32_t)k * bd] : BmEnt{0u, 0u};
  if (t < 64u) l = g_bm_lg[t];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (t + (uint32_t)k * bd < 256u) s_bm_cs[t + (uint32_t)k * bd] = c[k];
  if (t < 64u) s_bm_lg[t] = l;
  for (uint32_t i = t + 4u * bd; i < 256u; i += bd) s_bm_cs[i] = g_bm_cs[i];  // (workgroups below 64 threads: not generated)
  __syncthreads();
}
#define GJX_BM_LG s_bm_lg
#define GJX_BM_CS s_bm_cs
#else
#define GJX_BM_LG g_bm_lg
#define GJX_BM_CS g_bm_cs
#endif
// Correctly rounded sqrtf for arguments that are zero or not tiny (here: -2 log u in {0} U [1e-7, 45]).  Same
// result as __builtin_sqrtf bit for bit; on the device it is the hardware estimate (<= 1 ulp) stepped to the
// correctly rounded neighbour by two exact residuals, without the scaling and class handling the general
// lowering adds for tiny and special arguments (16 -> 9 instructions).
GJX_HD float sqrt_pos(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GJX_GENERAL_SQRT)
  const float s = __builtin_amdgcn_sqrtf(x);
  const float dn = u2f(f2u(s) - 1u), up = u2f(f2u(s) + 1u);
  const float e_dn = __builtin_fmaf(-dn, s, x), e_up = __builtin_fmaf(-up, s, x);
  float r = e_dn <= 0.0f ? dn : s;
  r = e_up > 0.0f ? up : r;
  return r;
#else
  return __builtin_sqrtf(x);
#endif
}
GJX_HD void bm_pair(uint32_t w_radius, uint32_t w_angle, float& z_cos, float& z_sin) {
  // radius: u in (0, 1] with full float resolution near 0 (tails to 6.6 sigma)
  const float u = ((float)w_radius + 1.0f) * 2.3283064365386963e-10f;
#ifdef GJX_FAST_MATH_DEVICE
  {  // the same map from the same two words through the hardware functions (v_sin / v_cos take revolutions)
    const float rf = __builtin_amdgcn_sqrtf(-1.38629436111989062f * __builtin_amdgcn_logf(u));
    const float tf = (float)(w_angle >> 8) * 5.9604644775390625e-08f;
    z_cos = rf * __builtin_amdgcn_cosf(tf);
    z_sin = rf * __builtin_amdgcn_sinf(tf);
    return;
  }
#endif
  const uint32_t t = f2u(u) - 0x3f3504f3u;
  const BmEnt lg = GJX_BM_LG[(t >> 17) & 63u];
  const float m = u2f((t & 0x007fffffu) + 0x3f3504f3u);
  const float fe = (float)((int32_t)t >> 23);
  const float q = __builtin_fmaf(m, u2f(lg.a), -1.0f);
  float p = __builtin_fmaf(q, -0.25f, 0.333333343f);
  p = __builtin_fmaf(p, q, -0.5f);
  const float l1 = __builtin_fmaf(p, q * q, q);
  const float lu = __builtin_fmaf(fe, 0.693147182f, u2f(lg.b)) + l1;
  const float r = sqrt_pos(-2.0f * lu);
  const BmEnt cs = GJX_BM_CS[w_angle >> 24];
  const float d = __builtin_fmaf((float)((w_angle >> 8) & 0xffffu), 3.74507035e-07f, -0.0122718466f);
  const float d2 = d * d;
  const float sd = __builtin_fmaf(d2 * d, -0.166666672f, d);
  const float cd = __builtin_fmaf(d2, __builtin_fmaf(d2, 0.0416666679f, -0.5f), 1.0f);
  const float ca = u2f(cs.a), sa = u2f(cs.b);
  z_cos = r * __builtin_fmaf(ca, cd, -(sa * sd));
  z_sin = r * __builtin_fmaf(sa, cd, ca * sd);
}
constexpr uint32_t kTagTwin = 0x54u;  // 'T': the angle word of a lane-0 key (it has no partner particle)
// The standard normal of a Normal SITE (generic, one particle: both words of its pair are derived here;
// kernels that own the whole pair use bm_pair directly).
template <int IMPL>
GJX_HD float site_normal(const Stream<IMPL>& st) {
  if (IMPL == 0 || !st.hf) return std_normal(st.bits32(0));
  float zc, zs;
  uint32_t o[4];
  if ((st.k.l0 | st.k.l1) == 0u) {  // lane-0 key: radius word from its draw block, angle word from the twin block
    philox4x32(st.k.k0, st.k.k1, 0u, 0u, st.f >> 2, kTagTwin, o[0], o[1], o[2], o[3]);
    const uint32_t sel = st.f & 3u;
    bm_pair(philox_single_draw(st.k, st.f), sel == 0 ? o[0] : (sel == 1 ? o[1] : (sel == 2 ? o[2] : o[3])), zc, zs);
    return zc;
  }
  philox_pair_block(st.k, st.f >> 1, o);  // both words of the pair are in its block
  if (st.f & 1u) bm_pair(o[2], o[3], zc, zs);
  else bm_pair(o[0], o[1], zc, zs);
  return ((st.k.l0 - 1u) & 1u) ? zs : zc;
}

// Standard normals of SMC slots 4g .. 4g+3 (the LGSSM filter).  THREEFRY: erfinv of each slot's draw.  PHILOX: two
// Box-Muller transforms over the quad's words, (w0, w1) -> slots 4g, 4g+1 and (w2, w3) -> slots 4g+2, 4g+3.
template <int IMPL>
GJX_HD void smc_quad_normals(Key step_key, uint64_t g, float (&z)[4]) {
  uint32_t w[4];
  smc_quad_bits<IMPL>(step_key, g, w);
  if (IMPL == 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) z[u] = std_normal(w[u]);
  } else {
    bm_pair(w[0], w[1], z[0], z[1]);
    bm_pair(w[2], w[3], z[2], z[3]);
  }
}
template <int IMPL>
GJX_HD float smc_slot_normal(Key step_key, uint64_t j) {  // one slot on its own
  if (IMPL == 0) return std_normal(smc_slot_bits<IMPL>(step_key, j));
  float z[4];
  smc_quad_normals<IMPL>(step_key, j >> 2, z);
  const uint32_t sel = (uint32_t)j & 3u;
  return sel == 0 ? z[0] : (sel == 1 ? z[1] : (sel == 2 ? z[2] : z[3]));
}

// --- log-densities (TFP formulas).  The *_pre forms take the per-site constants a plan hoists.
GJX_HD float normal_rs(float scale) { return 1.0f / scale; }
GJX_HD float normal_lognorm(float scale) { return 0.91893853320467f + d_log(scale); }
GJX_HD float logpdf_normal_pre(float x, float loc, float rs, float lognorm) {
  const float d = x * rs - loc * rs;
  return (-0.5f * d) * d - lognorm;
}
GJX_HD float logpdf_normal(float x, float loc, float scale) {
  return logpdf_normal_pre(x, loc, normal_rs(scale), normal_lognorm(scale));
}
GJX_HD float xlogy(float a, float y) { return a == 0.0f ? 0.0f : a * d_log(y); }
GJX_HD float gamma_lognorm(float conc, float rate) { return m_lgamma(conc) - conc * m_log(rate); }
GJX_HD float logpdf_gamma_pre(float x, float conc, float rate, float lognorm) {
  return (xlogy(conc - 1.0f, x) - rate * x) - lognorm;
}
GJX_HD float logpdf_gamma(float x, float conc, float rate) {
  return logpdf_gamma_pre(x, conc, rate, gamma_lognorm(conc, rate));
}
GJX_HD float beta_lbeta(float a, float b) { return (m_lgamma(a) + m_lgamma(b)) - m_lgamma(a + b); }
GJX_HD float logpdf_beta_pre(float x, float a, float b, float lbeta) {
  return (xlogy(a - 1.0f, x) + xlogy(b - 1.0f, 1.0f - x)) - lbeta;
}
GJX_HD float logpdf_beta(float x, float a, float b) {
  return logpdf_beta_pre(x, a, b, beta_lbeta(a, b));
}
GJX_HD float logpdf_bernoulli(bool e, float p) { return e ? d_log(p) : d_log(1.0f - p); }

// --- Marsaglia-Tsang Gamma(conc, 1).  Attempt a of gamma `which` uses sub-stream 1 + 2a + which
// (w0 -> normal, w1 -> uniform); the conc<1 boost uniform is word `which` of sub-stream 0.
template <int IMPL>
GJX_DEV float std_gamma(const Stream<IMPL>& st, int which, float conc) {
  const bool boost = conc < 1.0f;
  const float a = boost ? conc + 1.0f : conc;
  const float d = a - 0.33333334f;
  const float c = 1.0f / __builtin_sqrtf(9.0f * d);
  float v = 1.0f;
  for (int att = 0; att < 64; ++att) {
    uint32_t w0, w1;
    st.words((uint32_t)(1 + 2 * att + which), w0, w1);
    const float x = std_normal(w0);
    const float t = 1.0f + c * x;
    if (t <= 0.0f) continue;
    v = (t * t) * t;
    const float u = uniform01(w1);
    float rhs = (0.5f * x) * x + d;
    rhs = rhs - d * v;
    rhs = rhs + d * m_log(v);
    if (m_log(u) < rhs) break;
  }
  float g = d * v;
  if (boost) {
    uint32_t w0, w1;
    st.words(0u, w0, w1);
    const float ub = uniform01(which ? w1 : w0);
    g = g * m_exp(m_log(ub) / conc);
  }
  return g;
}

// r04: the same draws for the P particles of a lane (the pair / quad forms of the generated kernels), scheduled for the
// wavefront.  std_gamma's loop runs as long as ANY lane of the wave still rejects: with ~96 % acceptance nearly every wave
// runs a second trip for ~3 lanes, for every one of a lane's draws in turn (measured on the README's beta-bernoulli model, 8
// gammas per lane: 0.68 of the VALU lane-cycles active).  Here the first attempt of all P particles is straight-line code —
// independent chains the compiler interleaves — and the retries share ONE loop: each trip a lane advances its first
// unfinished particle by one attempt.  Attempt a of particle u draws from the same sub-stream 1 + 2a + which of u's own
// stream and the same comparisons decide: the same values as P calls of std_gamma, bit for bit, in fewer wave-trips.
template <int IMPL>
GJX_DEV bool gamma_attempt(const Stream<IMPL>& st, int which, int att, float d, float c, float& v) {
  uint32_t w0, w1;
  st.words((uint32_t)(1 + 2 * att + which), w0, w1);
  const float x = std_normal(w0);
  const float t = 1.0f + c * x;
  if (t <= 0.0f) return false;
  v = (t * t) * t;
  const float u = uniform01(w1);
  float rhs = (0.5f * x) * x + d;
  rhs = rhs - d * v;
  rhs = rhs + d * m_log(v);
  return m_log(u) < rhs;
}
template <int IMPL, int P>
GJX_DEV void std_gamma_multi(const Stream<IMPL> (&st)[P], int which, const float (&conc)[P], float (&out)[P]) {
  float d[P], c[P], v[P];
  int att[P];
  bool done[P];
#pragma unroll
  for (int u = 0; u < P; ++u) {
    const float a = conc[u] < 1.0f ? conc[u] + 1.0f : conc[u];
    d[u] = a - 0.33333334f;
    c[u] = 1.0f / __builtin_sqrtf(9.0f * d[u]);
    v[u] = 1.0f;
  }
#pragma unroll
  for (int u = 0; u < P; ++u) {
    done[u] = gamma_attempt<IMPL>(st[u], which, 0, d[u], c[u], v[u]);
    att[u] = 1;
  }
  for (;;) {
    int pick = -1;
#pragma unroll
    for (int u = P - 1; u >= 0; --u) pick = (!done[u] && att[u] < 64) ? u : pick;
#if defined(__HIP_DEVICE_COMPILE__)
    if (__ballot(pick >= 0) == 0) break;  // (wave-uniform: no lane has an unfinished particle)
#else
    if (pick < 0) break;
#endif
    if (pick >= 0) {
      Stream<IMPL> s = st[0];
      float dd = d[0], cc = c[0], vv = v[0];
      int aa = att[0];
#pragma unroll
      for (int u = 1; u < P; ++u)
        if (pick == u) { s = st[u]; dd = d[u]; cc = c[u]; vv = v[u]; aa = att[u]; }
      const bool ok = gamma_attempt<IMPL>(s, which, aa, dd, cc, vv);
#pragma unroll
      for (int u = 0; u < P; ++u)
        if (pick == u) { v[u] = vv; att[u] = aa + 1; done[u] = ok; }
    }
  }
#pragma unroll
  for (int u = 0; u < P; ++u) {
    float g = d[u] * v[u];
    if (conc[u] < 1.0f) {
      uint32_t w0, w1;
      st[u].words(0u, w0, w1);
      const float ub = uniform01(which ? w1 : w0);
      g = g * m_exp(m_log(ub) / conc[u]);
    }
    out[u] = g;
  }
}

// --- fixed-point weights: q = rint(exp(lw - m) * 2^frac) as u64 (exact, order-independent sums).
GJX_HD uint64_t fixw(float lw, float m, int frac) {
  if (lw == m) return (uint64_t)1 << frac;
  const float d = lw - m;
  if (!(d >= -80.0f)) return 0;
  const float s = m_exp(d);
  const float t = s * u2f((uint32_t)(127 + frac) << 23);
  return (uint64_t)__builtin_rintf(t);
}
// --- row-anchored fixed point: a 256-particle row is anchored at the power of two
// 2^e just above its own maximum, so the producing kernel can emit (e, sum) without knowing the
// global maximum; rows combine by exact right shifts.
constexpr int kRowFrac = 30;
constexpr int32_t kRowEmpty = -(1 << 30);
constexpr int kLseBuckets = 64;                     // shifts 0..63 relative to the anchor
constexpr int kLseRecordWords = 1 + kLseBuckets;    // [0] = anchor e (sign-extended), [1+d] = bucket d
GJX_HD int32_t row_anchor(float m) {
  if (!(m > -__builtin_inff())) return kRowEmpty;  // -inf or NaN: the row carries no mass
  float t = m * 1.44269504088896341f;
  t = t > 16777216.0f ? 16777216.0f : (t < -16777216.0f ? -16777216.0f : t);
  return (int32_t)__builtin_ceilf(t);
}
GJX_HD uint64_t rowfix(float lw, int32_t e) {
#if defined(__HIP_DEVICE_COMPILE__)
  {
    // the same function as selects (r04: four calls per lane in the step's emission, each with two divergent branches): a
    // dead weight (empty row, -inf / NaN, below m_exp's flush at -86) runs the polynomial on 0 and is zeroed at the end
    const float fe = (float)e;
    float d = __builtin_fmaf(-fe, 0.693359375f, lw);
    d = __builtin_fmaf(-fe, -2.12194440e-4f, d);
    d = d > 1.0f ? 1.0f : d;
    const bool live = e != kRowEmpty && lw > -__builtin_inff() && d >= -86.0f;
    const uint32_t q = (uint32_t)__builtin_rintf(d_exp_core(live ? d : 0.0f) * 1073741824.0f);
    return live ? q : 0u;
  }
#endif
  if (e == kRowEmpty || !(lw > -__builtin_inff())) return 0;
  const float fe = (float)e;
  float d = __builtin_fmaf(-fe, 0.693359375f, lw);
  d = __builtin_fmaf(-fe, -2.12194440e-4f, d);
  d = d > 1.0f ? 1.0f : d;  // only a weight beyond the clamped anchor (+inf, > 1.1e7): keeps the conversion defined
  // (d <= 1: the product is below e 2^30 < 2^32 — one 32-bit conversion instead of the 64-bit sequence)
  return (uint64_t)(uint32_t)__builtin_rintf(d_exp(d) * 1073741824.0f);
}

GJX_HD uint32_t cat_fix(float l, float m) {
  if (l == m) return 1u << kCatFrac;
  const float d = l - m;
  if (!(d >= -80.0f)) return 0u;
  return (uint32_t)__builtin_rintf(m_exp(d) * 8388608.0f);
}
GJX_HD int frac_bits(uint64_t n_total) {
  int lg = 0;
  while (lg < 63 && ((uint64_t)1 << lg) < n_total) ++lg;
  const int f = 62 - lg;
  return f > 40 ? 40 : (f < 8 ? 8 : f);
}
GJX_HD float gumbel_from_bits(uint32_t bits) {
  const float tiny = 1.17549435e-38f;
  float u = uniform01(bits) + tiny;
  u = u > tiny ? u : tiny;
  return -m_log(-m_log(u));
}

// Systematic comb: number of teeth (j + u0), j in [0, n_out), strictly below mass C * scale.
GJX_HD int64_t teeth_below(uint64_t C, double scale, double u0, int64_t n_out) {
  const double P = (double)C * scale;
  const double c = __builtin_ceil(P - u0);
  if (!(c > 0.0)) return 0;
  if (c >= (double)n_out) return n_out;
  return (int64_t)(int32_t)c;  // (n_out < 2^31: one v_cvt_i32_f64 instead of the 64-bit conversion sequence)
}
GJX_HD double u0_from_bits(uint64_t U) { return (double)(U >> 11) * 0x1.0p-53; }

// ------------------------------------------------------------------------------------------------
// 64-wide wavefront / 256-thread workgroup reductions and scans (LDS-staged across the 4 waves).
// ------------------------------------------------------------------------------------------------
// Cross-lane steps are DPP modifiers (data-parallel primitives: row_shr / row_bcast), i.e. plain VALU
// instructions: a 64-lane inclusive scan is 6 combine steps with no LDS round trip and no s_waitcnt (the
// ds_bpermute form of __shfl_* costs an LDS access and a wait per step; the importance kernel's row statistics
// were 18 such dependent round trips per row).  Pattern (GCN3+ cross-lane reference): shifts by 1, 2, 3 of the
// input inside each row of 16, then row_shr:4 / row_shr:8 on the upper banks, then row_bcast:15 / row_bcast:31
// to carry row totals; lanes without a source keep the identity.
template <int CTRL, int ROW_MASK, int BANK_MASK>
GJX_DEV uint32_t dpp_u32(uint32_t identity, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}
template <int CTRL, int ROW_MASK, int BANK_MASK>
GJX_DEV uint64_t dpp_u64(uint64_t identity, uint64_t v) {
  const uint32_t lo = dpp_u32<CTRL, ROW_MASK, BANK_MASK>((uint32_t)identity, (uint32_t)v);
  const uint32_t hi = dpp_u32<CTRL, ROW_MASK, BANK_MASK>((uint32_t)(identity >> 32), (uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr3 = 0x113, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118,
              kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143, kDppWaveShr1 = 0x138;
// Inclusive scan over the wave; `op` associative and commutative, `id` its identity.  Mov is the per-type DPP move.
#define GJX_WAVE_SCAN_BODY(T, MOV)                                               \
  T r = op(v, MOV<kDppRowShr1, 0xf, 0xf>(id, v));                                \
  r = op(r, MOV<kDppRowShr2, 0xf, 0xf>(id, v));                                  \
  r = op(r, MOV<kDppRowShr3, 0xf, 0xf>(id, v));                                  \
  r = op(r, MOV<kDppRowShr4, 0xf, 0xe>(id, r));                                  \
  r = op(r, MOV<kDppRowShr8, 0xf, 0xc>(id, r));                                  \
  r = op(r, MOV<kDppRowBcast15, 0xa, 0xf>(id, r));                               \
  r = op(r, MOV<kDppRowBcast31, 0xc, 0xf>(id, r));                               \
  return r;
template <class Op>
GJX_DEV uint32_t wave_scan_u32(uint32_t v, uint32_t id, Op op) { GJX_WAVE_SCAN_BODY(uint32_t, dpp_u32) }
template <class Op>
GJX_DEV uint64_t wave_scan_u64(uint64_t v, uint64_t id, Op op) { GJX_WAVE_SCAN_BODY(uint64_t, dpp_u64) }
#undef GJX_WAVE_SCAN_BODY
GJX_DEV uint32_t wave_last_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
GJX_DEV uint64_t wave_last_u64(uint64_t v) {
  return ((uint64_t)wave_last_u32((uint32_t)(v >> 32)) << 32) | wave_last_u32((uint32_t)v);
}
// Wave-wide reductions; result valid in every lane.
GJX_DEV float wave_max(float v) {
  // float max through the u32 scan: bit patterns are combined with a float compare (exact; order-free once a NaN — which
  // every sequential `x > m ? x : m` maximum of the specification skips — has been replaced by the identity)
  v = v == v ? v : -__builtin_inff();
  const uint32_t r = wave_scan_u32(f2u(v), f2u(-__builtin_inff()),
                                   [](uint32_t a, uint32_t b) { return u2f(b) > u2f(a) ? b : a; });
  return u2f(wave_last_u32(r));
}
GJX_DEV uint64_t wave_sum(uint64_t v) {
  return wave_last_u64(wave_scan_u64(v, 0, [](uint64_t a, uint64_t b) { return a + b; }));
}
GJX_DEV int wave_sum_int(int v) {
  return (int)wave_last_u32(wave_scan_u32((uint32_t)v, 0u, [](uint32_t a, uint32_t b) { return a + b; }));
}
// Block-wide max; result valid in every thread.  `sh` needs 4 floats.
GJX_DEV float block_max(float v, float* sh) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
#pragma unroll
  for (int i = 1; i < kBlock / kWave; ++i) r = sh[i] > r ? sh[i] : r;
  return r;
}
GJX_DEV uint64_t block_sum(uint64_t v, uint64_t* sh) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  uint64_t r = 0;
#pragma unroll
  for (int i = 0; i < kBlock / kWave; ++i) r += sh[i];
  return r;
}
// Inclusive wave scan of u64.
GJX_DEV uint64_t wave_scan_incl(uint64_t v) {
  return wave_scan_u64(v, 0, [](uint64_t a, uint64_t b) { return a + b; });
}
// ... of values below 2^52 (tile masses: a tile sums 1024 weights below 2^31.5, a thread at most 16 tiles): two 26-bit halves
// scanned as u32 — a u32 add takes its DPP operand directly (8 instructions per scan), the 64-bit add does not (45) — and
// the halves' lane sums stay below 2^32
GJX_DEV uint64_t wave_scan_incl_52(uint64_t v) {
  const uint32_t lo = wave_scan_u32((uint32_t)v & 0x3ffffffu, 0u, [](uint32_t a, uint32_t b) { return a + b; });
  const uint32_t hi = wave_scan_u32((uint32_t)(v >> 26), 0u, [](uint32_t a, uint32_t b) { return a + b; });
  return ((uint64_t)hi << 26) + lo;
}
GJX_DEV uint64_t wave_sum_52(uint64_t v) { return wave_last_u64(wave_scan_incl_52(v)); }
// Exclusive block scan of one u64 per thread; returns the exclusive prefix, total in `total`.
GJX_DEV uint64_t block_scan_excl(uint64_t v, uint64_t* sh, uint64_t& total) {
  const uint64_t incl = wave_scan_incl(v);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 63) sh[w] = incl;
  __syncthreads();
  uint64_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kBlock / kWave; ++i) {
    if (i < w) base += sh[i];
    tot += sh[i];
  }
  total = tot;
  return base + incl - v;
}

// Exclusive block max-scan of one int per thread (identity 0) and a block-wide int sum.  `sh` needs 4 ints.
GJX_DEV int block_scan_max_excl(int v, int* sh) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // values are >= 0 (identity 0): the unsigned scan orders them like the signed one
  const int incl = (int)wave_scan_u32((uint32_t)v, 0u, [](uint32_t a, uint32_t b) { return b > a ? b : a; });
  int excl = (int)dpp_u32<kDppWaveShr1, 0xf, 0xf>(0u, (uint32_t)incl);  // lane i <- lane i-1, lane 0 keeps 0
  __syncthreads();
  if (lane == 63) sh[w] = incl;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kBlock / kWave; ++i)
    if (i < w) excl = sh[i] > excl ? sh[i] : excl;
  return excl;
}
GJX_DEV int block_sum_int(int v, int* sh) {
  v = wave_sum_int(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  int r = 0;
#pragma unroll
  for (int i = 0; i < kBlock / kWave; ++i) r += sh[i];
  return r;
}

// ------------------------------------------------------------------------------------------------
// Systematic resampling, OUTPUT-tile-centric, one launch per SMC step.
//
// Weights are TILE-ANCHORED fixed point: tile t (1024 consecutive particles) is anchored at the power of two just
// above its own maximum, e_t = row_anchor(max_t lw); q_i = rowfix(lw_i, e_t) (30 fractional bits, one u32 per particle:
// what a step stores instead of the log-weight); the tile's record holds S_t = sum q, e_t, the running sum of q at
// every 64th particle (16 sub-prefixes) and the ESS sums.  All of that is known to the workgroup that PRODUCES the
// tile's log-weights — no grid-wide maximum is needed, so the kernel that propagates a population also emits what the
// next resampling reads, and a bootstrap step is ONE launch.  The consumer merges the records: e = max e_t, d_t = e - e_t,
// M_t = S_t >> d_t, P_t = sum_{t' < t} M_t', total Q = P_ntiles: exact integers, the same bits for every tiling of the
// work, every number of ranks and the oracle.  Teeth of the comb below particle i of tile t (c_i = the running sum of q
// inside the tile, exact in float64): n_i = min(clamp(ceil(fma(c_i, scale 2^-d_t, P_t scale - u0))), nhi_t), the tile
// itself ending at nhi_t = clamp(ceil(P_{t+1} scale - u0)) (comb_* below): monotone, and consistent across tiles.
//
// Workgroup b owns OUTPUT slots [b*1024, (b+1)*1024): from the merged records it finds the source tiles whose teeth
// fall into its slots (ancestors are monotone: a contiguous range, typically 2-3 tiles), re-scans their stored weights
// (integer adds: no exponential), turns the running sums into teeth counts, marks where every source's run of slots
// starts, spreads the marks with a max-scan, gathers the ancestors' state, propagates, weights, and emits its own tile's
// weights and record.  Collapse-proof by construction: under weight collapse every output tile reads the same heavy
// source tile; when an output tile has MANY light sources (more than kScanMax tiles with a tooth in it) every slot finds
// its ancestor by binary search of the tile prefix, then of the tile's 16 sub-prefixes, then a walk over at most 64
// stored weights — the cost of a step is bounded whatever the weights are.
// ------------------------------------------------------------------------------------------------
// Several independent filters stepping in ONE launch (the bootstrap filter vmapped over keys): workgroup
// f * tiles + b serves output tile b of filter f.  Filter f's particles, records, (e, q) results and keys lie
// f * stride / f * tiles / f * mq_stride further.
constexpr int kMaxFilters = 16;
struct FilterBatch {
  uint32_t n_filters = 0;  // <= 1: a single filter (nothing below is read)
  uint32_t tiles = 0;      // tiles per filter
  uint64_t stride = 0;     // particles between consecutive filters in every per-particle array (tiles * 1024)
  uint64_t mq_stride = 0;  // entries between the filters' per-step (e, q) results
  Key step_key[kMaxFilters];
  double u0[kMaxFilters];  // comb offsets (from the filters' resampling keys)
};

constexpr int kTileFrac = kRowFrac;   // fractional bits of the tile-anchored weights (rowfix)
constexpr int kEssShift = kTileFrac - 16;
constexpr int kMaxLdsTiles = 1024;    // populations up to 2^20 particles keep the merged tile prefix in LDS
#ifndef GJX_SCAN_LANES
#define GJX_SCAN_LANES 192
#endif
constexpr int kScanLanes = GJX_SCAN_LANES;  // lanes of the workgroup that scan sources (8 consecutive sources each)
constexpr int kWindow = 8 * kScanLanes;      // sources one round of the window scan covers: 1.5 tiles (the slots of an
                                             // output tile draw from ~1024 + 64 sources; a wider window only reads more)
constexpr int kScanMax = 4;           // rounds of the window scan before an output tile searches per slot
constexpr int kGroupTiles = 256;      // r04, populations beyond kMaxLdsTiles: tiles per GROUP record (k_group_records)
constexpr int kMaxGroups = 256;       // groups a workgroup merges (one per thread): up to 65 536 tiles (67M particles)
constexpr int kSubs = 16;             // sub-prefixes per tile: the running sum of q after every 64th particle
constexpr int kSubLen = kTile / kSubs;
// A tile's record is kept as three DENSE arrays indexed by tile — the 16-byte header every workgroup merges (a strided
// header inside a 160-byte record cost every workgroup of a 1e6-particle step 156 KB of cache lines for 15 KB of
// payload), the 16 sub-prefixes only the window scan's few lanes read, the ESS sums only adaptive filters read.
struct alignas(16) TileRec {
  uint64_t s;   // S_t: the tile's mass relative to its own anchor
  int32_t e;    // e_t (kRowEmpty: no mass)
  int32_t pad;
};
struct alignas(16) TileSub {
  uint64_t sub[kSubs];    // sub[b] = sum of q over the tile's particles [0, 64 (b + 1))  (sub[15] = S_t)
};
struct alignas(16) TileEss {
  uint64_t r1, r2;        // ESS sums of the tile (tile-anchored): sum r_i, sum r_i^2, r_i = q_i >> 14
};
// r04: the record of a GROUP of kGroupTiles consecutive tiles, anchored at the group's own maximum e: for EVERY shift D of
// that anchor (the merged anchor of a population is e + D for some D >= 0) the group's merged mass, sum over its tiles of
// S_t >> (e - e_t + D) — the per-tile floors summed, so a consumer that picks entry D gets exactly the sum of the masses it
// would have computed tile by tile — and likewise the ESS sums (r1: shift by d, r2: by 2 d).  D >= 64: nothing is left.
struct alignas(16) GroupRec {
  uint64_t mass[64];
  uint64_t r1[64];
  uint64_t r2[64];
  int32_t e;              // kRowEmpty: no mass in the group
  int32_t pad[3];
};
static_assert(sizeof(TileRec) == 16 && sizeof(TileSub) == 128 && sizeof(TileEss) == 16, "gjx.h gjx_tile_rec / gjx_tile_sub / gjx_tile_ess");
// shift of a tile's fixed point relative to the merged anchor e (>= every e_t): 64 = the tile carries no mass
GJX_HD int tile_shift(int32_t e, int32_t et) {
  if (et == kRowEmpty) return 64;
  const int64_t d = (int64_t)e - (int64_t)et;
  return d > 63 ? 64 : (int)d;
}
GJX_HD uint64_t shr64(uint64_t v, int d) { return d >= 64 ? 0 : v >> d; }
// layout of the precomputed prefix of large populations (k_scan_records): [0 .. ntiles] exclusive prefix (entry
// ntiles = total), then e (sign-extended), R1, R2
GJX_HD uint64_t prefix_words(uint64_t ntiles) { return ntiles + 4; }

// ---- r04: the peer transport (gjx.h: gjx_smc_peers) ----------------------------------------------------------------
// The SOURCE population of a step is distributed over `world` ranks: tile k (and its particles) lives in the arena of rank
// k / tiles_per_rank, at the address it has in this rank's arena plus delta[owner] bytes.  The tile RECORDS (and ESS sums)
// are read from this rank's own arena, where every rank deposited them (k_peer_signal); sub-prefixes, weights and state
// columns are read where they live.  flags / wait_value: the step reads nothing of the source before every peer has
// arrived (peer_wait_wave).
constexpr int kMaxPeers = 8;
struct PeerMap {
  int32_t world = 0;             // 0: the source population is local (nothing below is read)
  uint32_t tiles_per_rank = 0;
  int64_t delta[kMaxPeers] = {0, 0, 0, 0, 0, 0, 0, 0};
  const uint64_t* flags = nullptr;  // this rank's arrival words [world]
  uint32_t* error = nullptr;        // set to 1 by a wait that timed out
  uint64_t wait_value = 0;
  uint64_t timeout_ticks = 0;       // of s_memrealtime (100 MHz)
  // a deferred signal (gjx.h gjx_smc_peers.signal_*): what the group-record launch deposits and raises before it waits
  int32_t rank = 0;
  const TileRec* sig_recs = nullptr;
  const TileEss* sig_ess = nullptr;
  uint64_t sig_first = 0, sig_n = 0, sig_value = 0;
};
template <class T>
GJX_DEV const T* peer_ptr(const T* p, int64_t delta) {
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(p) + delta);
}
// Called by ONE whole wave: lanes q < world poll flags[q] (system-scope relaxed loads, a sleep between polls) until every
// one is >= wait_value; then ONE system-scope acquire, so that whatever the peers published before raising their words is
// what this CU loads from now on (the caller's workgroup barrier hands that to the other waves).  BOUNDED: after
// timeout_ticks the wave gives up, sets *error and returns false — every wave of every workgroup reaches an exit.
GJX_DEV bool peer_wait_wave(const PeerMap& pm) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int lane = threadIdx.x & 63;
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
  bool ready = false;
  // (a wait that has already timed out on this rank is not waited for again: every later wait of the run fails at once, so a
  // peer that never arrives costs ONE timeout, not one per launch)
  if (__hip_atomic_load(pm.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return false;
  for (;;) {
    const uint64_t v = lane < pm.world ? __hip_atomic_load(pm.flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : ~0ull;
    ready = __ballot(v < pm.wait_value) == 0;
    if (ready) break;
    if (__builtin_amdgcn_s_memrealtime() - t0 > pm.timeout_ticks) break;
    __builtin_amdgcn_s_sleep(16);
    // a peer's store reaches this device's memory without passing through its caches: the next poll must not be served by a
    // line cached before it (whatever the memory type of the arena)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  }
  if (ready) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  } else if (lane == 0) {
    __hip_atomic_store(pm.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  return ready;
#else
  (void)pm;
  return false;
#endif
}

// Profiling hooks of the SMC step (early exits by phase, tools/phases_smc.sh): compiled in only with -DGJX_PROFILE_HOOKS — a
// separate build of the library; the shipped kernels carry none of them, and GJX_SMC_DEBUG_STOP / GJX_SMC_DEBUG_FIXED do
// nothing there (a run that stops early returns GJX_OK with truncated state: not something an environment variable of a
// production library may cause).
#ifdef GJX_PROFILE_HOOKS
#define GJX_DBG_STOP(A, k, ...) if (((A).debug_stop & 15) == (k)) { __VA_ARGS__; return; }
#else
#define GJX_DBG_STOP(A, k, ...)
#endif
// r04: what differs between two runs of the same filter over the same buffers — a step's keys, its observation, the comb
// offset, the model's scalar parameters — read from DEVICE memory instead of from the kernel's arguments, so that the T
// launches of a whole run are one hipGraph that any later run replays after one small copy (gjx_hip.hip: RunGraphs).
struct StepParams {
  uint32_t k0, k1;   // the step's key
  uint32_t y_bits;   // the step's observation (f32 bits / category)
  uint32_t pad;
  double u0;         // comb offset of the step's resampling
};
struct ResampleArgs {
  const StepParams* sp = nullptr;       // nullable: this step's entry (then u0 and the policy's key / observation come from it)
  const float* rp = nullptr;            // nullable: the run's model parameters (policy-defined layout)
  const uint32_t* qw = nullptr;         // [n] tile-anchored fixed-point weights of the SOURCE population
  const float* lw = nullptr;            // [n] source log-weights (adaptive filters: a kept step accumulates them)
  const TileRec* recs = nullptr;        // [ntiles] source records
  const TileSub* subs = nullptr;        // [ntiles] their sub-prefixes
  const TileEss* ess = nullptr;         // [ntiles] their ESS sums (adaptive filters)
  uint64_t n = 0, ntiles = 0;
  uint64_t n_out = 0;                   // number of comb teeth (global output slots)
  int64_t out_lo = 0, out_hi = 0;       // slots this launch serves (out_lo a multiple of the tile size)
  double u0 = 0.0;                      // the comb offset: u0_from_bits(64-bit draw of the resampling key), from the host
  int32_t* e_out = nullptr;             // nullable: workgroup 0 stores the merged anchor of the source weights
  uint64_t* q_out = nullptr;            // nullable: ... and their total mass
  int32_t* resampled_out = nullptr;     // nullable: workgroup 0 of each filter stores 1 (resampled) / 0 (kept)
  const uint64_t* prefix = nullptr;     // nullable: [prefix_words(ntiles)] (k_scan_records): filter batches, populations beyond kMaxGroups groups
  const GroupRec* groups = nullptr;     // nullable (r04): [ceil(ntiles / kGroupTiles)] group records: the route of populations beyond kMaxLdsTiles
  FilterBatch fb;                       // several filters per launch (n, ntiles, n_out, out_lo/out_hi are then PER FILTER)
  double ess_thr = 0.0;                 // threshold * n_total, 0 = resample always
  // what the step emits for the NEXT resampling (policies with weights)
  uint32_t* qw_out = nullptr;           // [n_local] (slot - out_lo)
  float* logw_out = nullptr;            // nullable [n_local]
  TileRec* recs_out = nullptr;          // GLOBAL [tiles of n_out]: entry of every output tile served
  TileSub* subs_out = nullptr;          // GLOBAL, likewise
  TileEss* ess_out = nullptr;           // GLOBAL, likewise (adaptive filters)
  int scan_max = kScanMax;              // rounds of the window scan (test knob: 0 = every output tile takes the per-slot search)
  int debug_stop = 0;                   // profiling builds only (-DGJX_PROFILE_HOOKS, GJX_SMC_DEBUG_STOP): leave the kernel after phase k
  int xcd_map = 1;                      // contiguous output tiles per XCD (GJX_SMC_XCD_MAP=0: plain order)
  int wt_stores = 0;                    // write-through stores of the step's output columns (store16_out)
  int wave_route = 1;                   // populations of up to kWave * 4 tiles: every WAVE merges the records itself (GJX_SMC_WAVE_ROUTE=0: LDS route)
  PeerMap pm;                           // r04: the source population is distributed over peers (PEERS instantiations only)
};

// resample iff ESS = R1^2 / R2 < thr (thr in particles); every backend evaluates exactly these double operations
GJX_HD bool ess_says_resample(uint64_t r1, uint64_t r2, double thr) {
  if (!(thr > 0.0) || r2 == 0) return true;
  const double a = (double)r1 * (double)r1;
  const double b = thr * (double)r2;
  return a < b;
}
// the reduced weight of the ESS sums: the top 16 bits of the tile-anchored fixed-point weight
GJX_HD uint64_t ess_r(uint64_t q) { return q >> kEssShift; }

GJX_HD double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }
// ---- the comb: every backend evaluates exactly these float64 operations ------------------------------
// teeth (j + u0), j in [0, n_out), strictly below a position: ceil, clamped to [0, n_out] (a NaN counts as 0)
GJX_HD int32_t comb_clamp(double t, int32_t n_out) {
  const double c = __builtin_ceil(t);
#if defined(__HIP_DEVICE_COMPILE__)
  // the same function without branches (r04: the step kernel evaluates it a dozen times per lane; as two early returns it
  // compiled to two divergent branches each): fmax drops a NaN and anything <= 0 to 0, fmin caps at n_out, the conversion
  // of an integral double in [0, n_out] is exact
  return (int32_t)__builtin_fmin(__builtin_fmax(c, 0.0), (double)n_out);
#else
  if (!(c > 0.0)) return 0;
  if (c >= (double)n_out) return n_out;
  return (int32_t)c;
#endif
}
// the position of a tile's start: P_t scale - u0 (two roundings)
GJX_HD double comb_base(uint64_t P, double scale, double u0) { return (double)P * scale - u0; }
// teeth below the start of the tile whose exclusive mass prefix is P
GJX_HD int32_t comb_tile(uint64_t P, double scale, double u0, int32_t n_out) { return comb_clamp(comb_base(P, scale, u0), n_out); }
// teeth below a particle inside a tile: c = running sum of q up to and including it (exact in float64), scale_t =
// scale 2^-d_t, base = comb_base of the tile, nhi = teeth below the tile's end (the cap keeps tiles consistent)
GJX_HD int32_t comb_in_tile(double c, double scale_t, double base, int32_t nhi, int32_t n_out) {
#if defined(__HIP_DEVICE_COMPILE__)
  // (0 <= nhi <= n_out: the cap at nhi is the only upper clamp needed)
  return (int32_t)__builtin_fmin(__builtin_fmax(__builtin_ceil(__builtin_fma(c, scale_t, base)), 0.0), (double)nhi);
#else
  const int32_t t = comb_clamp(__builtin_fma(c, scale_t, base), n_out);
  return t < nhi ? t : nhi;
#endif
}
GJX_HD double comb_tile_scale(double scale, int d) { return d >= 64 ? 0.0 : scale * u2d((uint64_t)(1023 - d) << 52); }
// ------------------------------------------------------------------------------------------------
// Row-anchored log-sum-exp of a whole pass: e = max e_b; buckets B_d = sum of S_b over
// the rows with e - e_b == d (d < 64, exact); Q = sum_d B_d >> d; lse = e ln2 + log(Q 2^-30).  The
// (e, B_0..B_63) record is what ranks exchange: bucket sums are exact integers, so records merge
// (k_lse_combine) into the same bits for any sharding.
// ------------------------------------------------------------------------------------------------
GJX_DEV void lse_emit(int32_t e, uint64_t bucket, int32_t* out_e, uint64_t* out_q, float* out_lse,
                      uint64_t* out_record, float* out_shifted = nullptr, float shift = 0.0f) {
  // called by the first wave: lane d holds bucket d
  const uint64_t q = wave_sum(bucket >> (threadIdx.x & 63));
  if (out_record) {
    out_record[1 + threadIdx.x] = bucket;
    if (threadIdx.x == 0) out_record[0] = (uint64_t)(int64_t)e;
  }
  if (threadIdx.x == 0) {
    if (out_e) out_e[0] = e;
    if (out_q) out_q[0] = q;
    if (out_lse || out_shifted) {
      float lse = -__builtin_inff();
      if (!(e == kRowEmpty || q == 0)) {
        const float t1 = (float)e * 0.69314718055994531f;
        const float t2 = m_log((float)q * u2f((uint32_t)(127 - kRowFrac) << 23));
        lse = t1 + t2;
      }
      if (out_lse) out_lse[0] = lse;
      if (out_shifted) out_shifted[0] = lse - shift;
    }
  }
}
// One 256-thread workgroup folds n_rows (e_b, S_b) pairs.  Up to 16 rows per thread are held in
// registers (all loads in flight at once: one memory latency), which covers 4096 rows = 1M particles;
// larger populations take the two-pass loop.  DEVICE_SCOPE: the pairs were written by other
// workgroups of the SAME launch, so they are read with agent-scope (sc1) loads that bypass this CU's L1.
template <bool DEVICE_SCOPE>
GJX_DEV int32_t lse_load_e(const int32_t* p) {
  return DEVICE_SCOPE ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
template <bool DEVICE_SCOPE>
GJX_DEV uint64_t lse_load_s(const uint64_t* p) {
  return DEVICE_SCOPE ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
template <bool DEVICE_SCOPE>
GJX_DEV void lse_rows_block(const int32_t* row_e, const uint64_t* row_s, uint64_t n_rows,
                            int32_t* out_e, uint64_t* out_q, float* out_lse, uint64_t* out_record,
                            float* out_shifted = nullptr, float shift = 0.0f) {
  __shared__ int32_t she[kBlock / kWave];
  const int nthr = (int)blockDim.x, nwave = nthr / kWave;  // 256 (k_lse_rows, generic tail) or 128 (paired kernel)
  __shared__ unsigned long long shb[kLseBuckets];
  constexpr int kPer = 16;
  constexpr int kNear = 4;  // shifts 0..3 (practically every row) accumulate in registers
  const bool in_regs = n_rows <= (uint64_t)kPer * nthr;
  int32_t ev[kPer];
  uint64_t sv[kPer];
  int32_t e = kRowEmpty;
  if (threadIdx.x < kLseBuckets) shb[threadIdx.x] = 0;
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < kPer; ++k) {
      const uint64_t b = threadIdx.x + (uint64_t)k * nthr;
      ev[k] = b < n_rows ? lse_load_e<DEVICE_SCOPE>(row_e + b) : kRowEmpty;
      sv[k] = b < n_rows ? lse_load_s<DEVICE_SCOPE>(row_s + b) : 0;
    }
#pragma unroll
    for (int k = 0; k < kPer; ++k) e = ev[k] > e ? ev[k] : e;
  } else {  // kPer independent loads in flight per lane and trip (a one-wave workgroup folding 4k rows one dependent load at
            // a time, each a miss behind the acquire, was +34 us on a 21 us launch)
    for (uint64_t base = 0; base < n_rows; base += (uint64_t)kPer * nthr) {
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        const uint64_t b = base + threadIdx.x + (uint64_t)k * nthr;
        ev[k] = b < n_rows ? lse_load_e<DEVICE_SCOPE>(row_e + b) : kRowEmpty;
      }
#pragma unroll
      for (int k = 0; k < kPer; ++k) e = ev[k] > e ? ev[k] : e;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int32_t o = __shfl_xor(e, off, kWave);
    e = o > e ? o : e;
  }
  if ((threadIdx.x & 63) == 0) she[threadIdx.x >> 6] = e;
  __syncthreads();
  e = she[0];
  for (int i = 1; i < nwave; ++i) e = she[i] > e ? she[i] : e;
  uint64_t near[kNear] = {0, 0, 0, 0};
  auto add_row = [&](int32_t eb, uint64_t sb) {
    if (eb == kRowEmpty) return;
    const int64_t d = (int64_t)e - (int64_t)eb;
#pragma unroll
    for (int k = 0; k < kNear; ++k) near[k] += d == k ? sb : 0;
    if (d >= kNear && d < kLseBuckets && sb) atomicAdd(&shb[d], (unsigned long long)sb);
  };
  if (in_regs) {
#pragma unroll
    for (int k = 0; k < kPer; ++k) add_row(ev[k], sv[k]);
  } else {
    for (uint64_t base = 0; base < n_rows; base += (uint64_t)kPer * nthr) {
#pragma unroll
      for (int k = 0; k < kPer; ++k) {
        const uint64_t b = base + threadIdx.x + (uint64_t)k * nthr;
        ev[k] = b < n_rows ? lse_load_e<DEVICE_SCOPE>(row_e + b) : kRowEmpty;
        sv[k] = b < n_rows ? lse_load_s<DEVICE_SCOPE>(row_s + b) : 0;
      }
#pragma unroll
      for (int k = 0; k < kPer; ++k) add_row(ev[k], sv[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < kNear; ++k) {
    const uint64_t w = wave_sum(near[k]);
    if ((threadIdx.x & 63) == 0 && w) atomicAdd(&shb[k], (unsigned long long)w);
  }
  __syncthreads();
  if (threadIdx.x < kLseBuckets) lse_emit(e, (uint64_t)shb[threadIdx.x], out_e, out_q, out_lse, out_record, out_shifted, shift);
}

// The log-marginal of a pass fused into the kernel that produces the log-weights (mirrors gjx_lse_out):
// every workgroup publishes its row pairs, takes a ticket, and the workgroup that arrives LAST folds all
// of them — no second launch.  Hand-off form (MI355X: per-XCD L2s are not coherent, L1 is never
// refreshed by other CUs' stores): the pairs are written by ONE lane with agent-scope (sc1, write-through)
// stores, that lane drains them (s_waitcnt vmcnt(0)) and then takes the ticket with an agent-scope atomic
// add; the last arriver — told by the value its own add returned — reads them with sc1 loads after a
// workgroup barrier.  Tickets are sharded 16 ways (+1 top word) so 4k arrivals do not queue on one word,
// and the last workgroup leaves them zero for the next launch on the stream.
constexpr int kLseTicketShards = 16;
constexpr int kLseTicketStride = 64;  // words between counters: each on its own 256-byte line (atomics on one
                                      // line are served one at a time by the memory-side atomic unit)
struct LseTail {
  int32_t* e;
  uint64_t* q;
  float* lse;
  uint64_t* record;
  uint32_t* tickets;  // [(kLseTicketShards + 1) * kLseTicketStride], zero between launches; null = no fused tail
  float* lse_shifted; // nullable: lse - shift
  float shift;
};
GJX_DEV void lse_store_row(int32_t* row_e, uint64_t* row_s, uint64_t row, int32_t eb, uint64_t sb, bool device_scope) {
  if (device_scope) {  // read by the last workgroup of THIS launch: write-through
    __hip_atomic_store(row_e + row, eb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(row_s + row, sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {  // read by a later launch
    row_e[row] = eb;
    row_s[row] = sb;
  }
}
// Called by every thread of every workgroup once, after its last lse_store_row.
GJX_DEV void lse_tail(const int32_t* row_e, const uint64_t* row_s, uint64_t n_rows, const LseTail& t) {
  if (!t.tickets) return;
  __shared__ uint32_t sh_last;
  // every wave drains its own stores (a workgroup of several one-wave rows has a storing lane in each wave), then the
  // workgroup's barrier, then ONE ticket
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t shard = blockIdx.x % kLseTicketShards;
    const uint32_t in_shard = (gridDim.x - shard + kLseTicketShards - 1) / kLseTicketShards;
    uint32_t last = 0;
    if (__hip_atomic_fetch_add(t.tickets + shard * kLseTicketStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ==
        in_shard - 1) {
      const uint32_t n_shards = gridDim.x < (uint32_t)kLseTicketShards ? gridDim.x : (uint32_t)kLseTicketShards;
      last = __hip_atomic_fetch_add(t.tickets + kLseTicketShards * kLseTicketStride, 1u, __ATOMIC_RELAXED,
                                    __HIP_MEMORY_SCOPE_AGENT) ==
                     n_shards - 1
                 ? 1u
                 : 0u;
    }
    sh_last = last;
  }
  __syncthreads();
  if (!sh_last) return;
  // every pair was written through (sc1) and drained before its writer's ticket: ONE agent-scope acquire (invalidates this
  // CU's L1 and the XCD's non-local L2 lines) and the fold reads them with ordinary, pipelined loads — sixteen agent-scope
  // atomic loads per lane, one after the other, were +34 us on the 21 us kernel
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  lse_rows_block<false>(row_e, row_s, n_rows, t.e, t.q, t.lse, t.record, t.lse_shifted, t.shift);
  if (threadIdx.x <= kLseTicketShards)
    __hip_atomic_store(t.tickets + threadIdx.x * kLseTicketStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Kernel argument block of a plan-driven SMC step (the generated policy wraps it).
struct PlanPolicyArgs {
  const float* prev_state[4];
  float* state_out[4];
  int32_t* anc_out;
  Key step_key;
  float obs[8];
  int32_t wt;  // write-through stores of the state / ancestor columns (store16_out; the host sets it for one-filter launches)
};

template <int N>
struct IntC {
  static constexpr int value = N;
};

// A resampling policy serves the four consecutive output slots of a lane either at once (compute_quad /
// store_quad: shared cipher block, vector stores) or slot by slot (compute / store: generated policies).  `anc` are
// GLOBAL source indices: the policy gathers its ancestors' state itself.  prefetch (optional) is called before the
// kernel's first wait on memory: whatever does not depend on the ancestors (cipher blocks, Box-Muller) goes there.
template <class Policy>
GJX_DEV auto policy_prefetch(Policy& P, int64_t jq, int) -> decltype(P.prefetch(jq), void()) { P.prefetch(jq); }
template <class Policy>
GJX_DEV void policy_prefetch(Policy&, int64_t, long) {}
// stage (optional): what prefetch loaded goes to LDS; called once by every thread, a workgroup barrier before compute
template <class Policy>
GJX_DEV auto policy_stage(Policy& P, int) -> decltype(P.stage(), void()) { P.stage(); }
template <class Policy>
GJX_DEV void policy_stage(Policy&, long) {}
// set_peers (policies of PEERS instantiations): where the byte offsets of the peers' arenas are (LDS) and how many tiles a
// rank owns — the policy gathers its ancestors' state where it lives
template <class Policy>
GJX_DEV auto policy_set_peers(Policy& P, const int64_t* pd, uint32_t tpr, int) -> decltype(P.set_peers(pd, tpr), void()) { P.set_peers(pd, tpr); }
template <class Policy>
GJX_DEV void policy_set_peers(Policy&, const int64_t*, uint32_t, long) {}
// a 4-byte element of a source column: index i of the GLOBAL population, read from its owner's arena
template <bool PEERS, class T>
GJX_DEV T src_load(const T* base, uint32_t i, const int64_t* pd, uint32_t tpr) {
  if (PEERS) return *peer_ptr(base + i, pd[(i >> 10) / tpr]);
  return base[i];
}
template <class Policy>
GJX_DEV auto policy_compute_quad(Policy& P, int64_t jq, const uint32_t (&anc)[4], typename Policy::Out (&o)[4], float (&w)[4],
                                 int) -> decltype(P.compute_quad(jq, anc, o, w), void()) {
  P.compute_quad(jq, anc, o, w);
}
template <class Policy>
GJX_DEV void policy_compute_quad(Policy& P, int64_t jq, const uint32_t (&anc)[4], typename Policy::Out (&o)[4], float (&w)[4],
                                 long) {
#pragma unroll
  for (int u = 0; u < 4; ++u) w[u] = P.compute(jq + u, anc[u], o[u]);
}
template <class Policy>
GJX_DEV auto policy_store_quad(Policy& P, int64_t jq, int64_t out_lo, const uint32_t (&anc)[4],
                               const typename Policy::Out (&o)[4], const bool (&ok)[4], int)
    -> decltype(P.store_quad(jq, out_lo, anc, o, ok), void()) {
  P.store_quad(jq, out_lo, anc, o, ok);
}
template <class Policy>
GJX_DEV void policy_store_quad(Policy& P, int64_t jq, int64_t out_lo, const uint32_t (&anc)[4],
                               const typename Policy::Out (&o)[4], const bool (&ok)[4], long) {
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (ok[u]) P.store(jq + u, out_lo, anc[u], o[u]);
}

// 16-byte store of a step's output column; `wt`: WRITE-THROUGH (sc1) — the bytes leave the L2 as they are produced instead of
// in the write-back at the kernel's end (the next step's workgroups, on any XCD, read them from memory either way).
// Measured (r03), one filter of 1e6 particles: LGSSM step 13.17 -> 12.59 us, HMM 13.73 -> 13.22; with 16 filters per launch
// the write-back wins (127 vs 132 us: the next step finds part of its input in the L2s), so the host sets `wt` for
// one-filter launches only (ResampleArgs::wt_stores; GJX_SMC_WT=0|1 forces it).
GJX_DEV void store16_out(void* p, uint4 v, bool wt) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (wt) {
    // r04: a compiler BUILTIN carries the cache policy (rounds 1-3: inline asm `global_store_dwordx4 ... sc1` + a hand-placed
    // `s_nop`, because the hazard recogniser does not look inside asm — a correctness hazard owned by a comment).  The raw
    // buffer store takes a wave-uniform base (descriptor in SGPRs) and a per-lane byte offset: the base is the first active
    // lane's address minus 2^30, so lanes up to 1 GiB either side of it are addressable (the callers' lanes lie within a tile);
    // aux bit 4 = sc1 on gfx94x/gfx950 (`buffer_store_dwordx4 ... offen sc1`), the instruction the asm spelled.
    typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
    v4u_t x;
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    const uint64_t a = (uint64_t)(uintptr_t)p;
    // (the builtin returns int: through uint32_t, or the low word's bit 31 would sign-extend into the high word)
    const uint64_t first = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32)) << 32) |
                           (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    const uint64_t base = first - (1ull << 30);
    const uint32_t off = (uint32_t)(a - base);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(x, rs, (int)off, 0, 16 /* sc1 */);
    return;
  }
#endif
  *reinterpret_cast<uint4*>(p) = v;
}

// ---- emission: the fixed-point weights and the record of the tile a workgroup has just produced -------------------
// w[r], ok[r]: the log-weights of the thread's four consecutive slots (tile offset 4 tid + r) and whether the slot
// exists.  qw_at / logw_at: where the thread's first slot goes (logw_at nullable); rec_at: the tile's record.
// Called by every thread of the workgroup (two barriers inside).
template <bool ESS>
GJX_DEV void emit_tile(const float (&w)[kPer], const bool (&ok)[kPer], uint32_t* qw_at, float* logw_at, TileRec* rec_at,
                       TileSub* sub_at, TileEss* ess_at, bool wt = false) {
  constexpr int kW = kBlock / kWave;
  __shared__ uint64_t em_q[3 * kW];
  __shared__ int32_t em_e[kW];
  static_assert(kPer == 4 && kSubLen == 64, "four consecutive slots per lane: a 64-particle block is one DPP row of 16 lanes");
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  float tm = -__builtin_inff();
#pragma unroll
  for (int r = 0; r < kPer; ++r) tm = ok[r] && w[r] > tm ? w[r] : tm;
  // the tile's anchor row_anchor(max w): row_anchor is monotone, so the maximum is taken over the lanes' ANCHORS (int32: one
  // v_max_i32 with a DPP operand per step, where the float maximum took a compare and a select)
  int32_t te = row_anchor(tm);
  te = (int32_t)(wave_last_u32(wave_scan_u32((uint32_t)te ^ 0x80000000u, 0u, [](uint32_t a, uint32_t x) { return x > a ? x : a; })) ^ 0x80000000u);
  if (lane == 0) em_e[wv] = te;
  __syncthreads();
  int32_t e = em_e[0];
#pragma unroll
  for (int i = 1; i < kW; ++i) e = em_e[i] > e ? em_e[i] : e;
  uint32_t q[kPer];
  uint64_t run = 0, a1 = 0, a2 = 0;
#pragma unroll
  for (int r = 0; r < kPer; ++r) {
    q[r] = ok[r] ? (uint32_t)rowfix(w[r], e) : 0u;
    run += q[r];
    if (ESS) {
      const uint64_t rr = ess_r(q[r]);
      a1 += rr;
      a2 += rr * rr;
    }
  }
  const uint64_t incl = wave_scan_incl_52(run);  // (four weights: below 2^34)
  if (ESS) { a1 = wave_sum_52(a1); a2 = wave_sum_52(a2); }  // (four reduced weights below 2^17.5: sums below 2^20 / 2^37 per lane)
  if (lane == 63) {
    em_q[wv] = incl;
    if (ESS) { em_q[kW + wv] = a1; em_q[2 * kW + wv] = a2; }
  }
  __syncthreads();
  uint64_t base = 0;
#pragma unroll
  for (int i = 0; i < kW; ++i)
    if (i < wv) base += em_q[i];
  const bool all = ok[0] && ok[1] && ok[2] && ok[3];
  if (all && (((uintptr_t)qw_at & 15) == 0)) {
    store16_out(qw_at, make_uint4(q[0], q[1], q[2], q[3]), wt);
  } else {
#pragma unroll
    for (int r = 0; r < kPer; ++r)
      if (ok[r]) qw_at[r] = q[r];
  }
  if (logw_at) {
    if (all && (((uintptr_t)logw_at & 15) == 0)) {
      store16_out(logw_at, make_uint4(f2u(w[0]), f2u(w[1]), f2u(w[2]), f2u(w[3])), wt);
    } else {
#pragma unroll
      for (int r = 0; r < kPer; ++r)
        if (ok[r]) logw_at[r] = w[r];
    }
  }
  // the record: the last lane of every row of 16 lanes holds the running sum after its 64-particle block
  if ((lane & 15) == 15) sub_at->sub[tid >> 4] = base + incl;
  if (tid == kBlock - 1) {
    TileRec rec;
    rec.s = base + incl;
    rec.e = e;
    rec.pad = 0;
    *rec_at = rec;
    if (ESS && ess_at) {
      uint64_t t1 = 0, t2 = 0;
#pragma unroll
      for (int i = 0; i < kW; ++i) { t1 += em_q[kW + i]; t2 += em_q[2 * kW + i]; }
      ess_at->r1 = t1;
      ess_at->r2 = t2;
    }
  }
}

// Where a step's weights go for the NEXT resampling (kernel argument of the init kernels; the resample kernel carries
// the same fields in ResampleArgs): local weight / log-weight columns, the GLOBAL record array.
struct EmitOut {
  uint32_t* qw;    // [n_local]
  float* logw;     // nullable [n_local]
  TileRec* recs;   // GLOBAL [tiles]
  TileSub* subs;   // GLOBAL [tiles]
  TileEss* ess;    // nullable GLOBAL [tiles]: the filter is ESS-adaptive
};
GJX_DEV void emit_init_tile(const float (&w)[kPer], const bool (&ok)[kPer], const EmitOut& em, uint64_t loc, uint64_t gtile) {
  if (em.ess) emit_tile<true>(w, ok, em.qw + loc, em.logw ? em.logw + loc : nullptr, em.recs + gtile, em.subs + gtile, em.ess + gtile);
  else emit_tile<false>(w, ok, em.qw + loc, em.logw ? em.logw + loc : nullptr, em.recs + gtile, em.subs + gtile, nullptr);
}
GJX_DEV void select_filter_emit(EmitOut& em, const FilterBatch& fb, uint32_t f) {
  em.qw += (uint64_t)f * fb.stride;
  if (em.logw) em.logw += (uint64_t)f * fb.stride;
  em.recs += (uint64_t)f * fb.tiles;
  em.subs += (uint64_t)f * fb.tiles;
  if (em.ess) em.ess += (uint64_t)f * fb.tiles;
}

// Exclusive block max-scan of one u32 per thread (identity 0).  `sh` needs 4 words.  `sh_free`: the caller vouches that no
// thread can still be reading `sh` from an earlier use (a barrier has passed since), so the guarding barrier is left out.
GJX_DEV uint32_t block_scan_umax_excl(uint32_t v, uint32_t* sh, bool sh_free = false) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t incl = wave_scan_u32(v, 0u, [](uint32_t a, uint32_t b) { return b > a ? b : a; });
  uint32_t excl = dpp_u32<kDppWaveShr1, 0xf, 0xf>(0u, incl);  // lane i <- lane i-1, lane 0 keeps 0
  if (!sh_free) __syncthreads();
  if (lane == 63) sh[w] = incl;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kBlock / kWave; ++i)
    if (i < w) excl = sh[i] > excl ? sh[i] : excl;
  return excl;
}

// Inclusive scan of one u64 per lane inside groups of 8 consecutive lanes (one 64-particle block: 8 sources per lane).
GJX_DEV uint64_t scan8_incl(uint64_t v) {
  const int l8 = threadIdx.x & 7;
  uint64_t t = dpp_u64<kDppRowShr1, 0xf, 0xf>(0, v);
  v += l8 >= 1 ? t : 0;
  t = dpp_u64<kDppRowShr2, 0xf, 0xf>(0, v);
  v += l8 >= 2 ? t : 0;
  t = dpp_u64<kDppRowShr4, 0xf, 0xf>(0, v);
  v += l8 >= 4 ? t : 0;
  return v;
}

// A workgroup barrier that orders LDS only: the global loads a wave has in flight stay in flight (__syncthreads waits for
// them too: on gfx9 loads and stores share vmcnt and the workgroup-scope fence covers both).
GJX_DEV void lds_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
#endif
}
// ... and the wave-level form: LDS written by one lane is read by another lane of the SAME wave (in order on the hardware;
// this keeps the compiler from moving the accesses)
GJX_DEV void wave_lds_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
#endif
}

// ADAPTIVE: the launch may be a step of an ESS-adaptive filter (A.ess_thr > 0): only then does the kernel carry the
// decision, the ESS sums and the keep-your-particle path (the every-step filters are compiled without them).
// PEERS (r04): the source population is distributed (A.pm): the kernel first waits for its peers, then reads remote tiles
// where they live.  Everything it does with what it read is the code of the single-device step.
// LDSP: the merged prefix of the source tiles lives in LDS (the LDS, wave and grouped routes: A.prefix == nullptr) — a
// compile-time fact inside the body, so that every read of it is an LDS read (as a run-time choice between LDS and the prefix
// array in memory the reads were flat loads through a selected pointer).
template <int IMPL, class Policy, bool ADAPTIVE, bool PEERS, bool LDSP>
GJX_DEV void resample_body_impl(const ResampleArgs& A, Policy& P) {
  constexpr int kW = kBlock / kWave;
  constexpr int kSrc = 8;                        // sources per lane and round of the window scan
  static_assert(kSrc == 8 && kSubLen == 64, "8 consecutive sources per lane: a 64-particle block is 8 lanes");
  __shared__ uint64_t sh_pre[kMaxLdsTiles + 1];  // merged exclusive tile prefix (populations up to kMaxLdsTiles tiles)
  __shared__ uint8_t sh_d[kMaxLdsTiles];         // every tile's shift to the merged anchor
  __shared__ uint32_t marks[kTile];              // run-start marks of the ancestor search
  __shared__ uint64_t sh_scan[3 * kW];
  __shared__ int32_t sh_e[kW];
  __shared__ uint32_t sh_u[kW];
  __shared__ uint32_t sh_klo, sh_cov[2];
  __shared__ uint64_t sh_gpre[kMaxGroups + 1];   // grouped route: merged exclusive prefix of the GROUP masses
  __shared__ int64_t sh_delta[PEERS ? kMaxPeers : 1];
  __shared__ uint32_t sh_peer_ok;
  static_assert(kPer == 4, "four consecutive output slots per lane");
  uint64_t b = blockIdx.x;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  if (PEERS) {
#if defined(__HIP_DEVICE_COMPILE__)
    // nothing of the source population is read before every peer has arrived (bounded; a timeout ends the launch)
    if (tid < kMaxPeers) sh_delta[tid] = tid < A.pm.world ? A.pm.delta[tid] : 0;
    // (wait_value 0: a launch in front of this one on the stream has already waited and acquired — the group-record launch
    // of a large population, or the transport's wait launch — and the kernel boundary hands that to this launch)
    if (wv == 0) {
      const bool ready = A.pm.wait_value == 0 || peer_wait_wave(A.pm);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the acquire's invalidate has completed before the barrier releases the others)
      if (lane == 0) sh_peer_ok = ready ? 1u : 0u;
    }
    __syncthreads();
    if (!sh_peer_ok) return;
    policy_set_peers(P, sh_delta, A.pm.tiles_per_rank, 0);
#endif
  }
  // this workgroup's filter: local views of the per-filter arrays, keys and results
  const uint32_t* qw_all = A.qw;
  const float* lw_all = A.lw;
  const TileRec* recs = A.recs;
  const TileSub* subs = A.subs;
  const TileEss* ess = A.ess;
  int32_t* e_out = A.e_out;
  uint64_t* q_out = A.q_out;
  int32_t* resampled_out = A.resampled_out;
  uint32_t* qw_out = A.qw_out;
  float* logw_out = A.logw_out;
  TileRec* recs_out = A.recs_out;
  TileSub* subs_out = A.subs_out;
  TileEss* ess_out = A.ess_out;
  const uint64_t* prefix = LDSP ? nullptr : A.prefix;
  double u0 = A.sp ? A.sp->u0 : A.u0;
  if (A.fb.n_filters > 1) {
    const uint32_t f = (uint32_t)(b / A.fb.tiles);
    b -= (uint64_t)f * A.fb.tiles;
    qw_all += (uint64_t)f * A.fb.stride;
    if (lw_all) lw_all += (uint64_t)f * A.fb.stride;
    recs += (uint64_t)f * A.fb.tiles;
    subs += (uint64_t)f * A.fb.tiles;
    if (ess) ess += (uint64_t)f * A.fb.tiles;
    if (e_out) e_out += (uint64_t)f * A.fb.mq_stride;
    if (q_out) q_out += (uint64_t)f * A.fb.mq_stride;
    if (resampled_out) resampled_out += (uint64_t)f * A.fb.mq_stride;
    if (qw_out) qw_out += (uint64_t)f * A.fb.stride;
    if (logw_out) logw_out += (uint64_t)f * A.fb.stride;
    if (recs_out) recs_out += (uint64_t)f * A.fb.tiles;
    if (subs_out) subs_out += (uint64_t)f * A.fb.tiles;
    if (ess_out) ess_out += (uint64_t)f * A.fb.tiles;
    if (prefix) prefix += (uint64_t)f * prefix_words(A.fb.tiles);
    u0 = A.fb.u0[f];
    P.select_filter((uint64_t)f * A.fb.stride, A.fb.step_key[f]);
  }
  const uint32_t tpr = PEERS ? A.pm.tiles_per_rank : 1u;
  // where tile k of the source population lives (its owner's arena when the population is distributed)
  auto src_subs = [&](uint64_t k) -> const TileSub* { return PEERS ? peer_ptr(subs + k, sh_delta[(uint32_t)k / tpr]) : subs + k; };
  auto src_qw = [&](uint64_t i) -> const uint32_t* { return PEERS ? peer_ptr(qw_all + i, sh_delta[(uint32_t)(i >> 10) / tpr]) : qw_all + i; };
  GJX_DBG_STOP(A, 15);  // (profiling: the launch / dispatch floor)
  const bool adaptive = ADAPTIVE && A.ess_thr > 0.0;
  if (A.xcd_map) {
    // XCD-aware tile order: workgroups are dealt to the 8 XCDs round-robin, and neighbouring output tiles read
    // overlapping source windows — give every XCD a CONTIGUOUS range of output tiles so that the overlap is served by its
    // own L2 instead of a second trip to memory (a bijection on the launch's tiles: same results)
    const uint64_t nwg = ((uint64_t)(A.out_hi - A.out_lo) + kTile - 1) / kTile;
    const uint64_t x = b & 7, c = nwg >> 3, r = nwg & 7;
    b = x * c + (x < r ? x : r) + (b >> 3);
  }
  const uint64_t ot = (uint64_t)A.out_lo / kTile + b;  // this workgroup's output tile (global index)
  const int64_t j0 = (int64_t)(ot * kTile);
  const int64_t j1 = j0 + (int64_t)kTile < A.out_hi ? j0 + (int64_t)kTile : A.out_hi;
  const int64_t jq = j0 + (int64_t)kPer * tid;
  const int32_t n_out = (int32_t)A.n_out;

  // ---- the source records (issued first; the policy's ancestor-independent work runs under their latency) --------
  // Three routes to the merged prefix of the source tiles' masses (launch-uniform):
  //   LDS      (ntiles <= kMaxLdsTiles): every workgroup merges ALL tile records itself;
  //   grouped  (r04, A.groups: larger populations — one rank of BASELINE configs[3] merges 7 816 records): the workgroup
  //            merges the GROUP records (k_group_records: one per 256 tiles, the group's mass for every shift of its
  //            anchor), finds the group that holds its first tooth, and then merges only the 1024 tile records from that
  //            group on, exactly as the LDS route does — same integers, no serial whole-population scan, no search of a
  //            prefix array in memory;
  //   prefix   (A.prefix: batches of >= 4 filters, and populations beyond kMaxGroups groups): a precomputed global prefix.
  //   wave     (r04, ntiles <= 64 * kC = 256 — BASELINE configs[2] and [4] have 64 tiles): the LDS route without its
  //            barriers.  Every WAVE merges all records itself (kC per lane: anchor, masses and prefix by DPP scans inside
  //            the wave) and finds the first source tile by one ballot; the four waves write the SAME prefix values to LDS
  //            and each reads only what it wrote itself.  Four workgroup barriers and the per-thread walk of the LDS route
  //            leave the step's dependent chain: same integers, same float64 comb.
  constexpr int kC = kMaxLdsTiles / kBlock;  // tiles per thread of the in-kernel merge
  const bool grouped = LDSP && A.groups != nullptr;  // (launch-uniform)
  constexpr bool lds_prefix = LDSP;               // (true for the LDS, the wave and the grouped route)
  const bool wave_route = lds_prefix && !grouped && A.wave_route != 0 && A.ntiles <= (uint64_t)(kWave * kC);  // (launch-uniform)
  const uint64_t ngroups = grouped ? (A.ntiles + kGroupTiles - 1) / kGroupTiles : 0;  // <= kMaxGroups = kBlock: one per thread
  // the tiles whose prefix will live in LDS: [k_base, k_base + nrange).  Grouped route: SPECULATIVELY the four groups around
  // the output tile's own position (ancestors stay near their slots unless the weights are very uneven), so that the tile
  // records are in flight together with the group records; the group merge then says whether the guess holds
  uint64_t k_base = 0, nrange = A.ntiles;
  if (grouped) {
    const uint64_t g_own = ot / kGroupTiles;
    k_base = (g_own > 0 ? g_own - 1 : 0) * kGroupTiles;
    if (k_base >= A.ntiles) k_base = (ngroups - 1) * kGroupTiles;
    nrange = A.ntiles - k_base < (uint64_t)kMaxLdsTiles ? A.ntiles - k_base : (uint64_t)kMaxLdsTiles;
  }
  uint64_t c_per = wave_route ? (nrange + kWave - 1) / kWave : (nrange + kBlock - 1) / kBlock;
  uint64_t k0 = (uint64_t)(wave_route ? lane : tid) * c_per;
  uint64_t rs_[kC], ev1[kC], ev2[kC];
  int32_t re_[kC];
  auto load_range = [&]() {  // the tile records of the range, contiguous per thread
#pragma unroll
    for (int i = 0; i < kC; ++i) {
      const uint64_t k = k_base + k0 + i;
      const bool in = (uint64_t)i < c_per && k0 + i < nrange;
      if (in) {
        const uint4 raw = *reinterpret_cast<const uint4*>(recs + k);
        rs_[i] = ((uint64_t)raw.y << 32) | raw.x;
        re_[i] = (int32_t)raw.z;
      } else {
        rs_[i] = 0;
        re_[i] = kRowEmpty;
      }
      if (adaptive && !grouped && in) {  // (one 16-byte load per tile, like the record)
        const uint4 er = *reinterpret_cast<const uint4*>(ess + k);
        ev1[i] = ((uint64_t)er.y << 32) | er.x;
        ev2[i] = ((uint64_t)er.w << 32) | er.z;
      } else {
        ev1[i] = 0;
        ev2[i] = 0;
      }
    }
  };
  // grouped route: thread g < ngroups holds group g's anchor and — speculatively — its masses for the shifts 0 and 1 of that
  // anchor (practically every group's anchor is the population's or one below; other shifts cost a second trip)
  int32_t ge = kRowEmpty;
  uint64_t gm0 = 0, gm1 = 0, g10 = 0, g11 = 0, g20 = 0, g21 = 0;
  if (grouped && (uint64_t)tid < ngroups) {
    const GroupRec* gr = A.groups + tid;
    ge = gr->e;
    gm0 = gr->mass[0]; gm1 = gr->mass[1];
    if (adaptive) { g10 = gr->r1[0]; g11 = gr->r1[1]; g20 = gr->r2[0]; g21 = gr->r2[1]; }
  }
  if (lds_prefix) load_range();
#pragma unroll
  for (int r = 0; r < kPer; ++r) marks[tid + r * kBlock] = 0;
  if (tid == 0) sh_klo = ~0u;
  if (wave_route) lds_barrier();  // (the marks are clear before any wave's scan sets one; the record loads stay in flight)
  policy_prefetch(P, jq, 0);  // (under the latency of the record loads)
  GJX_DBG_STOP(A, 14, if (lds_prefix && rs_[0] == 0x123456789abcdefull && re_[0] == 77) marks[0] = 1);  // (records loaded, prefetch done, nothing merged)

  // ---- merge: anchor, shifted masses, exclusive prefix, total, ESS sums ------------------------------------------
  int32_t e = kRowEmpty;
  uint64_t tot = 0, r1 = 0, r2 = 0;
  uint64_t mass[kC];
  uint64_t chunk_pre = 0, chunk_mass = 0;
  uint64_t p_base = 0;  // mass before tile k_base (grouped route)
  // the workgroup's anchor: max of one int32 per thread
  auto block_anchor = [&](int32_t mine) -> int32_t {
    int32_t m = (int32_t)(wave_last_u32(wave_scan_u32((uint32_t)mine ^ 0x80000000u, 0u, [](uint32_t a, uint32_t x) { return x > a ? x : a; })) ^ 0x80000000u);
    if (lane == 0) sh_e[wv] = m;
    __syncthreads();
    m = sh_e[0];
#pragma unroll
    for (int i = 1; i < kW; ++i) m = sh_e[i] > m ? sh_e[i] : m;
    return m;
  };
  if (grouped) {
    // ---- group level: the anchor is the maximum of the groups' anchors; a group's mass under it is entry (e - e_g) of the
    // group's table (exact: the table holds the sum of the group's per-tile shifted masses for every shift)
    e = block_anchor(ge);
    const int dg = tile_shift(e, ge);
    uint64_t gm = dg == 0 ? gm0 : gm1, l1 = dg == 0 ? g10 : g11, l2 = dg == 0 ? g20 : g21;
    if (dg >= 64 || (uint64_t)tid >= ngroups) {
      gm = 0; l1 = 0; l2 = 0;
    } else if (dg > 1) {  // (rare: a group whose best weight is 4x or more below the population's)
      const GroupRec* gr = A.groups + tid;
      gm = gr->mass[dg];
      if (adaptive) { l1 = gr->r1[dg]; l2 = gr->r2[dg]; }
    }
    const uint64_t incl = wave_scan_incl(gm);
    if (adaptive) { l1 = wave_sum(l1); l2 = wave_sum(l2); }
    if (lane == 63) {
      sh_scan[wv] = incl;
      if (adaptive) { sh_scan[kW + wv] = l1; sh_scan[2 * kW + wv] = l2; }
    }
    __syncthreads();
    uint64_t wbase = 0;
#pragma unroll
    for (int i = 0; i < kW; ++i) {
      if (i < wv) wbase += sh_scan[i];
      tot += sh_scan[i];
      if (adaptive) { r1 += sh_scan[kW + i]; r2 += sh_scan[2 * kW + i]; }
    }
    if ((uint64_t)tid < ngroups) sh_gpre[tid] = wbase + incl - gm;
    if (tid == 0) sh_gpre[ngroups] = tot;
    __syncthreads();
    // ---- the group that holds the first tooth of [j0, j1): the first group whose END has more than j0 teeth below it
    if (tot != 0 && (!adaptive || ess_says_resample(r1, r2, A.ess_thr))) {
      const double gscale = (double)A.n_out / (double)tot;
      if ((uint64_t)tid < ngroups) {
        const uint64_t g = tid;
        const int32_t hi_g = g + 1 >= ngroups ? n_out : comb_tile(sh_gpre[g + 1], gscale, u0, n_out);
        const int32_t lo_g = comb_tile(sh_gpre[g], gscale, u0, n_out);
        if (hi_g > j0 && (g == 0 || lo_g <= j0)) atomicMin(&sh_klo, (uint32_t)g);
      }
      __syncthreads();
      const uint64_t g_lo = sh_klo < ngroups ? sh_klo : ngroups - 1;
      __syncthreads();
      if (tid == 0) sh_klo = ~0u;
      // the guess holds when the range starts at or before g_lo and keeps >= two groups (512 tiles; a window is < 8) behind its
      // start — or reaches the population's end; otherwise the range is re-loaded from g_lo on (a dependent trip, rare)
      const uint64_t kb = g_lo * kGroupTiles;
      if (!(k_base <= kb && (kb + 2 * kGroupTiles <= k_base + nrange || k_base + nrange >= A.ntiles))) {
        k_base = kb;
        nrange = A.ntiles - k_base < (uint64_t)kMaxLdsTiles ? A.ntiles - k_base : (uint64_t)kMaxLdsTiles;
        c_per = (nrange + kBlock - 1) / kBlock;
        k0 = (uint64_t)tid * c_per;
        load_range();
      }
    }
    p_base = sh_gpre[k_base / kGroupTiles];
  }
  if (wave_route) {
#pragma unroll
    for (int i = 0; i < kC; ++i) e = re_[i] > e ? re_[i] : e;
    e = (int32_t)(wave_last_u32(wave_scan_u32((uint32_t)e ^ 0x80000000u, 0u, [](uint32_t a, uint32_t x) { return x > a ? x : a; })) ^ 0x80000000u);
    uint64_t l1 = 0, l2 = 0;
    int dsh[kC];
#pragma unroll
    for (int i = 0; i < kC; ++i) {
      dsh[i] = tile_shift(e, re_[i]);
      mass[i] = shr64(rs_[i], dsh[i]);
      chunk_mass += mass[i];
      if (adaptive) { l1 += shr64(ev1[i], dsh[i]); l2 += shr64(ev2[i], 2 * dsh[i]); }
    }
    const uint64_t incl = wave_scan_incl_52(chunk_mass);  // (at most 4 tile masses: below 2^44)
    tot = wave_last_u64(incl);
    if (adaptive) { r1 = wave_sum_52(l1); r2 = wave_sum_52(l2); }  // (per lane: 4 tiles of r1 < 2^28, r2 < 2^45)
    chunk_pre = incl - chunk_mass;
    uint64_t run = chunk_pre;
#pragma unroll
    for (int i = 0; i < kC; ++i) {
      if ((uint64_t)i < c_per && k0 + i < nrange) {
        sh_pre[k0 + i] = run;
        sh_d[k0 + i] = (uint8_t)dsh[i];
        run += mass[i];
      }
    }
    if (lane == 0) sh_pre[nrange] = tot;
    wave_lds_fence();
  } else if (lds_prefix) {
    if (!grouped) {
#pragma unroll
      for (int i = 0; i < kC; ++i) e = re_[i] > e ? re_[i] : e;
      e = block_anchor(e);
    }
    uint64_t l1 = 0, l2 = 0;
    int dsh[kC];
#pragma unroll
    for (int i = 0; i < kC; ++i) {
      dsh[i] = tile_shift(e, re_[i]);
      mass[i] = shr64(rs_[i], dsh[i]);
      chunk_mass += mass[i];
      if (adaptive && !grouped) { l1 += shr64(ev1[i], dsh[i]); l2 += shr64(ev2[i], 2 * dsh[i]); }
    }
    const uint64_t incl = wave_scan_incl_52(chunk_mass);  // (at most 4 tile masses: below 2^44)
    if (adaptive && !grouped) { l1 = wave_sum_52(l1); l2 = wave_sum_52(l2); }  // (per lane: 4 tiles of r1 < 2^28, r2 < 2^45)
    if (grouped) __syncthreads();  // (sh_scan was read by the group level)
    if (lane == 63) {
      sh_scan[wv] = incl;
      if (adaptive && !grouped) { sh_scan[kW + wv] = l1; sh_scan[2 * kW + wv] = l2; }
    }
    __syncthreads();
    uint64_t wbase = 0, range_tot = 0;
#pragma unroll
    for (int i = 0; i < kW; ++i) {
      if (i < wv) wbase += sh_scan[i];
      range_tot += sh_scan[i];
      if (adaptive && !grouped) { r1 += sh_scan[kW + i]; r2 += sh_scan[2 * kW + i]; }
    }
    if (!grouped) tot = range_tot;
    chunk_pre = p_base + wbase + incl - chunk_mass;
    uint64_t run = chunk_pre;
#pragma unroll
    for (int i = 0; i < kC; ++i) {
      if ((uint64_t)i < c_per && k0 + i < nrange) {
        sh_pre[k0 + i] = run;
        sh_d[k0 + i] = (uint8_t)dsh[i];
        run += mass[i];
      }
    }
    if (tid == 0) sh_pre[nrange] = p_base + range_tot;
  } else {
    tot = prefix[A.ntiles];
    e = (int32_t)(int64_t)prefix[A.ntiles + 1];
    if (adaptive) { r1 = prefix[A.ntiles + 2]; r2 = prefix[A.ntiles + 3]; }
  }
  if (b == 0 && tid == 0) {
    if (e_out) e_out[0] = e;
    if (q_out) q_out[0] = tot;
  }
  const bool resample = !adaptive || ess_says_resample(r1, r2, A.ess_thr);
  if (resampled_out && b == 0 && tid == 0) resampled_out[0] = resample ? 1 : 0;
  policy_stage(P, 0);  // (every path below passes a barrier before the policy computes)
  GJX_DBG_STOP(A, 1);

  // the merged prefix / shift of tile k: LDS for the tiles of the range (every tile on the LDS route), memory on the prefix route
  const uint64_t k_end = k_base + nrange;  // (LDS routes) one past the last tile whose prefix is in LDS
  auto pre_at = [&](uint64_t k) -> uint64_t { return lds_prefix ? sh_pre[k - k_base] : prefix[k]; };
  auto shift_at = [&](uint64_t k) -> int { return lds_prefix ? (int)sh_d[k - k_base] : tile_shift(e, recs[k].e); };

  uint32_t anc[kPer];
  bool ok[kPer];
#pragma unroll
  for (int r = 0; r < kPer; ++r) ok[r] = jq + r < j1;
  float lw_prev[kPer] = {0.0f, 0.0f, 0.0f, 0.0f};

  if (ADAPTIVE && !resample) {
    // ---- no resampling at this step: slot j keeps particle j, its log-weight accumulates -------------------------
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
      const uint64_t j = (uint64_t)jq + r;
      anc[r] = (uint32_t)(j < A.n ? j : A.n - 1);
      lw_prev[r] = ok[r] ? lw_all[anc[r]] : 0.0f;
    }
    __syncthreads();  // (the policy's staged data: a barrier before it computes)
  } else if (tot == 0) {
    // ---- no mass at all (every weight -inf / NaN / underflowed): the population is kept as it is — slot j takes
    // particle floor(j n / n_out) (the identity when n_out == n), the weights start afresh ---------------------------
    const double ratio = (double)A.n / (double)A.n_out;
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
      const uint64_t g = (uint64_t)__builtin_floor((double)(jq + r) * ratio);
      anc[r] = (uint32_t)(g < A.n ? g : A.n - 1);
    }
    __syncthreads();
  } else {
    const double scale = (double)A.n_out / (double)tot;
    auto nlo_of = [&](uint64_t k) -> int32_t { return comb_tile(pre_at(k), scale, u0, n_out); };  // teeth below the START of tile k
    auto nhi_of = [&](uint64_t k) -> int32_t { return k + 1 >= A.ntiles ? n_out : nlo_of(k + 1); };  // ... below its END
    // ---- the first source tile with a tooth in [j0, j1): k_lo = min{k : teeth below the END of tile k > j0} ----------
    uint64_t k_lo = 0;
    if (wave_route) {
      // (monotone: the first lane whose chunk of tiles ends beyond j0 holds k_lo — the last tile's end lies beyond every slot)
      const uint64_t r1_ = k0 + c_per < nrange ? k0 + c_per : nrange;
      const int32_t c_hi = r1_ >= A.ntiles ? n_out : comb_tile(chunk_pre + chunk_mass, scale, u0, n_out);
      const uint64_t hit = __ballot(k0 < nrange && c_hi > j0);
      const uint64_t first = (uint64_t)(hit ? __builtin_ctzll(hit) : 0) * c_per;
      k_lo = first;
      if (c_per > 1) {  // (wave-uniform walk of that lane's tiles: their prefix is in LDS)
        const uint64_t last = first + c_per < nrange ? first + c_per : nrange;
        for (k_lo = first; k_lo + 1 < last; ++k_lo)
          if (comb_tile(sh_pre[k_lo + 1], scale, u0, n_out) > j0) break;
      }
    } else if (lds_prefix) {
      __syncthreads();  // sh_pre / sh_d complete
      // r04: every WAVE finds k_lo on its own from the prefix in LDS — lane g looks at the END of the g-th group of 16 tiles
      // (at most 64 groups), a ballot names the first group whose end lies beyond j0, 16 lanes look at that group's tiles,
      // a second ballot names the tile.  Two LDS reads, two comb evaluations, no atomic and no barrier (r03: the one thread
      // whose 4-tile chunk brackets j0 walked it, an LDS atomic and a barrier handed the result round: 1.1 us of the step).
      // Monotone, so this is min{k in the range : teeth below the END of tile k > j0}, as before.
      const uint32_t nr = (uint32_t)nrange;
      const uint32_t ge = ((uint32_t)lane + 1u) * 16u < nr ? ((uint32_t)lane + 1u) * 16u : nr;  // one past group `lane`'s last tile
      const bool in_g = (uint32_t)lane * 16u < nr;
      const int32_t hi_g = !in_g ? 0 : (k_base + ge >= A.ntiles ? n_out : comb_tile(sh_pre[ge], scale, u0, n_out));
      const uint64_t hit = __ballot(in_g && hi_g > j0);
      const uint32_t g = hit ? (uint32_t)__builtin_ctzll(hit) : (nr - 1u) / 16u;
      const uint32_t kk = g * 16u + ((uint32_t)lane & 15u);
      const bool in_t = lane < 16 && kk < nr;
      const int32_t hi_t = !in_t ? 0 : (k_base + kk + 1 >= A.ntiles ? n_out : comb_tile(sh_pre[kk + 1], scale, u0, n_out));
      const uint64_t hit2 = __ballot(in_t && hi_t > j0);
      const uint32_t last = g * 16u + 15u < nr ? g * 16u + 15u : nr - 1u;
      k_lo = k_base + (hit2 ? g * 16u + (uint32_t)__builtin_ctzll(hit2) : last);
    } else if (A.ntiles <= (uint64_t)(kWave * 16)) {
      // a precomputed prefix of up to 1024 tiles (filter batches): ONE wave samples every 16th entry, then the 16 entries
      // of the bracket — two small loads instead of every thread reading its share of the whole prefix
      if (wv == 0) {
        const uint64_t ks = (uint64_t)lane * 16 + 15 < A.ntiles ? (uint64_t)lane * 16 + 15 : A.ntiles - 1;  // the END tile of group `lane`
        const bool in = (uint64_t)lane * 16 < A.ntiles;
        const uint64_t hit = __ballot(in && nhi_of(ks) > j0);  // (non-empty: the last tile's end lies beyond every slot)
        const uint64_t g = hit ? (uint64_t)__builtin_ctzll(hit) : 0;
        const uint64_t kk = g * 16 + (uint64_t)(lane & 15);
        const uint64_t hit2 = __ballot(lane < 16 && kk < A.ntiles && nhi_of(kk) > j0);
        if (lane == 0) sh_klo = (uint32_t)(g * 16 + (hit2 ? (uint64_t)__builtin_ctzll(hit2) : 0));
      }
      __syncthreads();
      k_lo = sh_klo;
    } else {
      // large populations: a 256-ary search of the precomputed prefix
      uint64_t lo = 0, hi = A.ntiles - 1;
      while (hi - lo >= (uint64_t)kBlock) {  // (workgroup-uniform)
        const uint64_t stride = (hi - lo + kBlock) / kBlock;
        uint64_t kk = lo + (uint64_t)tid * stride + (stride - 1);
        kk = kk > hi ? hi : kk;
        if (lo + (uint64_t)tid * stride <= hi && nhi_of(kk) > j0) atomicMin(&sh_klo, (uint32_t)tid);
        __syncthreads();
        const uint64_t i = sh_klo;
        __syncthreads();
        if (tid == 0) sh_klo = ~0u;
        const uint64_t nlo = lo + i * stride;
        uint64_t nhi = nlo + (stride - 1);
        hi = nhi > hi ? hi : nhi;
        lo = nlo;
        __syncthreads();
      }
      if (lo + (uint64_t)tid <= hi && nhi_of(lo + (uint64_t)tid) > j0) atomicMin(&sh_klo, (uint32_t)tid);
      __syncthreads();
      k_lo = lo + sh_klo;
    }
    GJX_DBG_STOP(A, 2);
    // ---- the window: it starts at the 64-particle block of tile k_lo that holds the first tooth of [j0, j1) ----------
    uint64_t i_base;
    {
      const uint64_t tbase = k_lo * kTile;
      const uint64_t cnt = tbase + kTile <= A.n ? (uint64_t)kTile : A.n - tbase;
      const int sb = lane & (kSubs - 1);  // (every 16 lanes evaluate the tile's 16 sub-prefixes: no exchange needed)
      const uint64_t cs = src_subs(k_lo)->sub[sb];
      const double scale_t = comb_tile_scale(scale, shift_at(k_lo)), tb = comb_base(pre_at(k_lo), scale, u0);
      const int32_t nhi = nhi_of(k_lo);
      const bool ends_tile = (uint64_t)(sb + 1) * kSubLen >= cnt;
      const int32_t ns = ends_tile ? nhi : comb_in_tile((double)cs, scale_t, tb, nhi, n_out);
      const uint64_t hit = __ballot(ns > j0) & 0xffffull;  // (non-empty: the tile's end lies beyond j0)
      const int blk0 = hit ? __builtin_ctzll(hit) : kSubs - 1;
      i_base = tbase + (uint64_t)blk0 * kSubLen;
    }
    bool covered = false;
    int rounds = 0;
    for (; rounds < A.scan_max && !covered; ++rounds) {
      // ---- one round: the lane's 8 consecutive sources, their running sums from the block's sub-prefix and an 8-lane
      // scan (integer adds: no exponential, no workgroup-wide scan), teeth below each, marks where runs start -----------
      const uint64_t i0 = i_base + (uint64_t)rounds * kWindow + (uint64_t)kSrc * tid;
      const uint64_t k = i0 >> 10;
      const bool live = i0 < A.n && tid < kScanLanes;
      uint32_t q[kSrc];
      if (live && i0 + kSrc <= A.n) {
        const uint4* qv4 = reinterpret_cast<const uint4*>(src_qw(i0));  // (8 consecutive sources: one tile, one owner)
        const uint4 v0 = qv4[0];
        const uint4 v1 = qv4[1];
        q[0] = v0.x; q[1] = v0.y; q[2] = v0.z; q[3] = v0.w; q[4] = v1.x; q[5] = v1.y; q[6] = v1.z; q[7] = v1.w;
      } else {
#pragma unroll
        for (int r = 0; r < kSrc; ++r) q[r] = live && i0 + r < A.n ? *src_qw(i0 + r) : 0u;
      }
      const uint32_t sbk = (uint32_t)(i0 >> 6) & (kSubs - 1);
      const uint64_t subpre = live && sbk ? src_subs(k)->sub[sbk - 1] : 0;
      uint64_t own = 0;
#pragma unroll
      for (int r = 0; r < kSrc; ++r) own += q[r];
      const uint64_t c_start = subpre + scan8_incl(own) - own;
      int32_t n_end = n_out;  // a lane beyond the population ends the comb
      if (live) {
        const double scale_t = comb_tile_scale(scale, shift_at(k)), tb = comb_base(pre_at(k), scale, u0);
        const int32_t nhi = nhi_of(k);
        const uint64_t tend = (k + 1) * kTile < A.n ? (k + 1) * kTile : A.n;  // the tile's (real) end
        double c = (double)c_start;
        int32_t start = comb_in_tile(c, scale_t, tb, nhi, n_out);  // (the tile's start for its first source: c == 0)
        n_end = i0 + kSrc >= tend ? nhi : comb_in_tile((double)(c_start + own), scale_t, tb, nhi, n_out);
        if (n_end > start && n_end > j0 && start < j1) {  // the lane's sources own a tooth of this tile's slots
          const uint32_t id_base = (uint32_t)(i0 - i_base) + 1u;
#pragma unroll
          for (int r = 0; r < kSrc; ++r) {
            c += (double)q[r];
            const int32_t nr = r == kSrc - 1 ? n_end : (i0 + r + 1 >= tend ? nhi : comb_in_tile(c, scale_t, tb, nhi, n_out));
            if (nr > start && nr > j0 && start < j1) marks[(start > j0 ? start : (int32_t)j0) - (int32_t)j0] = id_base + (uint32_t)r;
            start = nr;
          }
        }
      }
      // the window reaches the end of the workgroup's slots?  (two flags by round parity: a wave may be one barrier ahead)
      if (tid == kScanLanes - 1) sh_cov[rounds & 1] = n_end >= j1 ? 1u : 0u;
      __syncthreads();
      covered = sh_cov[rounds & 1] != 0;
    }
    GJX_DBG_STOP(A, 3);
    if (covered) {
      // ---- a max-scan spreads the marks over the runs ----------------------------------------------------------------
      uint32_t v[kPer];
      uint32_t run_max = 0;
#pragma unroll
      for (int r = 0; r < kPer; ++r) {
        const uint32_t x = marks[kPer * tid + r];
        run_max = x > run_max ? x : run_max;
        v[r] = run_max;
      }
      const uint32_t carry = block_scan_umax_excl(run_max, sh_u, true);  // (sh_u is used here only, once per launch)
#pragma unroll
      for (int r = 0; r < kPer; ++r) {
        const uint32_t a = v[r] > carry ? v[r] : carry;
        const uint64_t g = i_base + (uint64_t)(a ? a - 1u : 0u);
        anc[r] = (uint32_t)(g < A.n ? g : A.n - 1);
      }
    } else {
      // ---- many light sources (the slots' sources spread over more than scan_max windows): every slot searches the
      // tile prefix, then its tile's 16 sub-prefixes, then walks at most 64 stored weights ------------------------------
#pragma unroll 1
      for (int r = 0; r < kPer; ++r) {
        const int64_t j = jq + r;
        uint64_t k, pk;   // the slot's source tile, the mass before it
        int dk;           // ... its shift
        int32_t nhi;      // ... teeth below its end
        if (grouped && k_end < A.ntiles && (int64_t)nhi_of(k_end - 1) <= j) {
          // (grouped route, rarer still: the slot's tooth lies beyond the 1024 tiles whose prefix is in LDS — the first
          // group whose end has more than j teeth below it, by binary search of the group prefix; then a walk over that
          // group's tile records: at most kGroupTiles loads, the same shifted masses)
          uint64_t gl = k_end / kGroupTiles, gh = ngroups - 1;
          while (gl < gh) {
            const uint64_t mid = (gl + gh) >> 1;
            const int32_t hi_g = mid + 1 >= ngroups ? n_out : comb_tile(sh_gpre[mid + 1], scale, u0, n_out);
            if ((int64_t)hi_g > j) gh = mid;
            else gl = mid + 1;
          }
          uint64_t t = gl * kGroupTiles;
          const uint64_t t_end = t + kGroupTiles < A.ntiles ? t + kGroupTiles : A.ntiles;
          uint64_t run = sh_gpre[gl];
          k = t_end - 1; pk = run; dk = 64; nhi = n_out;
          bool hit = false;
          for (; t < t_end; ++t) {
            const int d = tile_shift(e, recs[t].e);
            const uint64_t m = shr64(recs[t].s, d);
            const int32_t hi_t = t + 1 >= A.ntiles ? n_out : comb_tile(run + m, scale, u0, n_out);
            if (!hit && ((int64_t)hi_t > j || t + 1 == t_end)) { k = t; pk = run; dk = d; nhi = hi_t; hit = true; }
            run += m;
          }
        } else {
          uint64_t tl = k_lo, th = (lds_prefix ? k_end : A.ntiles) - 1;
          while (tl < th) {
            const uint64_t mid = (tl + th) >> 1;
            if ((int64_t)nhi_of(mid) > j) th = mid;
            else tl = mid + 1;
          }
          k = tl; pk = pre_at(k); dk = shift_at(k); nhi = nhi_of(k);
        }
        const double scale_t = comb_tile_scale(scale, dk), tb = comb_base(pk, scale, u0);
        const uint64_t tbase = k * kTile;
        const uint64_t cnt = tbase + kTile <= A.n ? (uint64_t)kTile : A.n - tbase;  // real particles of the tile
        // block: the first 64-particle block whose END has more than j teeth below it (the tile's last block at the latest)
        uint32_t blk = kSubs - 1;
        bool have = false;
        uint64_t cbase = 0;  // running sum before the block
        for (int sb = 0; sb < kSubs; ++sb) {
          const uint64_t cs = src_subs(k)->sub[sb];
          const bool ends_tile = (uint64_t)(sb + 1) * kSubLen >= cnt;
          const int32_t ns = ends_tile ? nhi : comb_in_tile((double)cs, scale_t, tb, nhi, n_out);
          if (!have && (int64_t)ns > j) { blk = (uint32_t)sb; have = true; }
          if (!have) cbase = cs;
        }
        // walk the block: the first particle with more than j teeth below it
        double c = (double)cbase;
        uint32_t found = kSubLen - 1;
        bool done = false;
        for (uint32_t i = 0; i < (uint32_t)kSubLen; ++i) {
          const uint64_t li_ = (uint64_t)blk * kSubLen + i;
          const uint32_t qv = li_ < cnt ? *src_qw(tbase + li_) : 0u;
          c += (double)qv;
          const bool last = li_ + 1 >= cnt;
          const int32_t ni = last ? nhi : comb_in_tile(c, scale_t, tb, nhi, n_out);
          if (!done && (int64_t)ni > j) { found = i; done = true; }
        }
        const uint64_t g = tbase + (uint64_t)blk * kSubLen + found;
        anc[r] = (uint32_t)(g < A.n ? g : A.n - 1);
      }
      __syncthreads();
    }
  }

  GJX_DBG_STOP(A, 4, if (anc[0] == 0xffffffffu) marks[0] = anc[1] + anc[2] + anc[3]);
  typename Policy::Out out[kPer];
  float w[kPer];
  policy_compute_quad(P, jq, anc, out, w, 0);
  if (ADAPTIVE && !resample) {
#pragma unroll
    for (int r = 0; r < kPer; ++r) w[r] = w[r] + lw_prev[r];
  }
  policy_store_quad(P, jq, A.out_lo, anc, out, ok, 0);
  GJX_DBG_STOP(A, 5);
  if (Policy::kEmit) {
    const uint64_t loc = (uint64_t)(jq - A.out_lo);
    emit_tile<ADAPTIVE>(w, ok, qw_out + loc, logw_out ? logw_out + loc : nullptr, recs_out + ot, subs_out + ot,
                        adaptive && ess_out ? ess_out + ot : nullptr, A.wt_stores != 0);
  }
}
// The step's scalar arguments, fetched in ONE round.  Left to itself the compiler loads a kernel argument right before its
// first use, behind whatever branch decides that it is needed: the one-filter step went through five dependent rounds of
// scalar loads (~0.2 us each) before its first record load was issued, and another one in front of every later phase.
// Naming the values as inputs of an empty asm statement at the kernel's entry makes them live there, so their loads are issued
// together (no instruction is emitted; the values stay in scalar registers).
GJX_DEV void resample_args_anchor(const ResampleArgs& A) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" ::"s"(A.qw), "s"(A.recs), "s"(A.subs), "s"(A.n), "s"(A.ntiles), "s"(A.n_out), "s"(A.out_lo), "s"(A.out_hi));
  asm volatile("" ::"s"(A.u0), "s"(A.e_out), "s"(A.q_out), "s"(A.prefix), "s"(A.groups), "s"(A.fb.n_filters), "s"(A.ess_thr));
  asm volatile("" ::"s"(A.qw_out), "s"(A.logw_out), "s"(A.recs_out), "s"(A.subs_out), "s"(A.scan_max), "s"(A.xcd_map),
               "s"(A.wt_stores), "s"(A.wave_route), "s"(A.resampled_out));
#endif
}
template <int IMPL, class Policy, bool ADAPTIVE = true, bool PEERS = false>
GJX_DEV void resample_body(const ResampleArgs& A, Policy& P) {
#ifndef GJX_NO_ARGS_ANCHOR   // (A/B builds: profiles/r04_ab/README.md; anchoring the policy's members as well measured even)
  resample_args_anchor(A);
#endif
  if (ADAPTIVE) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::"s"(A.lw), "s"(A.ess), "s"(A.ess_out));
#endif
  }
  if (A.prefix == nullptr) resample_body_impl<IMPL, Policy, ADAPTIVE, PEERS, true>(A, P);  // (launch-uniform)
  else resample_body_impl<IMPL, Policy, ADAPTIVE, PEERS, false>(A, P);
}

}  // namespace gjx
