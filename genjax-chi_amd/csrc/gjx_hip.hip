// gjx_hip.hip — libgjx_hip.so: hand-written gfx950 kernels behind the C-ABI of include/gjx.h.
//
// Layout in HBM: a trace is a flat struct-of-arrays buffer — one 4-byte column per `@` site with
// the particle axis contiguous, plus score[n] and logw[n].  Workgroups are 256 threads (one wave
// per SIMD) and own a tile of 1024 consecutive particles, so every global access is a full
// 256-B-per-wave coalesced row.  Reductions are staged wave -> LDS -> per-tile partials; weight
// sums are exact u64 fixed-point, so any reduction order (and any number of GPUs) gives the same
// bits.  No entry point allocates, frees or synchronises (graph-capturable), except plan
// create/destroy.
#include "../../include/gjx.h"
#include "gjx_device.hpp"

#include <hip/hip_runtime.h>
#include <math.h>
#include <new>
#include <string.h>

#include "gjx_plan_jit.hpp"

using namespace gjx;

namespace {

inline int launch_status() { return hipGetLastError() == hipSuccess ? GJX_OK : GJX_ERR_LAUNCH; }
inline hipStream_t S(gjx_stream s) { return reinterpret_cast<hipStream_t>(s); }
inline uint64_t ntiles_of(uint64_t n) { return (n + kTile - 1) / kTile; }
inline uint64_t nrows_of(uint64_t n) { return (n + kBlock - 1) / kBlock; }  // max-partial granularity
inline unsigned grid_for(uint64_t n) {
  uint64_t b = ntiles_of(n);
  return (unsigned)(b < 1 ? 1 : (b > 0x7fffffffull ? 0x7fffffffull : b));
}

bool keys_ok(const gjx_keys* k) {
  if (!k) return false;
  if (k->impl != 0 && k->impl != 1) return false;
  if (k->mode == 0) return k->keys != nullptr;
  if (k->impl == 0 && k->parent_lane != 0) return false;  // lanes are a PHILOX notion
  return k->mode == 1 || k->mode == 2;
}
KeySrc key_src(const gjx_keys* k) {
  KeySrc s;
  s.keys = k->keys;
  s.parent = Key{k->parent[0], k->parent[1], (uint32_t)k->parent_lane, (uint32_t)(k->parent_lane >> 32)};
  s.first = k->first;
  s.mode = k->mode;
  s.has_fold = k->has_fold;
  s.fold = k->fold;
  return s;
}

struct Opnd {
  const float* p;
  float s;
  GJX_DEV float at(uint64_t i) const { return p ? p[i] : s; }
};
inline Opnd opnd(gjx_f32 a) { return Opnd{a.ptr, a.scalar}; }

// Every elementwise kernel walks its tile as 4 rows of 256 lanes: i = tile*1024 + r*256 + tid.
#define GJX_TILE_LOOP(i, n)                                                              \
  for (uint64_t tile_ = blockIdx.x; tile_ * kTile < (n); tile_ += gridDim.x)             \
    for (uint64_t i = tile_ * kTile + threadIdx.x; i < (n) && i < (tile_ + 1) * kTile;  \
         i += kBlock)

// ------------------------------------------------------------------------------------------------
// RNG kernels
// ------------------------------------------------------------------------------------------------
template <int IMPL>
__global__ __launch_bounds__(kBlock) void k_rng_keys(KeySrc ks, uint64_t n, uint32_t* out) {
  GJX_TILE_LOOP(i, n) {
    Key k = key_at<IMPL>(ks, i);
    if (ks.has_fold) k = fold_in<IMPL>(k, ks.fold);
    store_key<IMPL>(out, i, k);
  }
}
template <int IMPL>
__global__ __launch_bounds__(kBlock) void k_rng_split_each(KeySrc ks, uint64_t n, uint32_t m,
                                                           uint32_t* out) {
  const uint64_t total = n * m;
  GJX_TILE_LOOP(e, total) {
    const uint64_t i = e / m, j = e % m;
    Key k = key_at<IMPL>(ks, i);
    if (ks.has_fold) k = fold_in<IMPL>(k, ks.fold);
    k = split_at<IMPL>(k, j);
    store_key<IMPL>(out, e, k);
  }
}
template <int IMPL>
__global__ __launch_bounds__(kBlock) void k_rng_bits(KeySrc ks, uint32_t sub, uint64_t n,
                                                     uint32_t* out) {
  GJX_TILE_LOOP(i, n) {
    const Stream<IMPL> st(key_at<IMPL>(ks, i), ks.has_fold != 0, ks.fold);
    out[i] = st.bits32(sub);
  }
}

// ------------------------------------------------------------------------------------------------
// Categorical rows (small K, evaluated on the fly: sequential in k exactly as the spec states)
// ------------------------------------------------------------------------------------------------
GJX_DEV float row_max(const float* l, uint32_t K) {
  float m = l[0];
  for (uint32_t c = 1; c < K; ++c) m = l[c] > m ? l[c] : m;
  return m;
}
GJX_DEV float row_lse(const float* l, uint32_t K) {
  const float m = row_max(l, K);
  float acc = 0.0f;
  for (uint32_t c = 0; c < K; ++c) acc = acc + m_exp(l[c] - m);
  return m + m_log(acc);
}
GJX_DEV int32_t cat_invcdf(const float* l, uint32_t K, uint32_t bits) {
  const float m = row_max(l, K);
  uint64_t Q = 0;
  for (uint32_t c = 0; c < K; ++c) Q += cat_fix(l[c], m);
  const uint64_t thr = ((uint64_t)bits * Q) >> 32;
  uint64_t C = 0;
  for (uint32_t c = 0; c < K; ++c) {
    C += cat_fix(l[c], m);
    if (C > thr) return (int32_t)c;
  }
  return (int32_t)(K - 1);
}
template <int IMPL>
GJX_DEV int32_t cat_gumbel(const float* l, uint32_t K, const Stream<IMPL>& st) {
  int32_t best = 0;
  float bv = -__builtin_inff();
  for (uint32_t c = 0; c < K; ++c) {
    const float v = l[c] + gumbel_from_bits(st.bits32(c));
    if (v > bv || c == 0) {
      bv = v;
      best = (int32_t)c;
    }
  }
  return best;
}
GJX_DEV const float* cat_row(const float* logits, uint64_t n_rows, uint32_t K,
                             const int32_t* row_index, uint64_t i) {
  const uint64_t r = row_index ? (uint64_t)row_index[i] : (n_rows == 1 ? 0 : i);
  return logits + r * K;
}

// ------------------------------------------------------------------------------------------------
// Elementwise fused sample + log-density
// ------------------------------------------------------------------------------------------------
template <int IMPL>
__global__ __launch_bounds__(kBlock) void k_sample_normal(KeySrc ks, Opnd loc, Opnd scale,
                                                          float* val, float* score, uint64_t n) {
  GJX_TILE_LOOP(i, n) {
    const Stream<IMPL> st(key_at<IMPL>(ks, i), ks.has_fold != 0, ks.fold);
    const float mu = loc.at(i), sg = scale.at(i);
    const float eps = site_normal<IMPL>(st);
    const float t = sg * eps;
    const float v = mu + t;
    val[i] = v;
    if (score) score[i] = logpdf_normal(v, mu, sg);
  }
}
template <int IMPL>
__global__ __launch_bounds__(kBlock) void k_sample_gamma(KeySrc ks, Opnd conc, Opnd rate,
                                                         float* val, float* score, uint64_t n) {
  GJX_TILE_LOOP(i, n) {
    const Stream<IMPL> st(key_at<IMPL>(ks, i), ks.has_fold != 0, ks.fold);
    const float a = conc.at(i), b = rate.at(i);
    const float v = std_gamma<IMPL>(st, 0, a) / b;
    val[i] = v;
    if (score) score[i] = logpdf_gamma(v, a, b);
  }
}
template <int IMPL>
__global__ __launch_bounds__(kBlock) void k_sample_beta(KeySrc ks, Opnd a_, Opnd b_, float* val,
                                                        float* score, uint64_t n) {
  GJX_TILE_LOOP(i, n) {
    const Stream<IMPL> st(key_at<IMPL>(ks, i), ks.has_fold != 0, ks.fold);
    const float a = a_.at(i), b = b_.at(i);
    const float g1 = std_gamma<IMPL>(st, 0, a);
    const float g2 = std_gamma<IMPL>(st, 1, b);
    const float v = g1 / (g1 + g2);
    val[i] = v;
    if (score) score[i] = logpdf_beta(v, a, b);
  }
}
template <int IMPL>
__global__ __launch_bounds__(kBlock) void k_sample_bernoulli(KeySrc ks, Opnd probs, uint8_t* val,
                                                             float* score, uint64_t n) {
  GJX_TILE_LOOP(i, n) {
    const Stream<IMPL> st(key_at<IMPL>(ks, i), ks.has_fold != 0, ks.fold);
    const float p = probs.at(i);
    const bool e = uniform01(st.bits32(0)) < p;
    val[i] = e ? 1 : 0;
    if (score) score[i] = logpdf_bernoulli(e, p);
  }
}
template <int IMPL>
__global__ __launch_bounds__(kBlock) void k_sample_categorical(KeySrc ks, const float* logits,
                                                               uint64_t n_rows, uint32_t K,
                                                               const int32_t* row_index, int mode,
                                                               int32_t* val, float* score,
                                                               uint64_t n) {
  GJX_TILE_LOOP(i, n) {
    const Stream<IMPL> st(key_at<IMPL>(ks, i), ks.has_fold != 0, ks.fold);
    const float* l = cat_row(logits, n_rows, K, row_index, i);
    const int32_t v = mode == 0 ? cat_gumbel<IMPL>(l, K, st) : cat_invcdf(l, K, st.bits32(0));
    val[i] = v;
    if (score) score[i] = l[v] - row_lse(l, K);
  }
}

__global__ __launch_bounds__(kBlock) void k_logpdf_normal(Opnd v, Opnd loc, Opnd scale,
                                                          float* score, uint64_t n) {
  GJX_TILE_LOOP(i, n) score[i] = logpdf_normal(v.at(i), loc.at(i), scale.at(i));
}
__global__ __launch_bounds__(kBlock) void k_logpdf_gamma(Opnd v, Opnd conc, Opnd rate,
                                                         float* score, uint64_t n) {
  GJX_TILE_LOOP(i, n) score[i] = logpdf_gamma(v.at(i), conc.at(i), rate.at(i));
}
__global__ __launch_bounds__(kBlock) void k_logpdf_beta(Opnd v, Opnd a, Opnd b, float* score,
                                                        uint64_t n) {
  GJX_TILE_LOOP(i, n) score[i] = logpdf_beta(v.at(i), a.at(i), b.at(i));
}
__global__ __launch_bounds__(kBlock) void k_logpdf_bernoulli(const uint8_t* v, int vs, Opnd probs,
                                                             float* score, uint64_t n) {
  GJX_TILE_LOOP(i, n) score[i] = logpdf_bernoulli(v ? v[i] != 0 : vs != 0, probs.at(i));
}
__global__ __launch_bounds__(kBlock) void k_logpdf_categorical(const int32_t* v, int vs,
                                                               const float* logits,
                                                               uint64_t n_rows, uint32_t K,
                                                               const int32_t* row_index,
                                                               float* score, uint64_t n) {
  GJX_TILE_LOOP(i, n) {
    const float* l = cat_row(logits, n_rows, K, row_index, i);
    const int32_t c = v ? v[i] : vs;
    score[i] = (c < 0 || (uint32_t)c >= K) ? -__builtin_inff() : l[c] - row_lse(l, K);
  }
}

// ------------------------------------------------------------------------------------------------
// Fused static-model importance: the whole `@gen` body per particle in one kernel.
// The site table lives in device memory and is read through wave-uniform (scalar) loads; site
// values that later sites may reference are kept in LDS as vals[site][lane] (conflict-free).
// ------------------------------------------------------------------------------------------------
struct CArg {
  int32_t kind, ref;   // ref: LDS slot (SITE / TABLE) or input column (INPUT)
  int32_t ref_is_int;  // the referenced site value is an int32 (bernoulli / categorical)
  int32_t ref_site;    // SITE / TABLE: index of the referenced site (plan specialisation)
  float scale, offset;
  const float* table;
};
struct CSite {
  int32_t dist, observed, out_col, n_cat, n_rows, cat_mode;
  int32_t slot;  // LDS slot this site's value is kept in for later sites, -1 if never referenced
  CArg a0, a1, obs;
  const float* logits;
  // categorical sites of specialised kernels: per-row tables built once on the device (cat_tables_prepare) — the row's
  // inclusive fixed-point CDF (inverse-CDF sampling by a guided walk instead of two passes over the row) packed with the
  // normalised log-probabilities (log-density = row[v] - lse(row), the same f32 subtraction done once per entry); null =
  // evaluate the row on the fly.  Same integers / same f32 ops: the same bits.
  const uint2* cat_ent;       // [n_rows, n_cat]: {inclusive CDF, bits of row[c] - lse(row)} — the draw's walk ends on the entry
                              // that also holds the drawn category's log-density
  const uint4* cat_guide4;    // [n_rows, 1 << cat_gbits]: bucket g = the draw's top cat_gbits bits -> r04: {the LAST draw that maps to
                              // the bucket's first category c, c | next category with mass << 9 | (more than two categories
                              // share the bucket) << 18, log-density bits of c, log-density bits of the next}: ONE 16-byte
                              // load answers a draw unless more than two categories share its bucket (k_cat_prepare).  r03's
                              // 256 buckets held {c, CDF_c, log-density, total} and the walk read 1.4 further entries per draw
                              // of the 256-state HMM: 2.4 scattered L2 requests per draw, the bound of that scan
  int32_t cat_gbits;          // log2 of the buckets per row
  const float* cat_logp_t;    // observed sites whose value is the same for every particle: [n_cat, n_rows], row[c] - lse(row)
                              // TRANSPOSED — the launch reads one contiguous n_rows-float column, whatever rows the particles hold
  int32_t pre;       // 1: pre0/pre1 hold the hoisted per-site constants; 2: they depend on launch parameters
                     // (GJX_ARG_PARAM) and are re-derived by gjx_plan_set_params into PlanParams::d[2 site], [2 site + 1]
  float pre0, pre1;  // normal: rs, lognorm; gamma: -, lognorm; beta: -, lbeta
};
static_assert(GJX_MAX_SITES == 64, "RunCols::out is sized for 64 sites");

constexpr int kPPT = 4;             // particles per thread: ILP across independent particles
constexpr int kImpTile = kPPT * kBlock;  // the interpreter kernel walks 1024-particle tiles
constexpr int kMaskNormal = 1 << GJX_DIST_NORMAL;
constexpr int kMaskReal = kMaskNormal | (1 << GJX_DIST_GAMMA) | (1 << GJX_DIST_BETA) | (1 << GJX_DIST_BERNOULLI);
constexpr int kMaskAll = kMaskReal | (1 << GJX_DIST_CATEGORICAL);

// Site-outer / particle-inner interpreter: the site table is decoded once per tile row (scalar
// loads, wave-uniform branches) and each decoded site is applied to the thread's 4 particles, so
// four independent cipher / erfinv chains are in flight per lane.  Values a later site refers to
// live in LDS slots assigned by liveness at plan creation (vals[slot][r][lane], conflict-free).
// MASK is the compile-time set of distributions the plan may contain: an all-Normal model does not
// carry the gamma rejection loop or the categorical scans in its instruction stream.
template <int IMPL, int MASK>
__global__ __launch_bounds__(kBlock) void k_importance(const CSite* __restrict__ sites,
                                                       int n_sites, KeySrc ks, RunCols cols,
                                                       float* score, float* logw, uint64_t n,
                                                       float* max_partials, int32_t* row_e,
                                                       uint64_t* row_s, PlanParams prm) {
  extern __shared__ uint32_t vals[];  // [n_slots][kPPT][kBlock]
  __shared__ float sh_red[kBlock / kWave];
  __shared__ uint64_t sh_sum[kBlock / kWave];
  const int tid = threadIdx.x;

  for (uint64_t tile = blockIdx.x; tile * kImpTile < n; tile += gridDim.x) {
    uint64_t idx[kPPT];
    Key pkey[kPPT];
    float w[kPPT], sc[kPPT];
    uint32_t draws = 0;  // PHILOX: sampled sites so far (their fold)
#pragma unroll
    for (int r = 0; r < kPPT; ++r) {
      idx[r] = tile * kImpTile + (uint64_t)r * kBlock + tid;
      // out-of-range lanes of the last tile run on the last particle and are masked at the stores
      pkey[r] = key_at<IMPL>(ks, idx[r] < n ? idx[r] : n - 1);
      w[r] = 0.0f;
      sc[r] = 0.0f;
    }
    for (int q = 0; q < n_sites; ++q) {
      const CSite& st = sites[q];
      const int dist = st.dist;
      const bool is_int = dist >= GJX_DIST_BERNOULLI;
      auto eval = [&](const CArg& a, int r) -> float {
        switch (a.kind) {
          case GJX_ARG_CONST: return a.offset;
          case GJX_ARG_SITE: {
            const uint32_t raw = vals[(a.ref * kPPT + r) * kBlock + tid];
            const float base = a.ref_is_int ? (float)(int32_t)raw : u2f(raw);
            const float t = a.scale * base;
            return t + a.offset;
          }
          case GJX_ARG_INPUT: {
            const float t = a.scale * cols.in[a.ref][idx[r] < n ? idx[r] : n - 1];
            return t + a.offset;
          }
          case GJX_ARG_PARAM: {
            const float t = a.scale * prm.p[a.ref];
            return t + a.offset;
          }
          default: {
            const uint32_t raw = vals[(a.ref * kPPT + r) * kBlock + tid];
            const int32_t ti = a.ref_is_int ? (int32_t)raw : (int32_t)__builtin_rintf(u2f(raw));
            return a.table[ti];
          }
        }
      };
      float a0[kPPT], a1[kPPT];
      const float* row[kPPT];
#pragma unroll
      for (int r = 0; r < kPPT; ++r) {
        a0[r] = 0.0f;
        a1[r] = 0.0f;
        row[r] = nullptr;
        if ((MASK & (1 << GJX_DIST_CATEGORICAL)) && dist == GJX_DIST_CATEGORICAL) {
          int32_t rr;
          if (st.a0.kind == GJX_ARG_SITE) {
            const uint32_t raw = vals[(st.a0.ref * kPPT + r) * kBlock + tid];
            rr = st.a0.ref_is_int ? (int32_t)raw : (int32_t)__builtin_rintf(u2f(raw));
          } else if (st.a0.kind == GJX_ARG_CONST) {
            rr = (int32_t)__builtin_rintf(st.a0.offset);
          } else {
            rr = (int32_t)__builtin_rintf(eval(st.a0, r));
          }
          rr = rr < 0 ? 0 : (rr >= st.n_rows ? st.n_rows - 1 : rr);
          row[r] = st.logits + (size_t)rr * (size_t)st.n_cat;
        } else {
          a0[r] = eval(st.a0, r);
          if (dist != GJX_DIST_BERNOULLI) a1[r] = eval(st.a1, r);
        }
      }
      float vf[kPPT];
      int32_t vi[kPPT];
      if (st.observed) {
#pragma unroll
        for (int r = 0; r < kPPT; ++r) {
          const float ov = st.obs.kind == GJX_ARG_CONST ? st.obs.offset
                           : st.obs.kind == GJX_ARG_PARAM ? eval(st.obs, r)
                                                          : cols.in[st.obs.ref][idx[r] < n ? idx[r] : n - 1];
          vf[r] = ov;
          vi[r] = is_int ? (int32_t)__builtin_rintf(ov) : 0;
        }
      } else {
        const uint32_t fold = IMPL == 0 ? (uint32_t)(q + 1) : draws++;
        // single-word draws (normal / bernoulli / inverse-CDF categorical)
        uint32_t bits[kPPT];
        // (PHILOX Normal sites pair particles, site_normal: their words are derived there)
        const bool one_word = (dist == GJX_DIST_NORMAL && IMPL == 0) || dist == GJX_DIST_BERNOULLI ||
                              (dist == GJX_DIST_CATEGORICAL && st.cat_mode == 1);
        if (one_word) {
#pragma unroll
          for (int r = 0; r < kPPT; ++r) bits[r] = Stream<IMPL>(pkey[r], true, fold).bits32(0);
        }
#pragma unroll
        for (int r = 0; r < kPPT; ++r) {
          vf[r] = 0.0f;
          vi[r] = 0;
          if (dist == GJX_DIST_NORMAL) {
            const float eps = IMPL == 0 ? std_normal(bits[r]) : site_normal<IMPL>(Stream<IMPL>(pkey[r], true, fold));
            const float t = a1[r] * eps;
            vf[r] = a0[r] + t;
          } else if ((MASK & (1 << GJX_DIST_BERNOULLI)) && dist == GJX_DIST_BERNOULLI) {
            vi[r] = uniform01(bits[r]) < a0[r] ? 1 : 0;
          } else if ((MASK & (1 << GJX_DIST_GAMMA)) && dist == GJX_DIST_GAMMA) {
            const Stream<IMPL> strm(pkey[r], true, fold);
            vf[r] = std_gamma<IMPL>(strm, 0, a0[r]) / a1[r];
          } else if ((MASK & (1 << GJX_DIST_BETA)) && dist == GJX_DIST_BETA) {
            const Stream<IMPL> strm(pkey[r], true, fold);
            const float g1 = std_gamma<IMPL>(strm, 0, a0[r]);
            const float g2 = std_gamma<IMPL>(strm, 1, a1[r]);
            vf[r] = g1 / (g1 + g2);
          } else if ((MASK & (1 << GJX_DIST_CATEGORICAL)) && dist == GJX_DIST_CATEGORICAL) {
            if (st.cat_mode == 0) {
              const Stream<IMPL> strm(pkey[r], true, fold);
              vi[r] = cat_gumbel<IMPL>(row[r], (uint32_t)st.n_cat, strm);
            } else {
              vi[r] = cat_invcdf(row[r], (uint32_t)st.n_cat, bits[r]);
            }
          }
        }
      }
#pragma unroll
      for (int r = 0; r < kPPT; ++r) {
        float lp = 0.0f;
        const float pre0 = st.pre == 2 ? prm.d[2 * q] : st.pre0, pre1 = st.pre == 2 ? prm.d[2 * q + 1] : st.pre1;
        if (dist == GJX_DIST_NORMAL) {
          lp = st.pre ? logpdf_normal_pre(vf[r], a0[r], pre0, pre1) : logpdf_normal(vf[r], a0[r], a1[r]);
        } else if ((MASK & (1 << GJX_DIST_GAMMA)) && dist == GJX_DIST_GAMMA) {
          lp = st.pre ? logpdf_gamma_pre(vf[r], a0[r], a1[r], pre1) : logpdf_gamma(vf[r], a0[r], a1[r]);
        } else if ((MASK & (1 << GJX_DIST_BETA)) && dist == GJX_DIST_BETA) {
          lp = st.pre ? logpdf_beta_pre(vf[r], a0[r], a1[r], pre1) : logpdf_beta(vf[r], a0[r], a1[r]);
        } else if ((MASK & (1 << GJX_DIST_BERNOULLI)) && dist == GJX_DIST_BERNOULLI) {
          lp = logpdf_bernoulli(vi[r] != 0, a0[r]);
        } else if ((MASK & (1 << GJX_DIST_CATEGORICAL)) && dist == GJX_DIST_CATEGORICAL) {
          lp = (vi[r] < 0 || vi[r] >= st.n_cat) ? -__builtin_inff()
                                               : row[r][vi[r]] - row_lse(row[r], (uint32_t)st.n_cat);
        }
        sc[r] = sc[r] + lp;
        if (st.observed) w[r] = w[r] + lp;
        const uint32_t raw = is_int ? (uint32_t)vi[r] : f2u(vf[r]);
        if (st.slot >= 0) vals[(st.slot * kPPT + r) * kBlock + tid] = raw;
        if (st.out_col >= 0 && idx[r] < n) reinterpret_cast<uint32_t*>(cols.out[st.out_col])[idx[r]] = raw;
      }
    }
#pragma unroll
    for (int r = 0; r < kPPT; ++r) {
      if (idx[r] < n) {
        if (logw) logw[idx[r]] = w[r];
        if (score) score[idx[r]] = sc[r];
      }
      if ((max_partials || row_e) && (tile * kImpTile + (uint64_t)r * kBlock) < n) {  // per 256-particle row
        const float bm = block_max(idx[r] < n ? w[r] : -__builtin_inff(), sh_red);
        if (max_partials && tid == 0) max_partials[tile * kPPT + r] = bm;
        if (row_e) {  // row-anchored partial sum: no global maximum needed
          const int32_t eb = row_anchor(bm);
          const uint64_t sb = block_sum(idx[r] < n ? rowfix(w[r], eb) : 0, sh_sum);
          if (tid == 0) {
            row_e[tile * kPPT + r] = eb;
            row_s[tile * kPPT + r] = sb;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// log-sum-exp: tile partials -> tiny finishing kernels
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_max_partials(const float* x, uint64_t n,
                                                         float* partials) {
  __shared__ float sh[kBlock / kWave];
  for (uint64_t tile = blockIdx.x; tile * kTile < n; tile += gridDim.x) {
    float m = -__builtin_inff();
    for (uint64_t i = tile * kTile + threadIdx.x; i < n && i < (tile + 1) * kTile; i += kBlock) {
      const float v = x[i];
      m = v > m ? v : m;
    }
    m = block_max(m, sh);
    if (threadIdx.x == 0) partials[tile] = m;
  }
}
// one block: out[0] = max(partials[0..np))
__global__ __launch_bounds__(kBlock) void k_reduce_max(const float* partials, uint64_t np,
                                                       float* out) {
  __shared__ float sh[kBlock / kWave];
  float m = -__builtin_inff();
  for (uint64_t i = threadIdx.x; i < np; i += kBlock) {
    const float v = partials[i];
    m = v > m ? v : m;
  }
  m = block_max(m, sh);
  if (threadIdx.x == 0) out[0] = m;
}
// m_ptr == nullptr: every block reduces the row maxima itself (L2-resident, 4 B per 256 particles)
// instead of waiting for a separate one-block reduction kernel; block 0 publishes the max.
__global__ __launch_bounds__(kBlock) void k_expsum_partials(const float* x, uint64_t n,
                                                            const float* m_ptr,
                                                            const float* max_partials, uint64_t n_mp,
                                                            float* max_out, int frac,
                                                            uint64_t* partials) {
  __shared__ uint64_t sh[kBlock / kWave];
  __shared__ float shf[kBlock / kWave];
  float m;
  if (m_ptr) {
    m = m_ptr[0];
  } else {
    m = -__builtin_inff();
    for (uint64_t k = threadIdx.x; k < n_mp; k += kBlock) {
      const float v = max_partials[k];
      m = v > m ? v : m;
    }
    m = block_max(m, shf);
    if (max_out && blockIdx.x == 0 && threadIdx.x == 0) max_out[0] = m;
  }
  for (uint64_t tile = blockIdx.x; tile * kTile < n; tile += gridDim.x) {
    uint64_t acc = 0;
    for (uint64_t i = tile * kTile + threadIdx.x; i < n && i < (tile + 1) * kTile; i += kBlock)
      acc += fixw(x[i], m, frac);
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) partials[tile] = acc;
  }
}
// one block: out_q = sum(partials) (+= if accumulate); optional lse = m + log(q 2^-frac)
__global__ __launch_bounds__(kBlock) void k_reduce_sum(const uint64_t* partials, uint64_t np,
                                                       uint64_t* out_q, int accumulate,
                                                       const float* m_ptr, int frac,
                                                       float* out_lse, float* out_max) {
  __shared__ uint64_t sh[kBlock / kWave];
  uint64_t acc = 0;
  for (uint64_t i = threadIdx.x; i < np; i += kBlock) acc += partials[i];
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) {
    if (out_q) out_q[0] = accumulate ? out_q[0] + acc : acc;
    if (out_lse) {
      const float qf = (float)acc * u2f((uint32_t)(127 - frac) << 23);
      out_lse[0] = m_ptr[0] + m_log(qf);
    }
    if (out_max) out_max[0] = m_ptr[0];
  }
}
// Row-anchored partial sums of arbitrary log-weights: one pass, one workgroup per 256-particle row.
__global__ __launch_bounds__(kBlock) void k_row_stats(const float* x, uint64_t n, int32_t* row_e,
                                                      uint64_t* row_s) {
  __shared__ float shf[kBlock / kWave];
  __shared__ uint64_t sh64[kBlock / kWave];
  for (uint64_t row = blockIdx.x; row * kBlock < n; row += gridDim.x) {
    const uint64_t i = row * kBlock + threadIdx.x;
    const float v = i < n ? x[i] : -__builtin_inff();
    const int32_t eb = row_anchor(block_max(v, shf));
    const uint64_t sb = block_sum(i < n ? rowfix(v, eb) : 0, sh64);
    if (threadIdx.x == 0) {
      row_e[row] = eb;
      row_s[row] = sb;
    }
  }
}
// gjx_lse_rows / gjx_lse_rows_batch: one workgroup per pass folds that pass's row pairs
// (lse_rows_block, gjx_device.hpp); passes lie batch_stride rows apart.
__global__ __launch_bounds__(kBlock) void k_lse_rows(const int32_t* row_e, const uint64_t* row_s,
                                                     uint64_t n_rows, uint64_t batch_stride, int32_t* out_e,
                                                     uint64_t* out_q, float* out_lse,
                                                     uint64_t* out_record) {
  const uint64_t b = blockIdx.x;
  lse_rows_block<false>(row_e + b * batch_stride, row_s + b * batch_stride, n_rows, out_e ? out_e + b : nullptr,
                        out_q ? out_q + b : nullptr, out_lse ? out_lse + b : nullptr,
                        out_record ? out_record + b * kLseRecordWords : nullptr);
}
// Merge `n_records` records (one per rank; consecutive records `record_stride` words apart) for each of
// gridDim.x independent passes (`batch_stride` words apart): buckets re-indexed to the common anchor.
__global__ __launch_bounds__(kWave) void k_lse_combine(const uint64_t* records, int n_records,
                                                       uint64_t record_stride, uint64_t batch_stride,
                                                       int32_t* out_e, uint64_t* out_q, float* out_lse,
                                                       uint64_t* out_record) {
  const uint64_t* base = records + (uint64_t)blockIdx.x * batch_stride;
  int32_t e = kRowEmpty;
  for (int r = 0; r < n_records; ++r) {
    const int32_t er = (int32_t)(int64_t)base[(uint64_t)r * record_stride];
    e = er > e ? er : e;
  }
  uint64_t bucket = 0;
  const int d = threadIdx.x;
  for (int r = 0; r < n_records; ++r) {
    const uint64_t* rec = base + (uint64_t)r * record_stride;
    const int32_t er = (int32_t)(int64_t)rec[0];
    if (er == kRowEmpty) continue;
    const int64_t k = (int64_t)d - ((int64_t)e - (int64_t)er);
    if (k >= 0) bucket += rec[1 + k];
  }
  lse_emit(e, bucket, out_e ? out_e + blockIdx.x : nullptr, out_q ? out_q + blockIdx.x : nullptr,
           out_lse ? out_lse + blockIdx.x : nullptr,
           out_record ? out_record + (uint64_t)blockIdx.x * kLseRecordWords : nullptr);
}
__global__ void k_lse_finish(const float* m_ptr, const uint64_t* q_ptr, int frac, float* out) {
  const float qf = (float)q_ptr[0] * u2f((uint32_t)(127 - frac) << 23);
  out[0] = m_ptr[0] + m_log(qf);
}

// A policy turns (output slot j, its ancestor's GLOBAL index) into the new particle.  `compute` is pure so the
// kernel can run four slots' cipher / transform chains interleaved; `store` writes the results.
struct AncestorOnly {
  static constexpr bool kEmit = false;  // no weights: nothing to emit for a next resampling
  int32_t* anc;  // [out_hi - out_lo]
  struct Out {};
  GJX_DEV void select_filter(uint64_t off, Key) { anc += off; }
  GJX_DEV float compute(int64_t, uint32_t, Out&) const { return 0.0f; }
  GJX_DEV void store(int64_t j, int64_t out_lo, uint32_t src, const Out&) const { anc[j - out_lo] = (int32_t)src; }
};

// ONE launch per SMC step: resample (from the previous step's records and in-tile CDFs) + gather + propagate + weight
// + this step's CDFs and records (resample_body, gjx_device.hpp).
template <class Policy>
constexpr auto policy_peers(int) -> decltype(Policy::kPeers) { return Policy::kPeers; }
template <class Policy>
constexpr bool policy_peers(long) { return false; }
template <class Policy>
GJX_DEV auto policy_bind(Policy& P, const StepParams* sp, const float* rp, int) -> decltype(P.bind(sp, rp), void()) { P.bind(sp, rp); }
template <class Policy>
GJX_DEV void policy_bind(Policy&, const StepParams*, const float*, long) {}
template <int IMPL, class Policy, bool ADAPTIVE = false>
__global__ __launch_bounds__(kBlock) void k_resample(ResampleArgs A, Policy P) {
  if (A.sp) {  // (a replayed run: the step's key / observation / parameters from device memory)
    policy_bind(P, A.sp, A.rp, 0);
    // the NEXT step's entry is pulled into this XCD's L2 now (every workgroup: all eight L2s): the next launch reads its key
    // from cache instead of from memory in front of its first cipher round (the block has one entry to spare)
    if (threadIdx.x == 0) (void)*reinterpret_cast<const volatile uint32_t*>(A.sp + 1);
  }
  resample_body<IMPL, Policy, ADAPTIVE, policy_peers<Policy>(0)>(A, P);
}

// ---- r04: the two small launches of the peer transport (gjx.h: gjx_smc_peer_signal / gjx_smc_peer_wait) ------------------
// Workgroup o serves rank o: it copies this rank's tile records (and ESS sums) into rank o's arena, every copying thread
// makes its stores visible at system scope, and then one thread raises this rank's arrival word in rank o's flags.  The
// launch runs after the step's launch on the same stream, so what the step wrote to this rank's own arena (weights, state,
// sub-prefixes) is complete in memory when a peer sees the word.
GJX_DEV void peer_deposit_and_raise(const PeerMap& pm, int o, int32_t rank, const TileRec* recs, const TileEss* ess, uint64_t first_tile,
                                    uint64_t n_tiles, uint64_t value) {
  if (o != rank && recs) {
    TileRec* dst = const_cast<TileRec*>(peer_ptr(recs, pm.delta[o]));
    for (uint64_t k = threadIdx.x; k < n_tiles; k += kBlock) {
      const uint4 v = *reinterpret_cast<const uint4*>(recs + first_tile + k);
      *reinterpret_cast<uint4*>(dst + first_tile + k) = v;
    }
    if (ess) {
      TileEss* de = const_cast<TileEss*>(peer_ptr(ess, pm.delta[o]));
      for (uint64_t k = threadIdx.x; k < n_tiles; k += kBlock) {
        const uint4 v = *reinterpret_cast<const uint4*>(ess + first_tile + k);
        *reinterpret_cast<uint4*>(de + first_tile + k) = v;
      }
    }
  }
  __threadfence_system();  // release: this thread's copies, and everything earlier launches left in this XCD's L2
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t* word = const_cast<uint64_t*>(peer_ptr(pm.flags, pm.delta[o])) + rank;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the flag must not overtake the write-back: MI355X guide, compiler hazard)
    __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ __launch_bounds__(kBlock) void k_peer_signal(PeerMap pm, int32_t rank, const TileRec* recs, const TileEss* ess,
                                                        uint64_t first_tile, uint64_t n_tiles, uint64_t value) {
  peer_deposit_and_raise(pm, (int)blockIdx.x, rank, recs, ess, first_tile, n_tiles, value);
}
__global__ __launch_bounds__(kWave) void k_peer_wait(PeerMap pm) { (void)peer_wait_wave(pm); }

// Per-tile fixed-point mass of log-weights under their GLOBAL maximum (the multinomial / single-draw
// paths, which materialise a global CDF).  The max is reduced redundantly by every block from the per-tile maxima.
__global__ __launch_bounds__(kBlock) void k_tile_sums_block(const float* lw, uint64_t n_local, const float* max_partials,
                                                            uint64_t n_mp, int frac, uint64_t* tile_sums_at, float* max_out) {
  __shared__ uint64_t sh64[kBlock / kWave];
  __shared__ float shf[kBlock / kWave];
  const uint64_t tile = blockIdx.x;
  float lwv[kPer];
#pragma unroll
  for (int r = 0; r < kPer; ++r) {  // issued before the max reduction so the latencies overlap
    const uint64_t i = tile * kTile + (uint64_t)r * kBlock + threadIdx.x;
    lwv[r] = i < n_local ? lw[i] : -__builtin_inff();
  }
  float m = -__builtin_inff();
  for (uint64_t k = threadIdx.x; k < n_mp; k += kBlock) {
    const float v = max_partials[k];
    m = v > m ? v : m;
  }
  m = block_max(m, shf);
  if (max_out && tile == 0 && threadIdx.x == 0) max_out[0] = m;
  uint64_t acc = 0;
#pragma unroll
  for (int r = 0; r < kPer; ++r) {
    const uint64_t i = tile * kTile + (uint64_t)r * kBlock + threadIdx.x;
    if (i < n_local) acc += fixw(lwv[r], m, frac);
  }
  acc = block_sum(acc, sh64);
  if (threadIdx.x == 0) tile_sums_at[tile] = acc;
}

// Tile-anchored fixed-point weights and records of ARBITRARY log-weights (the generic resampler's front half; the SMC
// kernels emit theirs themselves): one workgroup per tile.
__global__ __launch_bounds__(kBlock) void k_tile_weights(const float* lw, uint64_t n, uint32_t* qw, TileRec* recs, TileSub* subs,
                                                         TileEss* ess) {
  const uint64_t base = (uint64_t)blockIdx.x * kTile + (uint64_t)kPer * threadIdx.x;
  float w[kPer];
  bool ok[kPer];
#pragma unroll
  for (int r = 0; r < kPer; ++r) {
    ok[r] = base + r < n;
    w[r] = ok[r] ? lw[base + r] : -__builtin_inff();
  }
  if (ess) emit_tile<true>(w, ok, qw + base, nullptr, recs + blockIdx.x, subs + blockIdx.x, ess + blockIdx.x);
  else emit_tile<false>(w, ok, qw + base, nullptr, recs + blockIdx.x, subs + blockIdx.x, nullptr);
}

// The merge of a population's tile records by ONE workgroup per filter: anchor e, total mass Q, ESS sums, and — for
// populations beyond kMaxLdsTiles, where every resample workgroup merging all records itself would dominate — the
// exclusive prefix of the shifted tile masses (layout: gjx_device.hpp prefix_words).  Also the closing (e, Q) of a run.
__global__ __launch_bounds__(kBlock) void k_scan_records(const TileRec* recs, const TileEss* ess, uint64_t ntiles,
                                                         uint64_t* prefix, int32_t* e_out, uint64_t* q_out, uint64_t mq_stride) {
  __shared__ uint64_t sh64[kBlock / kWave];
  __shared__ uint64_t sh_e[2 * (kBlock / kWave)];
  __shared__ float shf[kBlock / kWave];
  recs += (uint64_t)blockIdx.x * ntiles;  // one workgroup per filter
  if (ess) ess += (uint64_t)blockIdx.x * ntiles;
  if (prefix) prefix += (uint64_t)blockIdx.x * prefix_words(ntiles);
  const uint64_t per = (ntiles + kBlock - 1) / kBlock;
  const uint64_t lo = threadIdx.x * per < ntiles ? threadIdx.x * per : ntiles, hi = lo + per < ntiles ? lo + per : ntiles;
  if (per <= 4) {
    // r04: up to 1024 tiles (every batch of filters: this launch stands between two step launches) — the thread's records
    // are loaded ONCE, all loads in flight together; the three passes below then run on registers.  Same integers.
    uint64_t sv[4], a1[4], a2[4];
    int32_t ev[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint64_t k = lo + (uint64_t)i;
      const bool in = k < hi;
      const uint4 raw = in ? *reinterpret_cast<const uint4*>(recs + k) : make_uint4(0u, 0u, (uint32_t)kRowEmpty, 0u);
      sv[i] = ((uint64_t)raw.y << 32) | raw.x;
      ev[i] = (int32_t)raw.z;
      const uint4 er = in && ess ? *reinterpret_cast<const uint4*>(ess + k) : make_uint4(0u, 0u, 0u, 0u);
      a1[i] = ((uint64_t)er.y << 32) | er.x;
      a2[i] = ((uint64_t)er.w << 32) | er.z;
    }
    float ef4 = (float)kRowEmpty;
#pragma unroll
    for (int i = 0; i < 4; ++i) ef4 = (float)ev[i] > ef4 ? (float)ev[i] : ef4;
    const int32_t e4 = (int32_t)block_max(ef4, shf);
    uint64_t m[4], local4 = 0, l14 = 0, l24 = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = tile_shift(e4, ev[i]);
      m[i] = shr64(sv[i], d);
      local4 += m[i];
      l14 += shr64(a1[i], d);
      l24 += shr64(a2[i], 2 * d);
    }
    uint64_t total4;
    uint64_t run4 = block_scan_excl(local4, sh64, total4);
    if (ess) {
      l14 = block_sum(l14, sh_e);
      l24 = block_sum(l24, sh_e + kBlock / kWave);
    }
    if (prefix) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (lo + (uint64_t)i < hi) prefix[lo + i] = run4;
        run4 += m[i];
      }
    }
    if (threadIdx.x == 0) {
      if (prefix) {
        prefix[ntiles] = total4;
        prefix[ntiles + 1] = (uint64_t)(int64_t)e4;
        prefix[ntiles + 2] = l14;
        prefix[ntiles + 3] = l24;
      }
      if (e_out) e_out[(uint64_t)blockIdx.x * mq_stride] = e4;
      if (q_out) q_out[(uint64_t)blockIdx.x * mq_stride] = total4;
    }
    return;
  }
  // anchors are integers of magnitude <= 2^24 (row_anchor): exact as floats, so the float block max serves
  float ef = (float)kRowEmpty;
  for (uint64_t k = lo; k < hi; ++k) {
    const float v = (float)recs[k].e;
    ef = v > ef ? v : ef;
  }
  const int32_t e = (int32_t)block_max(ef, shf);
  uint64_t local = 0, l1 = 0, l2 = 0;
  for (uint64_t k = lo; k < hi; ++k) {
    const int d = tile_shift(e, recs[k].e);
    local += shr64(recs[k].s, d);
    if (ess) { l1 += shr64(ess[k].r1, d); l2 += shr64(ess[k].r2, 2 * d); }
  }
  uint64_t total;
  uint64_t run = block_scan_excl(local, sh64, total);
  if (ess) {
    l1 = block_sum(l1, sh_e);
    l2 = block_sum(l2, sh_e + kBlock / kWave);
  }
  if (prefix) {
    for (uint64_t k = lo; k < hi; ++k) {
      prefix[k] = run;
      run += shr64(recs[k].s, tile_shift(e, recs[k].e));
    }
  }
  if (threadIdx.x == 0) {
    if (prefix) {
      prefix[ntiles] = total;
      prefix[ntiles + 1] = (uint64_t)(int64_t)e;
      prefix[ntiles + 2] = l1;
      prefix[ntiles + 3] = l2;
    }
    if (e_out) e_out[(uint64_t)blockIdx.x * mq_stride] = e;
    if (q_out) q_out[(uint64_t)blockIdx.x * mq_stride] = total;
  }
}

// r04: the same merge for LARGE populations (beyond kMaxLdsTiles: the precomputed-prefix route of every step of such a filter,
// e.g. one rank of BASELINE configs[3], 7 816 records).  k_scan_records walks a strided chunk per thread three times, one
// dependent 16-byte load at a time: 29 us at 7 816 records — more than two step kernels.  Here 1024 threads hold a CONTIGUOUS
// chunk of up to kBigPer records each in registers (every load issued before the first use: one memory latency), the
// anchor, the scan and the stores work on registers.  Same integers, same layout.  `pm` (peer transport): the records in
// this rank's arena are complete only once every peer has arrived, so wave 0 first waits (bounded) — the wait launch and the
// merge launch of a peer step are ONE launch.
constexpr int kBigBlock = 1024;
constexpr int kBigPer = 16;  // -> up to 16 384 tiles (16.7M particles) per filter; beyond that k_scan_records serves
template <bool ESS>
__global__ __launch_bounds__(kBigBlock) void k_scan_records_big(const TileRec* recs, const TileEss* ess, uint64_t ntiles, uint64_t* prefix,
                                                                PeerMap pm) {
  constexpr int kW = kBigBlock / kWave;
  __shared__ uint64_t sh_s[3 * kW];
  __shared__ int32_t sh_e[kW];
  __shared__ uint32_t sh_ok;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  if (pm.world > 0) {
    if (wv == 0) {
      const bool ready = peer_wait_wave(pm);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) sh_ok = ready ? 1u : 0u;
    }
    __syncthreads();
    if (!sh_ok) return;
  }
  const uint64_t per = (ntiles + kBigBlock - 1) / kBigBlock;  // <= kBigPer (checked by the host)
  const uint64_t lo = (uint64_t)tid * per;
  uint64_t sv[kBigPer], r1v[ESS ? kBigPer : 1], r2v[ESS ? kBigPer : 1];
  int32_t ev[kBigPer];
#pragma unroll
  for (int i = 0; i < kBigPer; ++i) {
    const uint64_t k = lo + i;
    const bool in = (uint64_t)i < per && k < ntiles;
    if (in) {
      const uint4 raw = *reinterpret_cast<const uint4*>(recs + k);
      sv[i] = ((uint64_t)raw.y << 32) | raw.x;
      ev[i] = (int32_t)raw.z;
    } else {
      sv[i] = 0;
      ev[i] = kRowEmpty;
    }
    if (ESS) {
      r1v[i] = in ? ess[k].r1 : 0;
      r2v[i] = in ? ess[k].r2 : 0;
    }
  }
  int32_t e = kRowEmpty;
#pragma unroll
  for (int i = 0; i < kBigPer; ++i) e = ev[i] > e ? ev[i] : e;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int32_t o = __shfl_xor(e, off, kWave);
    e = o > e ? o : e;
  }
  if (lane == 0) sh_e[wv] = e;
  __syncthreads();
  e = sh_e[0];
#pragma unroll
  for (int i = 1; i < kW; ++i) e = sh_e[i] > e ? sh_e[i] : e;
  uint64_t local = 0, l1 = 0, l2 = 0;
#pragma unroll
  for (int i = 0; i < kBigPer; ++i) {
    const int d = tile_shift(e, ev[i]);
    sv[i] = shr64(sv[i], d);  // the tile's shifted mass
    local += sv[i];
    if (ESS) {
      l1 += shr64(r1v[i], d);
      l2 += shr64(r2v[i], 2 * d);
    }
  }
  const uint64_t incl = wave_scan_incl(local);
  if (ESS) {
    l1 = wave_sum(l1);
    l2 = wave_sum(l2);
  }
  if (lane == 63) { sh_s[wv] = incl; sh_s[kW + wv] = l1; sh_s[2 * kW + wv] = l2; }
  __syncthreads();
  uint64_t base = 0, total = 0, t1 = 0, t2 = 0;
#pragma unroll
  for (int i = 0; i < kW; ++i) {
    if (i < wv) base += sh_s[i];
    total += sh_s[i];
    t1 += sh_s[kW + i];
    t2 += sh_s[2 * kW + i];
  }
  uint64_t run = base + incl - local;
#pragma unroll
  for (int i = 0; i < kBigPer; ++i) {
    const uint64_t k = lo + i;
    if ((uint64_t)i < per && k < ntiles) prefix[k] = run;
    run += sv[i];
  }
  if (tid == 0) {
    prefix[ntiles] = total;
    prefix[ntiles + 1] = (uint64_t)(int64_t)e;
    prefix[ntiles + 2] = t1;
    prefix[ntiles + 3] = t2;
  }
}

// r04: the GROUP records of a large population (gjx_device.hpp GroupRec; the grouped route of resample_body).  One workgroup
// per group of kGroupTiles = 256 tiles, thread t <-> tile t of the group: the group's anchor, then — lane L of every wave —
// the sum over the wave's 64 tiles of their masses shifted by (e_G - e_t) + L (a readlane loop: no cross-lane reduction), the
// four waves' partial tables added through LDS.  Parallel over the groups (31 workgroups for one rank of configs[3]) where
// k_scan_records_big is one workgroup's serial scan; `pm` (peer transport): every workgroup first waits, bounded, for the peers.
template <bool ESS>
__global__ __launch_bounds__(kBlock) void k_group_records(const TileRec* recs, const TileEss* ess, uint64_t ntiles, GroupRec* groups,
                                                          PeerMap pm) {
  static_assert(kGroupTiles == kBlock, "one thread per tile of the group");
  constexpr int kW = kBlock / kWave;
  __shared__ uint64_t sh_t[3][kW][64];
  __shared__ int32_t sh_e[kW];
  __shared__ uint32_t sh_ok;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  if (pm.world > 0) {
    // a deferred signal of the previous step (gjx_smc_peers.signal_*): workgroup o serves rank o before anybody waits — the
    // signal launch of step t and this waiting launch of step t + 1 are one launch
    if (pm.sig_value != 0 && (int)blockIdx.x < pm.world)
      peer_deposit_and_raise(pm, (int)blockIdx.x, pm.rank, pm.sig_recs, pm.sig_ess, pm.sig_first, pm.sig_n, pm.sig_value);
    if ((uint64_t)blockIdx.x * kGroupTiles >= ntiles) return;  // (workgroups launched for the signal alone)
    if (wv == 0) {
      const bool ready = peer_wait_wave(pm);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) sh_ok = ready ? 1u : 0u;
    }
    __syncthreads();
    if (!sh_ok) return;
  }
  const uint64_t g = blockIdx.x, k = g * kGroupTiles + (uint64_t)tid;
  const bool in = k < ntiles;
  uint64_t S = 0, R1 = 0, R2 = 0;
  int32_t et = kRowEmpty;
  if (in) {
    const uint4 raw = *reinterpret_cast<const uint4*>(recs + k);
    S = ((uint64_t)raw.y << 32) | raw.x;
    et = (int32_t)raw.z;
    if (ESS) { R1 = ess[k].r1; R2 = ess[k].r2; }
  }
  int32_t e = et;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int32_t o = __shfl_xor(e, off, kWave);
    e = o > e ? o : e;
  }
  if (lane == 0) sh_e[wv] = e;
  __syncthreads();
  e = sh_e[0];
#pragma unroll
  for (int i = 1; i < kW; ++i) e = sh_e[i] > e ? sh_e[i] : e;
  const int d = tile_shift(e, et);  // 64: the tile carries nothing under this anchor
  uint64_t tm = 0, t1 = 0, t2 = 0;
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const uint64_t Si = __shfl(S, i, kWave);
    const int di = __shfl(d, i, kWave);
    tm += shr64(Si, di + lane);
    if (ESS) {
      t1 += shr64(__shfl(R1, i, kWave), di + lane);
      t2 += shr64(__shfl(R2, i, kWave), 2 * (di + lane));
    }
  }
  sh_t[0][wv][lane] = tm;
  if (ESS) { sh_t[1][wv][lane] = t1; sh_t[2][wv][lane] = t2; }
  __syncthreads();
  if (tid < 64) {
    uint64_t a = 0, b1 = 0, b2 = 0;
#pragma unroll
    for (int i = 0; i < kW; ++i) {
      a += sh_t[0][i][tid];
      if (ESS) { b1 += sh_t[1][i][tid]; b2 += sh_t[2][i][tid]; }
    }
    groups[g].mass[tid] = a;
    groups[g].r1[tid] = b1;
    groups[g].r2[tid] = b2;
    if (tid == 0) groups[g].e = e;
  }
}
// Which route a population beyond kMaxLdsTiles takes (resample_body): group records (default) or the precomputed global
// prefix (GJX_SMC_BIG_ROUTE=prefix: the r03 route, kept for A/B and for populations beyond kMaxGroups groups).  `scratch`:
// the caller's u64[prefix_words(ntiles)].  -> true: A.groups is set and the group launch is enqueued.
static bool launch_group_records(ResampleArgs& A, uint64_t* scratch, hipStream_t st) {
  static const bool allow = [] { const char* e = std::getenv("GJX_SMC_BIG_ROUTE"); return !(e && e[0] == 'p'); }();
  const uint64_t ng = (A.ntiles + kGroupTiles - 1) / kGroupTiles;
  if (!allow || !scratch || ng > (uint64_t)kMaxGroups || (((uintptr_t)scratch) & 15) != 0 ||
      ng * sizeof(GroupRec) > prefix_words(A.ntiles) * sizeof(uint64_t))
    return false;
  GroupRec* groups = reinterpret_cast<GroupRec*>(scratch);
  const unsigned grid = (unsigned)(A.pm.world > 0 && A.pm.sig_value != 0 && (uint64_t)A.pm.world > ng ? (uint64_t)A.pm.world : ng);
  if (A.ess) k_group_records<true><<<grid, kBlock, 0, st>>>(A.recs, A.ess, A.ntiles, groups, A.pm);
  else k_group_records<false><<<grid, kBlock, 0, st>>>(A.recs, A.ess, A.ntiles, groups, A.pm);
  A.groups = groups;
  A.prefix = nullptr;
  return true;
}

// Inclusive fixed-point CDF materialised in HBM (multinomial / single-draw paths).
__global__ __launch_bounds__(kBlock) void k_cdf(const float* lw, uint64_t n, const float* m_ptr,
                                                const uint64_t* tile_sums, uint64_t ntiles,
                                                int frac, uint64_t* cdf) {
  __shared__ uint64_t sh64[kBlock / kWave];
  const uint64_t b = blockIdx.x;
  uint64_t pre = 0;
  for (uint64_t k = threadIdx.x; k < b; k += kBlock) pre += tile_sums[k];
  pre = block_sum(pre, sh64);
  const float m = m_ptr[0];
  const uint64_t base = b * kTile;
  uint64_t q[kPer], local = 0;
#pragma unroll
  for (int r = 0; r < kPer; ++r) {
    const uint64_t i = base + kPer * (uint64_t)threadIdx.x + r;
    q[r] = i < n ? fixw(lw[i], m, frac) : 0;
    local += q[r];
  }
  uint64_t tt;
  uint64_t run = pre + block_scan_excl(local, sh64, tt);
#pragma unroll
  for (int r = 0; r < kPer; ++r) {
    const uint64_t i = base + kPer * (uint64_t)threadIdx.x + r;
    run += q[r];
    if (i < n) cdf[i] = run;
  }
}
GJX_DEV uint64_t cdf_upper_bound(const uint64_t* cdf, uint64_t n, uint64_t thr) {
  uint64_t lo = 0, hi = n - 1;  // first i with cdf[i] > thr
  while (lo < hi) {
    const uint64_t mid = (lo + hi) >> 1;
    if (cdf[mid] > thr) hi = mid;
    else lo = mid + 1;
  }
  return lo;
}
template <int IMPL>
__global__ __launch_bounds__(kBlock) void k_multinomial(Key rkey, int has_fold, uint32_t fold,
                                                        const uint64_t* cdf, uint64_t n,
                                                        uint64_t n_out, int32_t* anc,
                                                        int64_t* idx64) {
  const Stream<IMPL> st(rkey, has_fold != 0, fold);
  const uint64_t Q = cdf[n - 1];
  GJX_TILE_LOOP(j, n_out) {
    const uint64_t thr = __umul64hi(st.bits64((uint32_t)j), Q);
    const uint64_t a = cdf_upper_bound(cdf, n, thr);
    if (anc) anc[j] = (int32_t)a;
    if (idx64) idx64[j] = (int64_t)a;
  }
}

// Gumbel-max single draw over n logits (jax.random.categorical semantics): tile argmax partials.
// One 1024-logit tile of a Gumbel-max draw: the workgroup's argmax of logits[i] + gumbel(bits32(i)), first index attaining the
// maximum (sequential-scan semantics).  Result in shv[0] / shi[0]; every thread calls it (barriers inside).
template <int IMPL>
GJX_DEV void gumbel_tile_argmax(const Stream<IMPL>& st, const float* logits, uint64_t n, uint64_t tile, float* shv, int64_t* shi) {
  float bv = -__builtin_inff();
  int64_t bi = INT64_MAX;
  for (uint64_t i = tile * kTile + threadIdx.x; i < n && i < (tile + 1) * kTile; i += kBlock) {
    const float v = logits[i] + gumbel_from_bits(st.bits32((uint32_t)i));
    if (v > bv || bi == INT64_MAX) { bv = v; bi = (int64_t)i; }
  }
  shv[threadIdx.x] = bv;
  shi[threadIdx.x] = bi;
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      const float ov = shv[threadIdx.x + off];
      const int64_t oi = shi[threadIdx.x + off];
      const float mv = shv[threadIdx.x];
      const int64_t mi = shi[threadIdx.x];
      // sequential-scan semantics: first index attaining the maximum wins
      if (oi != INT64_MAX && (mi == INT64_MAX || ov > mv || (ov == mv && oi < mi))) {
        shv[threadIdx.x] = ov;
        shi[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
}
template <int IMPL>
__global__ __launch_bounds__(kBlock) void k_gumbel_partials(Key key, int has_fold, uint32_t fold,
                                                            const float* logits, uint64_t n,
                                                            float* pv, int64_t* pi) {
  __shared__ float shv[kBlock];
  __shared__ int64_t shi[kBlock];
  const Stream<IMPL> st(key, has_fold != 0, fold);
  for (uint64_t tile = blockIdx.x; tile * kTile < n; tile += gridDim.x) {
    gumbel_tile_argmax<IMPL>(st, logits, n, tile, shv, shi);
    if (threadIdx.x == 0) { pv[tile] = shv[0]; pi[tile] = shi[0]; }
    __syncthreads();
  }
}
// r04: B independent draws in ONE launch (gjx_categorical_index_batch: `vmap(alg.random_weighted)` over keys draws one
// particle per trial): workgroup b draws from logits[b * stride .. + n) under key b; n <= one tile, so the workgroup's
// result is the single call's (k_gumbel_partials of the one tile, then k_argmax_final of one partial) bit for bit.
constexpr int kMaxDrawBatch = 64;
struct DrawKeys {
  Key key[kMaxDrawBatch];
  uint32_t fold[kMaxDrawBatch];
  uint32_t has_fold[kMaxDrawBatch];
};
template <int IMPL>
__global__ __launch_bounds__(kBlock) void k_gumbel_batch(DrawKeys dk, const float* logits, uint64_t n, uint64_t stride, int64_t* out) {
  __shared__ float shv[kBlock];
  __shared__ int64_t shi[kBlock];
  const uint32_t b = blockIdx.x;
  const Stream<IMPL> st(dk.key[b], dk.has_fold[b] != 0, dk.fold[b]);
  gumbel_tile_argmax<IMPL>(st, logits + (uint64_t)b * stride, n, 0, shv, shi);
  if (threadIdx.x == 0) out[b] = shi[0];
}

__global__ __launch_bounds__(kBlock) void k_argmax_final(const float* pv, const int64_t* pi,
                                                         uint64_t np, int64_t* out) {
  __shared__ float shv[kBlock];
  __shared__ int64_t shi[kBlock];
  float bv = -__builtin_inff();
  int64_t bi = INT64_MAX;
  for (uint64_t k = threadIdx.x; k < np; k += kBlock) {
    const float v = pv[k];
    const int64_t i = pi[k];
    if (bi == INT64_MAX || v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
  }
  shv[threadIdx.x] = bv;
  shi[threadIdx.x] = bi;
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      const float ov = shv[threadIdx.x + off];
      const int64_t oi = shi[threadIdx.x + off];
      const float mv = shv[threadIdx.x];
      const int64_t mi = shi[threadIdx.x];
      if (oi != INT64_MAX && (mi == INT64_MAX || ov > mv || (ov == mv && oi < mi))) {
        shv[threadIdx.x] = ov;
        shi[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = shi[0];
}

struct GatherCols {
  const uint32_t* src[16];
  uint32_t* dst[16];
};
__global__ __launch_bounds__(kBlock) void k_gather(const int32_t* anc, uint64_t n_out,
                                                   GatherCols g, int n_cols) {
  GJX_TILE_LOOP(j, n_out) {
    const int32_t a = anc[j];
    for (int c = 0; c < n_cols; ++c) g.dst[c][j] = g.src[c][a];
  }
}

// ------------------------------------------------------------------------------------------------
// Fused bootstrap-SMC policies: propagate + weight the four consecutive output slots of a lane.
// ------------------------------------------------------------------------------------------------
template <int IMPL, bool PEERS = false>
struct LgssmPolicy {
  static constexpr bool kEmit = true;
  static constexpr bool kPeers = PEERS;
  const float* prev_state;  // [n] (global) previous-step particles
  float* state_out;         // [n_local]
  int32_t* anc_out;         // nullable [n_local]
  Key step_key;
  float a, q, y, rs, lognorm;
  float z[kPer];            // the quad's standard normals (prefetch: they do not depend on the ancestors)
  int wt = 0;               // write-through stores of the state / ancestor columns (store16_out)
  const int64_t* pd = nullptr;  // PEERS: byte offsets of the peers' arenas (LDS), tiles per rank (set_peers)
  uint32_t tpr = 1;
  GJX_DEV void set_peers(const int64_t* d, uint32_t t) { pd = d; tpr = t; }
  GJX_DEV void select_filter(uint64_t off, Key k) {
    prev_state += off; state_out += off;
    if (anc_out) anc_out += off;
    step_key = k;
  }
  GJX_DEV void prefetch(int64_t jq) { smc_quad_normals<IMPL>(step_key, (uint64_t)jq >> 2, z); }
  // (rp: kLgssmRunParams floats — x0_loc, x0_scale, a, q, rs, lognorm)
  GJX_DEV void bind(const StepParams* sp, const float* rp) {
    step_key.k0 = sp->k0; step_key.k1 = sp->k1;
    y = u2f(sp->y_bits);
    if (rp) { a = rp[2]; q = rp[3]; rs = rp[4]; lognorm = rp[5]; }
  }
  GJX_DEV float source(uint32_t anc) const { return src_load<PEERS>(prev_state, anc, pd, tpr); }
  struct Out {
    float x;
  };
  // the lane's four consecutive slots jq .. jq+3 (jq a multiple of 4): one quad of normals
  GJX_DEV void compute_quad(int64_t, const uint32_t (&anc)[4], Out (&o)[4], float (&w)[4]) const {
    float xv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = source(anc[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float mean = a * xv[u];
      const float t = q * z[u];
      o[u].x = mean + t;
      w[u] = logpdf_normal_pre(y, o[u].x, rs, lognorm);
    }
  }
  GJX_DEV void store_quad(int64_t jq, int64_t out_lo, const uint32_t (&anc)[4], const Out (&o)[4], const bool (&ok)[4]) const {
    const int64_t k = jq - out_lo;
    const bool vec = ((((uintptr_t)state_out | (uintptr_t)anc_out) & 15) == 0);  // uniform
    if (vec && ok[0] && ok[1] && ok[2] && ok[3]) {
      store16_out(state_out + k, make_uint4(f2u(o[0].x), f2u(o[1].x), f2u(o[2].x), f2u(o[3].x)), wt != 0);
      if (anc_out) store16_out(anc_out + k, make_uint4(anc[0], anc[1], anc[2], anc[3]), wt != 0);
      return;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (ok[u]) {
        state_out[k + u] = o[u].x;
        if (anc_out) anc_out[k + u] = (int32_t)anc[u];
      }
    }
  }
};

// The transition table is an ALIAS table (gjx.h): row z holds K packed entries
// (threshold24 << 8) | alias, and a draw costs ONE 4-byte table load: column = floor(bits K / 2^32), the next
// 24 bits of the product choose between the column and its alias.  The HMM step is bound by the rate of
// scattered L2-resident loads (one cache line per lane per load), not by arithmetic: an inverse-CDF walk
// (row total + guide byte + CDF word: three lines per draw) ran at 0.55 of this form's speed.
GJX_DEV uint32_t hmm_alias_pick(uint32_t e, uint32_t col, uint32_t f24) { return f24 < (e >> 8) ? col : (e & 255u); }
GJX_DEV uint32_t hmm_alias_draw(const uint32_t* row, int32_t K, uint32_t bits) {
  const uint64_t t = (uint64_t)bits * (uint64_t)(uint32_t)K;
  const uint32_t col = (uint32_t)(t >> 32);
  return hmm_alias_pick(row[col], col, (uint32_t)t >> 8);
}

template <int IMPL, bool PEERS = false>
struct HmmPolicy {
  static constexpr bool kEmit = true;
  static constexpr bool kPeers = PEERS;
  const int32_t* prev_state;
  int32_t* state_out;
  int32_t* anc_out;
  Key step_key;
  const uint32_t* trans_cdf;  // alias table [K,K] (gjx.h: trans_alias)
  const float* obs_logp;      // [K,K]
  int32_t K, y;
  uint32_t col[kPer], f24[kPer];  // the quad's draws, split into column and fraction (prefetch)
  float oc;                       // obs_logp[tid, y] on its way to LDS
  float* ocol;                    // LDS: column y of the observation table (one entry per state)
  int wt = 0;                     // write-through stores of the state / ancestor columns (store16_out)
  const int64_t* pd = nullptr;    // PEERS: byte offsets of the peers' arenas (LDS), tiles per rank (set_peers)
  uint32_t tpr = 1;
  GJX_DEV void set_peers(const int64_t* d, uint32_t t) { pd = d; tpr = t; }
  GJX_DEV void select_filter(uint64_t off, Key k) {
    prev_state += off; state_out += off;
    if (anc_out) anc_out += off;
    step_key = k;
  }
  GJX_DEV void prefetch(int64_t jq) {
    uint32_t bits[4];
    smc_quad_bits<IMPL>(step_key, (uint64_t)jq >> 2, bits);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint64_t t = (uint64_t)bits[u] * (uint64_t)(uint32_t)K;
      col[u] = (uint32_t)(t >> 32);
      f24[u] = (uint32_t)t >> 8;
    }
    oc = (int)threadIdx.x < K ? obs_logp[(size_t)threadIdx.x * K + y] : 0.0f;
  }
  GJX_DEV void bind(const StepParams* sp, const float*) {
    step_key.k0 = sp->k0; step_key.k1 = sp->k1;
    y = (int32_t)sp->y_bits;
  }
  GJX_DEV void stage() {
    __shared__ float ocol_tile[256];
    ocol = ocol_tile;
    if ((int)threadIdx.x < K) ocol_tile[threadIdx.x] = oc;
  }
  GJX_DEV int32_t source(uint32_t anc) const { return src_load<PEERS>(prev_state, anc, pd, tpr); }
  struct Out {
    int32_t z;
  };
  GJX_DEV void compute_quad(int64_t, const uint32_t (&anc)[4], Out (&o)[4], float (&w)[4]) const {
    // the four states, then the four table words, are loaded together
    int32_t zs[4];
    uint32_t e[4], c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) zs[u] = source(anc[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) e[u] = trans_cdf[(size_t)zs[u] * K + col[u]];
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = hmm_alias_pick(e[u], col[u], f24[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      o[u].z = (int32_t)c[u];
      w[u] = ocol[c[u]];
    }
  }
  GJX_DEV void store_quad(int64_t jq, int64_t out_lo, const uint32_t (&anc)[4], const Out (&o)[4], const bool (&ok)[4]) const {
    const int64_t k = jq - out_lo;
    const bool vec = ((((uintptr_t)state_out | (uintptr_t)anc_out) & 15) == 0);  // uniform
    if (vec && ok[0] && ok[1] && ok[2] && ok[3]) {
      store16_out(state_out + k, make_uint4((uint32_t)o[0].z, (uint32_t)o[1].z, (uint32_t)o[2].z, (uint32_t)o[3].z), wt != 0);
      if (anc_out) store16_out(anc_out + k, make_uint4(anc[0], anc[1], anc[2], anc[3]), wt != 0);
      return;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (ok[u]) {
        state_out[k + u] = o[u].z;
        if (anc_out) anc_out[k + u] = (int32_t)anc[u];
      }
    }
  }
};

// Source-tile ranges of `world` equal blocks of output slots (gjx.h: gjx_smc_source_ranges).  One workgroup:
// every thread owns a contiguous chunk of tiles; the records are merged like in the resample kernel (anchor, shifted
// masses, chunk offsets by one block scan); tile b can own slots in [ceil(P_b) - 1, ceil(P_{b+1})) for some comb offset
// (P = prefix * N / Q in float64, the products teeth_below forms), both bounds monotone in b, so a block's range is
// [#tiles with upper <= lo, #tiles with lower < hi).
constexpr int kMaxRangeBlocks = 64;
__global__ __launch_bounds__(kBlock) void k_source_ranges(const TileRec* recs, const TileEss* ess, double ess_thr,
                                                          uint64_t ntiles, uint64_t n_total, int world, int64_t ticket,
                                                          int64_t* out) {
  __shared__ uint64_t sh64[kBlock / kWave];
  __shared__ uint64_t sh_e[2 * (kBlock / kWave)];
  __shared__ float shf[kBlock / kWave];
  __shared__ unsigned long long cnt[2 * kMaxRangeBlocks];
  const int tid = threadIdx.x;
  if (tid < 2 * kMaxRangeBlocks) cnt[tid] = 0;
  const uint64_t per = (ntiles + kBlock - 1) / kBlock;
  const uint64_t b0 = per * (uint64_t)tid < ntiles ? per * (uint64_t)tid : ntiles;
  const uint64_t b1 = b0 + per < ntiles ? b0 + per : ntiles;
  // r04: the thread's records are loaded ONCE, all loads in flight together (the loops below used to re-read every record
  // `world` + 2 times, one dependent strided load after the other: 98 us at the 7 816 records of BASELINE configs[3]);
  // chunks beyond kMassCache tiles per thread fall back to the loads
  constexpr int kMassCache = 32;
  uint64_t mcache[kMassCache];
  int32_t ecache[kMassCache];
  const bool cached = per <= (uint64_t)kMassCache;
  float ef = (float)kRowEmpty;  // (anchors are exact as floats: |e| <= 2^24)
  if (cached) {
#pragma unroll
    for (int i = 0; i < kMassCache; ++i) {
      // (unconditional loads from a clamped address, masked afterwards: a load inside `if (b < b1)` ends its basic block with
      // a wait, and 32 such loads are 32 memory round trips one after the other — 50 us of this kernel)
      const uint64_t b = b0 + i;
      const uint4 raw = *reinterpret_cast<const uint4*>(recs + (b < ntiles ? b : ntiles - 1));
      mcache[i] = b < b1 ? (((uint64_t)raw.y << 32) | raw.x) : 0;
      ecache[i] = b < b1 ? (int32_t)raw.z : kRowEmpty;
    }
#pragma unroll
    for (int i = 0; i < kMassCache; ++i) {
      const float v = (float)ecache[i];
      ef = v > ef ? v : ef;
    }
  } else {
    for (uint64_t b = b0; b < b1; ++b) {
      const float v = (float)recs[b].e;
      ef = v > ef ? v : ef;
    }
  }
  const int32_t e = (int32_t)block_max(ef, shf);
  if (cached) {
#pragma unroll
    for (int i = 0; i < kMassCache; ++i) mcache[i] = shr64(mcache[i], tile_shift(e, ecache[i]));
  }
  auto mass = [&](uint64_t b) { return shr64(recs[b].s, tile_shift(e, recs[b].e)); };
  uint64_t local = 0, l1 = 0, l2 = 0;
  if (cached) {
#pragma unroll
    for (int i = 0; i < kMassCache; ++i) local += mcache[i];
  }
  for (uint64_t b = b0; b < b1; ++b) {
    if (!cached) local += mass(b);
    if (ess_thr > 0.0) {
      const int d = tile_shift(e, recs[b].e);
      l1 += shr64(ess[b].r1, d);
      l2 += shr64(ess[b].r2, 2 * d);
    }
  }
  if (ess_thr > 0.0) {  // an adaptive filter that keeps its particles at the next step needs no exchange at all:
    l1 = block_sum(l1, sh_e);                       // every block's sources are its own tiles
    l2 = block_sum(l2, sh_e + kBlock / kWave);
    if (!ess_says_resample(l1, l2, ess_thr)) {
      const uint64_t tiles_per_block = (n_total / (uint64_t)world) / kTile;
      if (tid < 2 * world) {
        out[tid] = (int64_t)(((uint64_t)(tid >> 1) + (uint64_t)(tid & 1)) * tiles_per_block);
        __threadfence_system();
      }
      __syncthreads();
      if (tid == 0) __hip_atomic_store(out + 2 * world, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
  }
  uint64_t tot;
  const uint64_t pre0 = block_scan_excl(local, sh64, tot);  // (barriers inside also order the cnt[] clear)
  const double scale = (double)n_total / (double)tot;
  const double nd = (double)n_total;
  const uint64_t n_local = n_total / (uint64_t)world;
  // a tile's bounds do not depend on the block: computed once per tile (they were recomputed for every block: two u64 ->
  // f64 conversions, two products and two ceilings per tile and block), compared per block
  double lwv[kMassCache], upv[kMassCache];
  if (cached) {
    uint64_t pre = pre0;
#pragma unroll
    for (int i = 0; i < kMassCache; ++i) {
      const uint64_t b = b0 + i;
      double lower = __builtin_ceil((double)pre * scale);
      lower = lower < nd ? lower : nd;
      lwv[i] = lower - 1.0 > 0.0 ? lower - 1.0 : 0.0;
      pre += mcache[i];
      const double upper = __builtin_ceil((double)pre * scale);
      upv[i] = (b + 1 == ntiles || !(upper < nd)) ? nd : upper;
    }
  }
  for (int j = 0; j < world; ++j) {
    const double lo = (double)((uint64_t)j * n_local), hi = (double)((uint64_t)(j + 1) * n_local);
    unsigned long long first = 0, end = 0;
    if (cached) {
#pragma unroll
      for (int i = 0; i < kMassCache; ++i) {
        if (b0 + i < b1) {
          first += upv[i] <= lo ? 1 : 0;
          end += lwv[i] < hi ? 1 : 0;
        }
      }
    } else {
      uint64_t pre = pre0;
      for (uint64_t b = b0; b < b1; ++b) {
        double lower = __builtin_ceil((double)pre * scale);
        lower = lower < nd ? lower : nd;
        lower = lower - 1.0 > 0.0 ? lower - 1.0 : 0.0;
        pre += mass(b);
        double upper = __builtin_ceil((double)pre * scale);
        upper = (b + 1 == ntiles || !(upper < nd)) ? nd : upper;
        first += upper <= lo ? 1 : 0;
        end += lower < hi ? 1 : 0;
      }
    }
    // (one LDS atomic per wave and counter: 256 same-address atomics per counter serialise)
    const uint64_t wf = wave_sum((uint64_t)first), we = wave_sum((uint64_t)end);
    if ((tid & 63) == 0) {
      if (wf) atomicAdd(&cnt[2 * j], (unsigned long long)wf);
      if (we) atomicAdd(&cnt[2 * j + 1], (unsigned long long)we);
    }
  }
  __syncthreads();
  if (tid < 2 * world) {
    // no mass at all (every weight underflowed): the population is kept, every block's sources are its own tiles
    const uint64_t tiles_per_block = (n_total / (uint64_t)world) / kTile;
    // (system-scope stores: the words go straight to the host-visible buffer; the ticket's release below, behind the
    // workgroup barrier, orders them — one fence, not one per word)
    __hip_atomic_store(out + tid, tot == 0 ? (int64_t)(((uint64_t)(tid >> 1) + (uint64_t)(tid & 1)) * tiles_per_block) : (int64_t)cnt[tid],
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  // the ticket goes last, system scope: a host polling pinned memory may consume the ranges without waiting for
  // the stream
  if (tid == 0) __hip_atomic_store(out + 2 * world, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Step 0 (no resampling): one block per LOCAL tile of the rank.
template <int IMPL>
__global__ __launch_bounds__(kBlock) void k_lgssm_init(FilterBatch fb, Key step_key, uint64_t first_slot,
                                                       uint64_t n_local, float x0_loc,
                                                       float x0_scale, float y, float rs,
                                                       float lognorm, float* state_out, int32_t* anc_out, EmitOut em,
                                                       const StepParams* sp, const float* rp) {
  if (sp) {  // (a replayed run: RunGraphs)
    step_key.k0 = sp->k0; step_key.k1 = sp->k1;
    y = u2f(sp->y_bits);
    if (rp) { x0_loc = rp[0]; x0_scale = rp[1]; rs = rp[4]; lognorm = rp[5]; }
  }
  uint64_t ltile = blockIdx.x;
  if (fb.n_filters > 1) {  // several filters per launch: tile of filter f, its key, its outputs
    const uint32_t f = (uint32_t)(ltile / fb.tiles);
    ltile -= (uint64_t)f * fb.tiles;
    step_key = fb.step_key[f];
    state_out += (uint64_t)f * fb.stride;
    if (anc_out) anc_out += (uint64_t)f * fb.stride;
    select_filter_emit(em, fb, f);
  }
  const uint64_t loc = ltile * kTile + (uint64_t)kPer * threadIdx.x;  // the lane's four consecutive slots, local
  const uint64_t gq = first_slot + loc;
  float z[4];  // one quad of normals
  smc_quad_normals<IMPL>(step_key, gq >> 2, z);
  float w[kPer];
  bool ok[kPer];
#pragma unroll
  for (int r = 0; r < kPer; ++r) {
    ok[r] = loc + r < n_local;
    const float t = x0_scale * z[r];
    const float x = x0_loc + t;
    w[r] = logpdf_normal_pre(y, x, rs, lognorm);
    if (ok[r]) {
      state_out[loc + r] = x;
      if (anc_out) anc_out[loc + r] = (int32_t)(gq + r);
    }
  }
  emit_init_tile(w, ok, em, loc, first_slot / kTile + ltile);
}

template <int IMPL>
__global__ __launch_bounds__(kBlock) void k_hmm_init(FilterBatch fb, Key step_key, uint64_t first_slot,
                                                     uint64_t n_local, const uint32_t* trans_cdf,
                                                     const float* obs_logp, int32_t K,
                                                     int32_t init_state, int32_t y,
                                                     int32_t* state_out, int32_t* anc_out, EmitOut em, const StepParams* sp) {
  if (sp) {  // (a replayed run: RunGraphs)
    step_key.k0 = sp->k0; step_key.k1 = sp->k1;
    y = (int32_t)sp->y_bits;
  }
  uint64_t ltile = blockIdx.x;
  if (fb.n_filters > 1) {  // several filters per launch: tile of filter f, its key, its outputs
    const uint32_t f = (uint32_t)(ltile / fb.tiles);
    ltile -= (uint64_t)f * fb.tiles;
    step_key = fb.step_key[f];
    state_out += (uint64_t)f * fb.stride;
    if (anc_out) anc_out += (uint64_t)f * fb.stride;
    select_filter_emit(em, fb, f);
  }
  const uint64_t loc = ltile * kTile + (uint64_t)kPer * threadIdx.x;
  const uint64_t gq = first_slot + loc;
  const uint32_t* cdf = trans_cdf + (size_t)init_state * K;
  uint32_t qb[4];  // one quad of draws
  smc_quad_bits<IMPL>(step_key, gq >> 2, qb);
  float w[kPer];
  bool ok[kPer];
#pragma unroll
  for (int r = 0; r < kPer; ++r) {
    ok[r] = loc + r < n_local;
    const uint32_t lo = hmm_alias_draw(cdf, K, qb[r]);
    w[r] = obs_logp[(size_t)lo * K + y];
    if (ok[r]) {
      state_out[loc + r] = (int32_t)lo;
      if (anc_out) anc_out[loc + r] = (int32_t)(gq + r);
    }
  }
  emit_init_tile(w, ok, em, loc, first_slot / kTile + ltile);
}

// ------------------------------------------------------------------------------------------------
// Generated SMC filters WITHOUT the compiler: a table-walking policy (GJX_PLAN_JIT=0, or a failed compilation with
// GJX_PLAN_JIT_FALLBACK=1).  One slot at a time, the site table read from device memory, the values of earlier sites in a
// per-thread array — the same device functions in the same order as the generated policy (gjx_plan_jit.hpp GenSmc), so the
// same bits; several times slower (PHILOX: every slot derives its quad's block itself).  Programs (GJX_ARG_EXPR) are
// compiled, never interpreted: such filters have no route here.
struct InterpTable {
  const CSite* sites;  // device copy of the plan's init / step table
  int32_t n_sites, n_state;
  CArg state_args[GJX_SMC_MAX_STATE];
};
GJX_DEV float interp_site_f32(const CSite* sites, const uint32_t* vals, int q) {
  return sites[q].dist >= GJX_DIST_BERNOULLI ? (float)(int32_t)vals[q] : u2f(vals[q]);
}
GJX_DEV int32_t interp_site_i32(const CSite* sites, const uint32_t* vals, int q) {
  return sites[q].dist >= GJX_DIST_BERNOULLI ? (int32_t)vals[q] : (int32_t)__builtin_rintf(u2f(vals[q]));
}
GJX_DEV float interp_arg(const CArg& a, const CSite* sites, const uint32_t* vals, const float* st, const float* obs) {
  switch (a.kind) {
    case GJX_ARG_CONST: return a.offset;
    case GJX_ARG_SITE: return (a.scale * interp_site_f32(sites, vals, a.ref_site)) + a.offset;
    case GJX_ARG_STATE: return (a.scale * st[a.ref]) + a.offset;
    case GJX_ARG_OBS: return (a.scale * obs[a.ref]) + a.offset;
    default: return a.table[interp_site_i32(sites, vals, a.ref_site)];  // GJX_ARG_TABLE
  }
}
template <int IMPL>
GJX_DEV float interp_walk(const InterpTable& T, const float* obs, Key step_key, int64_t j, const float* st, float* st_out) {
  uint32_t vals[GJX_MAX_SITES];
  float w = 0.0f;
  const Key pkey = slot_key<IMPL>(step_key, (uint64_t)j);
  uint32_t draws = 0;
  for (int q = 0; q < T.n_sites; ++q) {
    const CSite& s = T.sites[q];
    const bool isint = s.dist >= GJX_DIST_BERNOULLI;
    float a0 = 0.0f, a1 = 0.0f;
    const float* row = nullptr;
    if (s.dist == GJX_DIST_CATEGORICAL) {
      int32_t rr = s.a0.kind == GJX_ARG_SITE ? interp_site_i32(T.sites, vals, s.a0.ref_site)
                   : (int32_t)__builtin_rintf(s.a0.kind == GJX_ARG_CONST ? s.a0.offset : interp_arg(s.a0, T.sites, vals, st, obs));
      rr = rr < 0 ? 0 : (rr >= s.n_rows ? s.n_rows - 1 : rr);
      row = s.logits + (size_t)rr * s.n_cat;
    } else {
      a0 = interp_arg(s.a0, T.sites, vals, st, obs);
      if (s.dist != GJX_DIST_BERNOULLI) a1 = interp_arg(s.a1, T.sites, vals, st, obs);
    }
    float vf = 0.0f;
    int32_t vi = 0;
    if (s.observed) {
      const float ov = s.obs.kind == GJX_ARG_CONST ? s.obs.offset : obs[s.obs.ref];
      if (isint) vi = (int32_t)__builtin_rintf(ov);
      else vf = ov;
    } else {
      const uint32_t fold = IMPL == 0 ? (uint32_t)(q + 1) : draws;  // THREEFRY: the `@` counter; PHILOX: index among the draws
      ++draws;
      uint32_t qw[4] = {0u, 0u, 0u, 0u};
      uint32_t bits = 0u;
      const bool one_word = s.dist == GJX_DIST_NORMAL || s.dist == GJX_DIST_BERNOULLI || (s.dist == GJX_DIST_CATEGORICAL && s.cat_mode == 1);
      if (one_word) {
        if (IMPL == 0) {
          bits = Stream<IMPL>(pkey, true, fold).bits32(0);
        } else {  // one block per quad of slots and draw: slot u takes word u
          const uint64_t g = (uint64_t)j >> 2;
          philox4x32(step_key.k0, step_key.k1, (uint32_t)g, (uint32_t)(g >> 32), fold, kTagQuad, qw[0], qw[1], qw[2], qw[3]);
          const uint32_t u = (uint32_t)j & 3u;
          bits = u == 0 ? qw[0] : (u == 1 ? qw[1] : (u == 2 ? qw[2] : qw[3]));
        }
      }
      switch (s.dist) {
        case GJX_DIST_NORMAL: {
          float eps;
          if (IMPL == 0) {
            eps = std_normal(bits);
          } else {  // two Box-Muller transforms over the quad's words: (w0, w1) -> slots 0, 1; (w2, w3) -> slots 2, 3
            float zc, zs;
            if ((uint32_t)j & 2u) bm_pair(qw[2], qw[3], zc, zs);
            else bm_pair(qw[0], qw[1], zc, zs);
            eps = ((uint32_t)j & 1u) ? zs : zc;
          }
          const float t = a1 * eps;
          vf = a0 + t;
          break;
        }
        case GJX_DIST_BERNOULLI: vi = uniform01(bits) < a0 ? 1 : 0; break;
        case GJX_DIST_GAMMA: {
          const Stream<IMPL> strm(pkey, true, fold);
          vf = std_gamma<IMPL>(strm, 0, a0) / a1;
          break;
        }
        case GJX_DIST_BETA: {
          const Stream<IMPL> strm(pkey, true, fold);
          const float g1 = std_gamma<IMPL>(strm, 0, a0);
          const float g2 = std_gamma<IMPL>(strm, 1, a1);
          vf = g1 / (g1 + g2);
          break;
        }
        default:
          if (s.cat_mode == 0) vi = cat_gumbel<IMPL>(row, (uint32_t)s.n_cat, Stream<IMPL>(pkey, true, fold));
          else vi = cat_invcdf(row, (uint32_t)s.n_cat, bits);
      }
    }
    float lp;
    switch (s.dist) {
      case GJX_DIST_NORMAL: lp = logpdf_normal(vf, a0, a1); break;
      case GJX_DIST_GAMMA: lp = logpdf_gamma(vf, a0, a1); break;
      case GJX_DIST_BETA: lp = logpdf_beta(vf, a0, a1); break;
      case GJX_DIST_BERNOULLI: lp = logpdf_bernoulli(vi != 0, a0); break;
      default: lp = (vi < 0 || vi >= s.n_cat) ? -__builtin_inff() : row[vi] - row_lse(row, (uint32_t)s.n_cat);
    }
    if (s.observed) w = w + lp;
    vals[q] = isint ? (uint32_t)vi : f2u(vf);
  }
  float nx[GJX_SMC_MAX_STATE];
  for (int k = 0; k < T.n_state; ++k) nx[k] = interp_arg(T.state_args[k], T.sites, vals, st, obs);
  for (int k = 0; k < T.n_state; ++k) st_out[k] = nx[k];
  return w;
}
template <int IMPL>
struct InterpPolicy {
  static constexpr bool kEmit = true;
  PlanPolicyArgs a;
  InterpTable T;
  struct Out { float s[GJX_SMC_MAX_STATE]; };
  GJX_DEV void select_filter(uint64_t off, Key k) {
    for (int c = 0; c < T.n_state; ++c) { a.prev_state[c] += off; a.state_out[c] += off; }
    if (a.anc_out) a.anc_out += off;
    a.step_key = k;
  }
  GJX_DEV float compute(int64_t j, uint32_t src_global, Out& out) const {
    float st[GJX_SMC_MAX_STATE];
    for (int k = 0; k < T.n_state; ++k) st[k] = a.prev_state[k][src_global];
    return interp_walk<IMPL>(T, a.obs, a.step_key, j, st, out.s);
  }
  GJX_DEV void store(int64_t j, int64_t out_lo, uint32_t src, const Out& out) const {
    for (int k = 0; k < T.n_state; ++k) a.state_out[k][j - out_lo] = out.s[k];
    if (a.anc_out) a.anc_out[j - out_lo] = (int32_t)src;
  }
};
template <int IMPL>
__global__ __launch_bounds__(kBlock) void k_smc_interp_step(ResampleArgs A, PlanPolicyArgs PA, InterpTable T) {
  InterpPolicy<IMPL> P;
  P.a = PA;
  P.T = T;
  resample_body<IMPL>(A, P);
}
template <int IMPL>
__global__ __launch_bounds__(kBlock) void k_smc_interp_init(PlanPolicyArgs a, uint64_t first_slot, uint64_t n_local, EmitOut em,
                                                            FilterBatch fb, InterpTable T) {
  uint64_t ltile = blockIdx.x;
  if (fb.n_filters > 1) {  // several filters per launch: tile of filter f, its key, its outputs
    const uint32_t f = (uint32_t)(ltile / fb.tiles);
    ltile -= (uint64_t)f * fb.tiles;
    a.step_key = fb.step_key[f];
    for (int k = 0; k < T.n_state; ++k) a.state_out[k] += (uint64_t)f * fb.stride;
    if (a.anc_out) a.anc_out += (uint64_t)f * fb.stride;
    select_filter_emit(em, fb, f);
  }
  const uint64_t loc = ltile * kTile + 4 * (uint64_t)threadIdx.x;
  const uint64_t gq = first_slot + loc;
  float wq[4];
  bool okq[4];
  for (int u = 0; u < 4; ++u) {
    okq[u] = loc + u < n_local;
    float ns[GJX_SMC_MAX_STATE];
    wq[u] = interp_walk<IMPL>(T, a.obs, a.step_key, (int64_t)(gq + u), nullptr, ns);
    if (okq[u]) {
      for (int k = 0; k < T.n_state; ++k) a.state_out[k][loc + u] = ns[k];
      if (a.anc_out) a.anc_out[loc + u] = (int32_t)(gq + u);
    }
  }
  emit_init_tile(wq, okq, em, loc, first_slot / kTile + ltile);
}

// HMM tables.  Alias construction in integers (DESIGN.md 3.6b states it in full): p_c = cat_fix, scaled_c = p_c K
// against Q = sum p; "small" columns (scaled < Q) in increasing order take their alias from the front "large"
// column, which gives up the difference and joins the back of the small queue once below Q.
// One 128-thread workgroup per row.  Parallel parts across the lanes: the fixed-point weights p_c, their sum Q
// (integers: any order), the thresholds (one 64-bit division per column) and the log-softmax subtraction.
// Sequential parts exactly in the spec's order: lane 0 of wave 0 runs the small / large pairing over LDS-resident
// queues; lane 0 of wave 1 meanwhile accumulates the observation row's log-sum-exp column by column.
__global__ __launch_bounds__(128) void k_hmm_prepare(const float* trans_logits, const float* obs_logits, int32_t K,
                                                     uint32_t* trans_alias, float* obs_logp) {
  __shared__ uint64_t scaled[256];  // p_c K, then (paired small columns) their accept mass
  __shared__ uint16_t small[256], large[256], alias[256];
  __shared__ float sh_m[2];
  __shared__ uint64_t sh_q[2];
  __shared__ float sh_lse;
  const int r = blockIdx.x, tid = threadIdx.x;
  const float* l = trans_logits + (size_t)r * K;
  float m = -__builtin_inff();
  for (int c = tid; c < K; c += 128) m = l[c] > m ? l[c] : m;
  m = wave_max(m);
  if ((tid & 63) == 0) sh_m[tid >> 6] = m;
  __syncthreads();
  m = sh_m[0] > sh_m[1] ? sh_m[0] : sh_m[1];
  uint64_t q = 0;
  for (int c = tid; c < K; c += 128) {
    const uint64_t p = (uint64_t)cat_fix(l[c], m);
    scaled[c] = p * (uint64_t)K;
    alias[c] = (uint16_t)c;
    q += p;
  }
  q = wave_sum(q);
  if ((tid & 63) == 0) sh_q[tid >> 6] = q;
  __syncthreads();
  const uint64_t Q = sh_q[0] + sh_q[1];
  if (tid == 0) {
    uint32_t hs = 0, ts = 0, hl = 0, tl = 0;
    for (int c = 0; c < K; ++c) {
      if (scaled[c] < Q) small[ts++] = (uint16_t)c;
      else large[tl++] = (uint16_t)c;
    }
    while (hs < ts && hl < tl) {
      const uint32_t sc = small[hs++], g = large[hl];
      alias[sc] = (uint16_t)g;  // scaled[sc] stays: it is the column's accept mass
      scaled[g] -= Q - scaled[sc];
      if (scaled[g] < Q) {
        ++hl;
        small[ts++] = (uint16_t)g;
      }
    }
  } else if (tid == 64) {
    const float* o = obs_logits + (size_t)r * K;
    sh_lse = row_lse(o, (uint32_t)K);
  }
  __syncthreads();
  uint32_t* row = trans_alias + (size_t)r * K;
  for (int c = tid; c < K; c += 128) {
    // a column that kept itself as alias accepts always (its mass ended at exactly Q, or it was never paired)
    const uint32_t thr = alias[c] == c ? 0xffffffu : (uint32_t)((scaled[c] << 24) / Q);
    row[c] = (thr << 8) | alias[c];
    obs_logp[(size_t)r * K + c] = obs_logits[(size_t)r * K + c] - sh_lse;
  }
}

// workspace carving
struct Carver {
  char* p;
  size_t left;
  bool ok = true;
  template <class T>
  T* take(size_t count) {
    const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
    if (!ok || bytes > left) {
      ok = false;
      return nullptr;
    }
    T* r = reinterpret_cast<T*>(p);
    p += bytes;
    left -= bytes;
    return r;
  }
};
inline size_t pad256(size_t b) { return (b + 255) & ~(size_t)255; }

// gjx_map_f32: the spec's exp / log and the IEEE division by a number over a column (four elements per lane)
template <int OP>
__global__ __launch_bounds__(kBlock) void k_map_f32(const float* x, float c, float* out, uint64_t n) {
  const uint64_t i0 = ((uint64_t)blockIdx.x * kBlock + threadIdx.x) * 4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint64_t i = i0 + r;
    if (i < n) {
      const float v = x[i];
      out[i] = OP == GJX_MAP_EXP ? e_exp(v) : OP == GJX_MAP_LOG ? m_log(v) : OP == GJX_MAP_DIV ? v / c : OP == GJX_MAP_RDIV ? c / v
               : OP == GJX_MAP_SQRT ? __builtin_sqrtf(v) : __builtin_fabsf(v);
    }
  }
}
}  // namespace

// ================================================================================================
// C-ABI
// ================================================================================================
extern "C" {

int gjx_version(int* major, int* minor) {
  if (major) *major = GJX_VERSION_MAJOR;
  if (minor) *minor = GJX_VERSION_MINOR;
  return GJX_OK;
}
const char* gjx_backend_name(void) { return "hip-gfx950"; }
int gjx_frac_bits(uint64_t n_total) { return frac_bits(n_total); }
uint64_t gjx_smc_tile(void) { return kTile; }
uint64_t gjx_num_tiles(uint64_t n) { return ntiles_of(n); }
uint64_t gjx_num_max_partials(uint64_t n) { return nrows_of(n); }

size_t gjx_workspace_bytes(int op, uint64_t n) {
  const uint64_t nt = ntiles_of(n);
  switch (op) {
    case GJX_OP_LOGSUMEXP: return pad256(nrows_of(n) * 4) + pad256(nt * 8) + 1024;
    case GJX_OP_CATEGORICAL_INDEX:
    case GJX_OP_RESAMPLE:  // (multinomial: tile maxima, masses, global CDF; systematic: in-tile CDF, records, prefix)
      return pad256(nt * 4) + 2 * pad256(nt * 8) + pad256(n * 8) + pad256(nt * 160) + pad256(prefix_words(nt) * 8) + 2048;
    case GJX_OP_SMC:  // the second copy of up to 4 state columns and the log-weights, two columns of fixed-point weights,
                      // two sets of records and merged prefixes, HMM tables
      return (GJX_SMC_MAX_STATE + 3) * pad256(n * 4) + 2 * (pad256(nt * 16) + pad256(nt * 128) + pad256(nt * 16)) + 2 * pad256(prefix_words(nt) * 8) + 4096 +
             pad256(256 * (256 + 64) * 4) + pad256(256 * 256 * 4) + 1024;
    default: return 0;
  }
}

#define GJX_DISPATCH_IMPL(impl, KERNEL, ...)  \
  do {                                        \
    if ((impl) == 0) KERNEL<0> __VA_ARGS__;   \
    else KERNEL<1> __VA_ARGS__;               \
  } while (0)

int gjx_rng_keys(const gjx_keys* k, uint64_t n, uint32_t* out, gjx_stream s) {
  if (!keys_ok(k) || (!out && n)) return GJX_ERR_INVALID;
  if (n == 0) return GJX_OK;
  GJX_DISPATCH_IMPL(k->impl, k_rng_keys, <<<grid_for(n), kBlock, 0, S(s)>>>(key_src(k), n, out));
  return launch_status();
}
int gjx_rng_split_each(const gjx_keys* k, uint64_t n, uint32_t m, uint32_t* out, gjx_stream s) {
  if (!keys_ok(k) || (!out && n) || m == 0) return GJX_ERR_INVALID;
  if (n == 0) return GJX_OK;
  GJX_DISPATCH_IMPL(k->impl, k_rng_split_each, <<<grid_for(n * m), kBlock, 0, S(s)>>>(key_src(k), n, m, out));
  return launch_status();
}
int gjx_rng_bits(const gjx_keys* k, uint32_t sub, uint64_t n, uint32_t* out, gjx_stream s) {
  if (!keys_ok(k) || (!out && n)) return GJX_ERR_INVALID;
  if (n == 0) return GJX_OK;
  GJX_DISPATCH_IMPL(k->impl, k_rng_bits, <<<grid_for(n), kBlock, 0, S(s)>>>(key_src(k), sub, n, out));
  return launch_status();
}

int gjx_sample_logpdf_normal(const gjx_keys* k, gjx_f32 loc, gjx_f32 scale, float* value_out,
                             float* score_out, uint64_t n, gjx_stream s) {
  if (!keys_ok(k) || !value_out) return GJX_ERR_INVALID;
  if (n == 0) return GJX_OK;
  GJX_DISPATCH_IMPL(k->impl, k_sample_normal,
                    <<<grid_for(n), kBlock, 0, S(s)>>>(key_src(k), opnd(loc), opnd(scale), value_out, score_out, n));
  return launch_status();
}
int gjx_sample_logpdf_gamma(const gjx_keys* k, gjx_f32 concentration, gjx_f32 rate,
                            float* value_out, float* score_out, uint64_t n, gjx_stream s) {
  if (!keys_ok(k) || !value_out) return GJX_ERR_INVALID;
  if (n == 0) return GJX_OK;
  GJX_DISPATCH_IMPL(k->impl, k_sample_gamma,
                    <<<grid_for(n), kBlock, 0, S(s)>>>(key_src(k), opnd(concentration), opnd(rate), value_out, score_out, n));
  return launch_status();
}
int gjx_sample_logpdf_beta(const gjx_keys* k, gjx_f32 a, gjx_f32 b, float* value_out,
                           float* score_out, uint64_t n, gjx_stream s) {
  if (!keys_ok(k) || !value_out) return GJX_ERR_INVALID;
  if (n == 0) return GJX_OK;
  GJX_DISPATCH_IMPL(k->impl, k_sample_beta,
                    <<<grid_for(n), kBlock, 0, S(s)>>>(key_src(k), opnd(a), opnd(b), value_out, score_out, n));
  return launch_status();
}
int gjx_sample_logpdf_bernoulli(const gjx_keys* k, gjx_f32 probs, uint8_t* value_out,
                                float* score_out, uint64_t n, gjx_stream s) {
  if (!keys_ok(k) || !value_out) return GJX_ERR_INVALID;
  if (n == 0) return GJX_OK;
  GJX_DISPATCH_IMPL(k->impl, k_sample_bernoulli,
                    <<<grid_for(n), kBlock, 0, S(s)>>>(key_src(k), opnd(probs), value_out, score_out, n));
  return launch_status();
}
int gjx_sample_logpdf_categorical(const gjx_keys* k, const float* logits, uint64_t n_rows,
                                  uint32_t n_cat, const int32_t* row_index, int mode,
                                  int32_t* value_out, float* score_out, uint64_t n, gjx_stream s) {
  if (!keys_ok(k) || !value_out || !logits || n_cat == 0 || (mode != 0 && mode != 1))
    return GJX_ERR_INVALID;
  if (n == 0) return GJX_OK;
  GJX_DISPATCH_IMPL(k->impl, k_sample_categorical,
                    <<<grid_for(n), kBlock, 0, S(s)>>>(key_src(k), logits, n_rows, n_cat, row_index, mode, value_out, score_out, n));
  return launch_status();
}

int gjx_logpdf_normal(gjx_f32 value, gjx_f32 loc, gjx_f32 scale, float* score_out, uint64_t n,
                      gjx_stream s) {
  if (!score_out) return GJX_ERR_INVALID;
  if (n == 0) return GJX_OK;
  k_logpdf_normal<<<grid_for(n), kBlock, 0, S(s)>>>(opnd(value), opnd(loc), opnd(scale), score_out, n);
  return launch_status();
}
int gjx_logpdf_gamma(gjx_f32 value, gjx_f32 concentration, gjx_f32 rate, float* score_out,
                     uint64_t n, gjx_stream s) {
  if (!score_out) return GJX_ERR_INVALID;
  if (n == 0) return GJX_OK;
  k_logpdf_gamma<<<grid_for(n), kBlock, 0, S(s)>>>(opnd(value), opnd(concentration), opnd(rate), score_out, n);
  return launch_status();
}
int gjx_logpdf_beta(gjx_f32 value, gjx_f32 a, gjx_f32 b, float* score_out, uint64_t n,
                    gjx_stream s) {
  if (!score_out) return GJX_ERR_INVALID;
  if (n == 0) return GJX_OK;
  k_logpdf_beta<<<grid_for(n), kBlock, 0, S(s)>>>(opnd(value), opnd(a), opnd(b), score_out, n);
  return launch_status();
}
int gjx_logpdf_bernoulli(const uint8_t* value, int value_scalar, gjx_f32 probs, float* score_out,
                         uint64_t n, gjx_stream s) {
  if (!score_out) return GJX_ERR_INVALID;
  if (n == 0) return GJX_OK;
  k_logpdf_bernoulli<<<grid_for(n), kBlock, 0, S(s)>>>(value, value_scalar, opnd(probs), score_out, n);
  return launch_status();
}
int gjx_logpdf_categorical(const int32_t* value, int value_scalar, const float* logits,
                           uint64_t n_rows, uint32_t n_cat, const int32_t* row_index,
                           float* score_out, uint64_t n, gjx_stream s) {
  if (!score_out || !logits || n_cat == 0) return GJX_ERR_INVALID;
  if (n == 0) return GJX_OK;
  k_logpdf_categorical<<<grid_for(n), kBlock, 0, S(s)>>>(value, value_scalar, logits, n_rows, n_cat, row_index, score_out, n);
  return launch_status();
}

// ---- plans ---------------------------------------------------------------------------------------
// ---- per-row tables of categorical sites (specialised kernels) ------------------------------------------------------
__global__ void k_cat_prepare(const float* logits, uint32_t n_rows, uint32_t K, uint2* ent, uint4* guide4, float* logp_t, int gbits) {
  // one thread per row, sequential in the category exactly as cat_invcdf / row_lse state it
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const float* l = logits + (size_t)r * K;
  const float lse = row_lse(l, K);
  if (logp_t) {
    for (uint32_t c = 0; c < K; ++c) logp_t[(size_t)c * n_rows + r] = l[c] - lse;
    return;
  }
  const float m = row_max(l, K);
  uint2* row = ent + (size_t)r * K;
  uint32_t C = 0;
  for (uint32_t c = 0; c < K; ++c) {
    C += cat_fix(l[c], m);
    row[c] = make_uint2(C, f2u(l[c] - lse));
  }
  // guide[g] (r04): the draws of bucket g are bits in [g << sh, (g + 1) << sh).  c = the category of the bucket's SMALLEST draw
  // (the threshold is monotone in the draw, so every draw of the bucket lands at or after c).  A draw maps to c exactly while
  // floor(bits C_tot / 2^32) < CDF_c, i.e. bits <= (CDF_c 2^32 - 1) / C_tot (integers): that bound, c, the next category
  // with mass and both log-densities ARE the bucket — one load decides the draw unless the bucket's LAST draw lands beyond
  // the next category (flag: the walk of the spec goes on from there, rare once buckets are finer than categories).
  const int sh = 32 - gbits;
  const uint32_t nb = 1u << gbits;
  uint32_t c = 0;
  for (uint32_t g = 0; g < nb; ++g) {
    const uint64_t thr = ((uint64_t)(g << sh) * (uint64_t)C) >> 32;
    while (c < K - 1 && (uint64_t)row[c].x <= thr) ++c;
    uint32_t cn = c;  // the next category with mass (c itself if there is none: never taken then)
    for (uint32_t k = c + 1; k < K; ++k)
      if (row[k].x > row[c].x) { cn = k; break; }
    const uint32_t last_bits = (uint32_t)((((uint64_t)g + 1) << sh) - 1);
    const uint64_t thr_last = ((uint64_t)last_bits * (uint64_t)C) >> 32;
    const bool multi = cn != c && (uint64_t)row[cn].x <= thr_last;  // the bucket's last draw lies beyond cn
    const uint32_t bm1 = C == 0 ? 0xffffffffu : (uint32_t)(((((uint64_t)row[c].x) << 32) - 1) / (uint64_t)C);
    guide4[((size_t)r << gbits) + g] = make_uint4(row[c].x == 0 ? 0xffffffffu : bm1, c | (cn << 9) | ((multi ? 1u : 0u) << 18), row[c].y, row[cn].y);
  }
}
// Build the tables of every categorical site of a table (first compilation of a plan: a GPU is present by then).
// K <= 511 keeps the inclusive CDF inside 32 bits (cat_fix <= 2^23).  Failure to allocate leaves the on-the-fly path.
static void cat_tables_prepare(CSite* sites, int n, std::vector<void*>* owned) {
  bool any = false;
  for (int q = 0; q < n; ++q) {
    CSite& st = sites[q];
    if (st.dist != GJX_DIST_CATEGORICAL || st.cat_ent || st.cat_logp_t || !st.logits || st.n_cat > 511 || st.n_rows < 1) continue;
    const size_t rows = (size_t)st.n_rows;
    const unsigned grid = ((unsigned)rows + 63) / 64;
    // (one thread per row: rows are few — a transition / emission matrix — and this runs once per plan)
    if (st.observed && st.obs.kind != GJX_ARG_INPUT) {  // the value is launch-uniform: the transposed log-probabilities only
      float* lt = nullptr;
      if (hipMalloc(&lt, sizeof(float) * rows * (size_t)st.n_cat) != hipSuccess) { (void)hipGetLastError(); continue; }
      k_cat_prepare<<<grid, 64>>>(st.logits, (uint32_t)rows, (uint32_t)st.n_cat, nullptr, nullptr, lt, 0);
      owned->push_back(lt);
      st.cat_logp_t = lt;
      any = true;
      continue;
    }
    uint2* ent = nullptr;
    uint4* guide4 = nullptr;
    // buckets per row: finer than the categories (so that a bucket rarely holds more than two), within a table of a few MB
    // that the L2s keep (GJX_CAT_GUIDE_BITS: tuning knob)
    static const int gknob = [] { const char* e = std::getenv("GJX_CAT_GUIDE_BITS"); return e ? atoi(e) : 0; }();
    int gbits = 8;
    while ((1 << gbits) < 4 * st.n_cat && gbits < 11) ++gbits;
    while (gbits > 8 && (sizeof(uint4) << gbits) * rows > ((size_t)2 << 20)) --gbits;
    if (gknob >= 4 && gknob <= 16) gbits = gknob;
    if (hipMalloc(&ent, sizeof(uint2) * rows * (size_t)st.n_cat) != hipSuccess || hipMalloc(&guide4, (sizeof(uint4) << gbits) * rows) != hipSuccess) {
      (void)hipGetLastError();
      if (ent) (void)hipFree(ent);
      continue;
    }
    k_cat_prepare<<<grid, 64>>>(st.logits, (uint32_t)rows, (uint32_t)st.n_cat, ent, guide4, nullptr, gbits);
    owned->push_back(ent);
    owned->push_back(guide4);
    st.cat_ent = ent;
    st.cat_guide4 = guide4;
    st.cat_gbits = gbits;
    any = true;
  }
  if (any && hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
}
static void free_owned(std::vector<void*>& owned) {
  for (void* p : owned) (void)hipFree(p);
  owned.clear();
}

// A plan's own copy of the programs of one site table (the caller's arrays need not outlive plan creation; only the
// host-side code generator reads them: plans with programs run as specialised kernels only).
struct ExprStore {
  gjx_expr_op ops[GJX_MAX_SITES][2][GJX_MAX_EXPR_OPS];
};

struct gjx_plan {
  int n_sites;
  int n_slots;
  int dist_mask;
  uint32_t flags;  // GJX_PLAN_*
  int n_params;    // values set by gjx_plan_set_params
  int max_param;   // highest GJX_ARG_PARAM index referenced (-1: none)
  PlanParams prm;
  CSite host[GJX_MAX_SITES];
  CSite* dev;
  // specialised kernels, built on first use: [0] THREEFRY, [1] PHILOX one particle per lane, [2] PHILOX pairs
  // (two adjacent particles per lane), [3] PHILOX quads (four per lane: one wave per 256-particle row)
  gjx_jit::Compiled jit[4];
  std::vector<void*> dev_owned;  // per-row tables of categorical sites (specialised kernels)
  std::mutex jit_mu;
  ExprStore expr;      // GJX_ARG_EXPR programs (host; read by the code generator)
  bool has_expr;       // ... any?  Then the plan runs as a specialised kernel only
  int expr_max_input;  // highest input column a program reads (-1: none)
  gjx_jit::ScopeInfo scopes;  // nested calls (gjx_plan_create_scoped); n_scopes > 0: a specialised kernel only, like programs
};

// GJX_ARG_EXPR (gjx.h): a postfix program as a distribution argument.  Well-formed: at most GJX_MAX_EXPR_OPS entries,
// operands in range for the plan kind, the stack never deeper than 8, exactly one value left.
static bool expr_ok(const gjx_arg& a, int s, int n_state, int n_obs, bool allow_state) {
  const gjx_expr_op* ops = reinterpret_cast<const gjx_expr_op*>(a.table);
  if (!ops || a.ref < 1 || a.ref > GJX_MAX_EXPR_OPS) return false;
  int depth = 0;
  for (int k = 0; k < a.ref; ++k) {
    const int r = ops[k].ref;
    switch (ops[k].op) {
      case GJX_EXPR_CONST: ++depth; break;
      case GJX_EXPR_SITE: if (r < 0 || r >= s) return false; ++depth; break;
      case GJX_EXPR_INPUT: if (n_state >= 0 || r < 0 || r >= 16) return false; ++depth; break;
      case GJX_EXPR_PARAM: if (n_state >= 0 || r < 0 || r >= GJX_MAX_PARAMS) return false; ++depth; break;
      case GJX_EXPR_STATE: if (n_state < 0 || !allow_state || r < 0 || r >= n_state) return false; ++depth; break;
      case GJX_EXPR_OBS: if (n_state < 0 || r < 0 || r >= n_obs) return false; ++depth; break;
      case GJX_EXPR_ADD: case GJX_EXPR_SUB: case GJX_EXPR_MUL: case GJX_EXPR_DIV: case GJX_EXPR_MAX: case GJX_EXPR_MIN:
      case GJX_EXPR_LT: case GJX_EXPR_LE: case GJX_EXPR_EQ:
        if (depth < 2) return false;
        --depth;
        break;
      case GJX_EXPR_SELECT:
        if (depth < 3) return false;
        depth -= 2;
        break;
      case GJX_EXPR_NEG: case GJX_EXPR_EXP: case GJX_EXPR_LOG: case GJX_EXPR_SQRT: case GJX_EXPR_ABS: if (depth < 1) return false; break;
      default: return false;
    }
    if (depth > 8) return false;
  }
  return depth == 1;
}

// Argument validity.  n_state / n_obs > -1 switch on the SMC-plan kinds (STATE only if allow_state).
static bool arg_ok(const gjx_arg& a, int s, int n_state = -1, int n_obs = -1, bool allow_state = false) {
  switch (a.kind) {
    case GJX_ARG_EXPR: return expr_ok(a, s, n_state, n_obs, allow_state);
    case GJX_ARG_CONST: return true;
    case GJX_ARG_SITE: return a.ref >= 0 && a.ref < s;
    case GJX_ARG_INPUT: return n_state < 0 && a.ref >= 0 && a.ref < 16;
    case GJX_ARG_TABLE: return a.ref >= 0 && a.ref < s && a.table != nullptr;
    case GJX_ARG_STATE: return allow_state && a.ref >= 0 && a.ref < n_state;
    case GJX_ARG_OBS: return n_obs >= 0 && a.ref >= 0 && a.ref < n_obs;
    case GJX_ARG_PARAM: return n_state < 0 && a.ref >= 0 && a.ref < GJX_MAX_PARAMS;
    default: return false;
  }
}
static CArg carg(const gjx_arg& a) { return CArg{a.kind, a.ref, 0, a.ref, a.scale, a.offset, a.table}; }

// Validate one site and convert it (hoisting per-site constants with the same spec functions).
static bool convert_site(const gjx_site& st, int s, CSite& c, int n_state = -1, int n_obs = -1,
                         bool allow_state = false) {
  bool ok = st.dist >= 0 && st.dist <= GJX_DIST_CATEGORICAL && arg_ok(st.arg[0], s, n_state, n_obs, allow_state);
  const bool two_args = st.dist != GJX_DIST_BERNOULLI && st.dist != GJX_DIST_CATEGORICAL;
  if (ok && two_args) ok = arg_ok(st.arg[1], s, n_state, n_obs, allow_state);
  if (ok && st.observed) {
    if (n_state < 0) ok = st.obs.kind == GJX_ARG_CONST || (st.obs.kind == GJX_ARG_INPUT && st.obs.ref >= 0 && st.obs.ref < 16) ||
                          (st.obs.kind == GJX_ARG_PARAM && st.obs.ref >= 0 && st.obs.ref < GJX_MAX_PARAMS);
    else ok = st.obs.kind == GJX_ARG_CONST || (st.obs.kind == GJX_ARG_OBS && st.obs.ref >= 0 && st.obs.ref < n_obs);
  }
  if (ok && st.dist == GJX_DIST_CATEGORICAL)
    ok = st.logits && st.n_cat > 0 && st.n_rows > 0 && (st.cat_mode == 0 || st.cat_mode == 1) && st.arg[0].kind != GJX_ARG_EXPR;
  if (!ok) return false;
  memset(&c, 0, sizeof(c));
  c.dist = st.dist; c.observed = st.observed; c.out_col = st.out_col;
  c.n_cat = st.n_cat; c.n_rows = st.n_rows; c.cat_mode = st.cat_mode;
  c.slot = -1;
  c.a0 = carg(st.arg[0]); c.a1 = carg(st.arg[1]); c.obs = carg(st.obs);
  if (!two_args) c.a1.kind = GJX_ARG_CONST;
  c.logits = st.logits;
  // Hoist per-site constants: same spec functions, evaluated once on the host (IEEE-exact ops
  // give the same bits as evaluating them per particle on the device).
  const bool c0 = st.arg[0].kind == GJX_ARG_CONST, c1 = st.arg[1].kind == GJX_ARG_CONST;
  const bool u0 = c0 || st.arg[0].kind == GJX_ARG_PARAM, u1 = c1 || st.arg[1].kind == GJX_ARG_PARAM;  // launch-uniform
  if (st.dist == GJX_DIST_NORMAL && c1) {
    c.pre = 1; c.pre0 = normal_rs(st.arg[1].offset); c.pre1 = normal_lognorm(st.arg[1].offset);
  } else if (st.dist == GJX_DIST_GAMMA && c0 && c1) {
    c.pre = 1; c.pre1 = gamma_lognorm(st.arg[0].offset, st.arg[1].offset);
  } else if (st.dist == GJX_DIST_BETA && c0 && c1) {
    c.pre = 1; c.pre1 = beta_lbeta(st.arg[0].offset, st.arg[1].offset);
  } else if ((st.dist == GJX_DIST_NORMAL && u1) || ((st.dist == GJX_DIST_GAMMA || st.dist == GJX_DIST_BETA) && u0 && u1)) {
    c.pre = 2;  // the same constants, derived from the launch's parameters (plan_derive_params)
  }
  return true;
}

// programs of state arguments (init_state / next_state of SMC and scan plans)
struct StateExprStore {
  gjx_expr_op ops[GJX_SMC_MAX_STATE][GJX_MAX_EXPR_OPS];
};
static void state_expr_adopt(CArg* args, int n, StateExprStore* store) {
  for (int k = 0; k < n; ++k)
    if (args[k].kind == GJX_ARG_EXPR) {
      memcpy(store->ops[k], args[k].table, sizeof(gjx_expr_op) * (size_t)args[k].ref);
      args[k].table = reinterpret_cast<const float*>(store->ops[k]);
    }
}
static bool expr_adopt(CSite* sites, int n, ExprStore* store) {  // -> does the table hold any program?
  bool any = false;
  for (int q = 0; q < n; ++q) {
    CArg* as[2] = {&sites[q].a0, &sites[q].a1};
    for (int k = 0; k < 2; ++k)
      if (as[k]->kind == GJX_ARG_EXPR) {
        memcpy(store->ops[q][k], as[k]->table, sizeof(gjx_expr_op) * (size_t)as[k]->ref);
        as[k]->table = reinterpret_cast<const float*>(store->ops[q][k]);
        any = true;
      }
  }
  return any;
}
// highest operand index of `opcode` in a site table's programs (-1: none)
static int expr_max_ref(const CSite* sites, int n, int opcode) {
  int mx = -1;
  for (int q = 0; q < n; ++q) {
    const CArg* as[2] = {&sites[q].a0, &sites[q].a1};
    for (int k = 0; k < 2; ++k)
      if (as[k]->kind == GJX_ARG_EXPR) {
        const gjx_expr_op* ops = reinterpret_cast<const gjx_expr_op*>(as[k]->table);
        for (int i = 0; i < as[k]->ref; ++i)
          if (ops[i].op == opcode && ops[i].ref > mx) mx = ops[i].ref;
      }
  }
  return mx;
}

int gjx_plan_create(const gjx_site* sites, int n_sites, gjx_plan** out) { return gjx_plan_create_ex(sites, n_sites, 0u, out); }
int gjx_plan_create_ex(const gjx_site* sites, int n_sites, uint32_t flags, gjx_plan** out) {
  if (!sites || !out || n_sites <= 0 || n_sites > GJX_MAX_SITES || (flags & ~(uint32_t)GJX_PLAN_FAST_MATH)) return GJX_ERR_INVALID;
  gjx_plan* p = new (std::nothrow) gjx_plan;
  if (!p) return GJX_ERR_LAUNCH;
  p->flags = flags;
  p->n_params = 0;
  p->max_param = -1;
  memset(&p->prm, 0, sizeof p->prm);
  p->n_sites = n_sites;
  p->dev = nullptr;
  p->dist_mask = 0;
  int last_use[GJX_MAX_SITES];
  for (int s = 0; s < n_sites; ++s) last_use[s] = -1;
  for (int s = 0; s < n_sites; ++s) {
    CSite& c = p->host[s];
    if (!convert_site(sites[s], s, c)) {
      delete p;
      return GJX_ERR_INVALID;
    }
    p->dist_mask |= 1 << sites[s].dist;
    for (CArg* a : {&c.a0, &c.a1, &c.obs})
      if (a->kind == GJX_ARG_PARAM && a->ref > p->max_param && (a != &c.obs || c.observed)) p->max_param = a->ref;
    for (CArg* a : {&c.a0, &c.a1})
      if (a->kind == GJX_ARG_SITE || a->kind == GJX_ARG_TABLE) last_use[a->ref] = s;
  }
  // LDS slots by liveness (linear scan): a value occupies a slot from its site to its last use.
  int slot_free_at[GJX_MAX_SITES];  // slot -> first site index at which it is free again
  int n_slots = 0;
  int site_slot[GJX_MAX_SITES];
  for (int s = 0; s < n_sites; ++s) {
    site_slot[s] = -1;
    CSite& c = p->host[s];
    // translate this site's references BEFORE taking a slot for its own value
    for (CArg* a : {&c.a0, &c.a1})
      if (a->kind == GJX_ARG_SITE || a->kind == GJX_ARG_TABLE) {
        a->ref_is_int = p->host[a->ref].dist >= GJX_DIST_BERNOULLI;
        a->ref = site_slot[a->ref];
      }
    if (last_use[s] < 0) continue;
    int slot = -1;
    for (int k = 0; k < n_slots; ++k)
      if (slot_free_at[k] <= s) { slot = k; break; }
    if (slot < 0) slot = n_slots++;
    slot_free_at[slot] = last_use[s] + 1;  // reusable by sites after the last reader
    site_slot[s] = slot;
    c.slot = slot;
  }
  p->n_slots = n_slots;
  p->has_expr = expr_adopt(p->host, n_sites, &p->expr);
  p->expr_max_input = expr_max_ref(p->host, n_sites, GJX_EXPR_INPUT);
  const int mp = expr_max_ref(p->host, n_sites, GJX_EXPR_PARAM);
  if (mp > p->max_param) p->max_param = mp;
  *out = p;  // the interpreter's device copy of the table is made on first use (plan_device_table)
  return GJX_OK;
}

int gjx_plan_create_scoped(const gjx_site* sites, int n_sites, const gjx_scope* scopes, int n_scopes, uint32_t flags,
                           gjx_plan** out) {
  gjx_plan* p = nullptr;
  const int rc = gjx_plan_create_ex(sites, n_sites, flags, &p);
  if (rc) return rc;
  if (!gjx_jit::derive_scopes(sites, n_sites, scopes, n_scopes, p->scopes)) {
    gjx_plan_destroy(p);
    return GJX_ERR_INVALID;
  }
  *out = p;
  return GJX_OK;
}

// Launch parameters: the caller's values and the per-site constants that depend on them (the spec functions on the
// host: IEEE-exact ops give the bits the device would compute per particle).
int gjx_plan_set_params(gjx_plan* p, const float* params, int n_params) {
  if (!p || n_params < 0 || n_params > GJX_MAX_PARAMS || (n_params && !params) || n_params <= p->max_param) return GJX_ERR_INVALID;
  for (int k = 0; k < n_params; ++k) p->prm.p[k] = params[k];
  p->n_params = n_params;
  auto val = [&](const CArg& a) {  // CONST or PARAM
    if (a.kind == GJX_ARG_CONST) return a.offset;
    const float t = a.scale * p->prm.p[a.ref];
    return t + a.offset;
  };
  for (int q = 0; q < p->n_sites; ++q) {
    const CSite& c = p->host[q];
    if (c.pre != 2) continue;
    if (c.dist == GJX_DIST_NORMAL) {
      p->prm.d[2 * q] = normal_rs(val(c.a1));
      p->prm.d[2 * q + 1] = normal_lognorm(val(c.a1));
    } else if (c.dist == GJX_DIST_GAMMA) {
      p->prm.d[2 * q + 1] = gamma_lognorm(val(c.a0), val(c.a1));
    } else {
      p->prm.d[2 * q + 1] = beta_lbeta(val(c.a0), val(c.a1));
    }
  }
  return GJX_OK;
}

// Device copy of the site table for the interpreter kernel (not needed by specialised kernels).
static int plan_device_table(gjx_plan* p) {
  std::lock_guard<std::mutex> lock(p->jit_mu);
  if (p->dev) return GJX_OK;
  if (hipMalloc(&p->dev, sizeof(CSite) * (size_t)p->n_sites) != hipSuccess) {
    p->dev = nullptr;
    return GJX_ERR_NO_DEVICE;
  }
  if (hipMemcpy(p->dev, p->host, sizeof(CSite) * (size_t)p->n_sites, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(p->dev);
    p->dev = nullptr;
    return GJX_ERR_LAUNCH;
  }
  return GJX_OK;
}

static int jit_form_pref() {  // GJX_JIT_FORM = one | pair | quad (test / tuning knob, read at every launch); default: the measured best
  const char* e = std::getenv("GJX_JIT_FORM");
  if (e && !strcmp(e, "one")) return 1;
  if (e && !strcmp(e, "pair")) return 2;
  if (e && !strcmp(e, "quad")) return 4;
  const char* old = std::getenv("GJX_JIT_PAIRED");  // 0: always the one-particle-per-lane form
  if (old && old[0] == '0') return 1;
  return 4;  // measured (tools/ab_importance.py, 1e6 particles): quad 13.1 / pair 13.6 / one 28.3 us per pass at 8 passes per launch
}
int gjx_plan_specialized_source(const gjx_plan* p, int impl, char* buf, size_t buf_len, size_t* needed) {
  if (!p || (impl != 0 && impl != 1)) return GJX_ERR_INVALID;
  gjx_jit::Gen<CSite, CArg> g;
  g.impl = impl; g.sites = p->host; g.n_sites = p->n_sites; g.laned = impl == 1 && jit_form_pref() >= 2;
  g.sc = p->scopes.n_scopes > 0 ? &p->scopes : nullptr;
  g.pairs_per_lane = jit_form_pref() == 4 ? 2 : 1;  // (GJX_JIT_FORM picks the PHILOX form shown)
  g.fast_math = (p->flags & GJX_PLAN_FAST_MATH) != 0;
  gjx_jit::TableScope ts;
  const std::string src = g.run();
  if (needed) *needed = src.size() + 1;
  if (buf && buf_len > 0) {
    const size_t k = src.size() < buf_len - 1 ? src.size() : buf_len - 1;
    memcpy(buf, src.data(), k);
    buf[k] = 0;
  }
  return GJX_OK;
}

int gjx_plan_compile_check(const gjx_plan* p, int impl) {
  if (!p || (impl != 0 && impl != 1)) return GJX_ERR_INVALID;
  for (int form = 0; form <= 2 * impl; ++form) {  // PHILOX: one particle per lane, pairs, quads
    gjx_jit::Gen<CSite, CArg> g;
    g.impl = impl; g.sites = p->host; g.n_sites = p->n_sites; g.laned = form != 0; g.pairs_per_lane = form == 2 ? 2 : 1;
    g.sc = p->scopes.n_scopes > 0 ? &p->scopes : nullptr;
    g.fast_math = (p->flags & GJX_PLAN_FAST_MATH) != 0;
    gjx_jit::TableScope ts;
    if (!gjx_jit::compile_only(g.run())) return GJX_ERR_UNSUPPORTED;
  }
  return GJX_OK;
}
int gjx_plan_destroy(gjx_plan* p) {
  if (!p) return GJX_OK;
  if (p->dev) (void)hipFree(p->dev);
  for (auto& c : p->jit) gjx_jit::release(&c);  // the modules stay cached (bounded, LRU) for plans of the same structure
  free_owned(p->dev_owned);
  delete p;
  return GJX_OK;
}

// The hiprtc-specialised kernel of a plan for this key form (compiled and loaded on first use).
// `lane_particles`: how many adjacent particles a lane may own given n and the alignment of the output buffers (1, 2, 4).
static gjx_jit::Compiled& plan_compiled(gjx_plan* mp, const gjx_keys* pk, int lane_particles) {
  // PHILOX children of a lane-0 key share one cipher key, and an even first index keeps particle pairs
  // (2i, 2i+1) together: the paired / quad kernel forms (gjx_plan_jit.hpp)
  const bool pairable = pk->impl == 1 && pk->mode == 1 && pk->parent_lane == 0 && (pk->first & 1) == 0;
  int P = pairable ? (lane_particles < jit_form_pref() ? lane_particles : jit_form_pref()) : 1;
  if (P == 3) P = 2;
  const bool laned = P >= 2;
  gjx_jit::Compiled& c = mp->jit[P == 4 ? 3 : (laned ? 2 : pk->impl)];
  if (c.state == 0) {
    std::lock_guard<std::mutex> lock(mp->jit_mu);
    if (c.state == 0) {
      cat_tables_prepare(mp->host, mp->n_sites, &mp->dev_owned);
      auto make = [&](int min_waves) {
        gjx_jit::Gen<CSite, CArg> g;
        g.impl = pk->impl; g.sites = mp->host; g.n_sites = mp->n_sites; g.laned = laned; g.pairs_per_lane = P == 4 ? 2 : 1;
        g.sc = mp->scopes.n_scopes > 0 ? &mp->scopes : nullptr;
        g.fast_math = (mp->flags & GJX_PLAN_FAST_MATH) != 0;
        g.min_waves = min_waves;
        gjx_jit::TableScope ts;  // the source numbers the plan's device tables; the addresses travel as a kernel argument
        std::string src = g.run();
        c.block = g.block;
        c.rows_per_block = g.rows_per_block;
        c.tabs = ts.reg.tables();
        return src;
      };
      // The kernels are bound by dependency latency, not by issue slots (a wave64 VALU instruction issues in ~2.4
      // cycles, tools/microbench/valu_rate.hip): a sixth wave per SIMD (<= 80 VGPRs) is worth 2-3 % on the paired
      // form as long as the allocator gets there with (next to) no spilling; otherwise the unconstrained build is kept.
      const char* e = std::getenv("GJX_JIT_MIN_WAVES");  // test knob: force the hint (0 = none)
      const int hint = e ? atoi(e) : (P == 2 ? 6 : 0);
      bool ok = gjx_jit::compile(make(hint), pk->impl, &c);
      if (ok && !e && hint > 0) {
        int scratch = 0;
        const hipError_t qe = hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, c.fn);
        if (qe != hipSuccess) (void)hipGetLastError();  // a failed query must not surface as a launch error later
        if (std::getenv("GJX_PLAN_JIT_VERBOSE")) fprintf(stderr, "gjx jit: waves-per-SIMD hint %d: query %d, scratch %d B\n", hint, (int)qe, scratch);
        if (qe != hipSuccess || scratch > 32) {  // (a couple of spilled words cost less than the lost wave)
          ok = gjx_jit::compile(make(0), pk->impl, &c);  // (releases the hinted module: it stays cached, unreferenced)
        }
      }
      if (!ok) {
        (void)hipGetLastError();
        fprintf(stderr, "[gjx] plan specialisation FAILED (hiprtc compile or module load; impl %d, %d particle(s) per lane). "
                        "gjx_importance_run returns GJX_ERR_JIT; set GJX_PLAN_JIT_VERBOSE=1 for the compiler log, GJX_PLAN_JIT=0 "
                        "or GJX_PLAN_JIT_FALLBACK=1 to run the (7x slower) table interpreter instead.\n", pk->impl, P);
      }
      c.state = ok ? 1 : -1;
    }
  }
  return c;
}
static bool jit_fallback_allowed() {
  static const bool allow = [] {
    const char* e = std::getenv("GJX_PLAN_JIT_FALLBACK");
    return e && e[0] == '1';
  }();
  return allow;
}
int gjx_plan_prepare(gjx_plan* p, const gjx_keys* pk) {
  if (!p || !keys_ok(pk)) return GJX_ERR_INVALID;
  if (!gjx_jit::enabled()) return p->has_expr ? GJX_ERR_UNSUPPORTED : plan_device_table(p);
  // every form a launch may take (which one depends on n and on buffer alignment)
  bool ok = plan_compiled(p, pk, 1).state == 1;
  ok = plan_compiled(p, pk, 2).state == 1 && ok;
  ok = plan_compiled(p, pk, 4).state == 1 && ok;
  if (ok) return GJX_OK;
  return jit_fallback_allowed() && !p->has_expr ? plan_device_table(p) : GJX_ERR_JIT;
}

// One launch of a plan over n_pass independent passes (n_pass == 1: the plain call).
int gjx_map_f32(int op, const float* x, float c, float* out, uint64_t n, gjx_stream s) {
  if (!x || !out || op < GJX_MAP_EXP || op > GJX_MAP_ABS) return GJX_ERR_INVALID;
  if (n == 0) return GJX_OK;
  const unsigned grid = (unsigned)((n + 4ull * kBlock - 1) / (4ull * kBlock));
  switch (op) {
    case GJX_MAP_EXP: k_map_f32<GJX_MAP_EXP><<<grid, kBlock, 0, S(s)>>>(x, c, out, n); break;
    case GJX_MAP_LOG: k_map_f32<GJX_MAP_LOG><<<grid, kBlock, 0, S(s)>>>(x, c, out, n); break;
    case GJX_MAP_DIV: k_map_f32<GJX_MAP_DIV><<<grid, kBlock, 0, S(s)>>>(x, c, out, n); break;
    case GJX_MAP_SQRT: k_map_f32<GJX_MAP_SQRT><<<grid, kBlock, 0, S(s)>>>(x, c, out, n); break;
    case GJX_MAP_ABS: k_map_f32<GJX_MAP_ABS><<<grid, kBlock, 0, S(s)>>>(x, c, out, n); break;
    default: k_map_f32<GJX_MAP_RDIV><<<grid, kBlock, 0, S(s)>>>(x, c, out, n); break;
  }
  return launch_status();
}

static int importance_launch(const gjx_plan* p, const gjx_keys* pk, int32_t n_pass, uint64_t pass_stride,
                             uint64_t row_stride, const float* const* input_cols, int n_input_cols,
                             void* const* value_cols, int n_value_cols, float* score, float* logw, uint64_t n,
                             float* max_partials, int32_t* row_e, uint64_t* row_s, const gjx_lse_out* lse,
                             gjx_stream s) {
  // logw may be null when the row sums are asked for: an estimate that needs only logsumexp(lw) writes no column at all
  if (!p || !pk || (!logw && !row_e) || n_input_cols < 0 || n_input_cols > 16 || n_value_cols < 0 ||
      n_value_cols > GJX_MAX_SITES || ((row_e == nullptr) != (row_s == nullptr)) || (lse && (!row_e || !lse->tickets)))
    return GJX_ERR_INVALID;
  for (int32_t b = 0; b < n_pass; ++b)
    if (!keys_ok(pk + b) || pk[b].has_fold) return GJX_ERR_INVALID;
  RunCols cols;
  memset(&cols, 0, sizeof(cols));
  for (int c = 0; c < n_input_cols; ++c) cols.in[c] = input_cols[c];
  for (int c = 0; c < n_value_cols; ++c) cols.out[c] = value_cols[c];
  for (int q = 0; q < p->n_sites; ++q) {
    const CSite& st = p->host[q];
    if (st.out_col >= n_value_cols) return GJX_ERR_INVALID;
    if (st.out_col >= 0 && !cols.out[st.out_col]) return GJX_ERR_INVALID;
    if (st.a0.kind == GJX_ARG_INPUT && st.a0.ref >= n_input_cols) return GJX_ERR_INVALID;
    if (st.a1.kind == GJX_ARG_INPUT && st.a1.ref >= n_input_cols) return GJX_ERR_INVALID;
    if (st.observed && st.obs.kind == GJX_ARG_INPUT && st.obs.ref >= n_input_cols) return GJX_ERR_INVALID;
  }
  if (p->max_param >= p->n_params) return GJX_ERR_INVALID;  // parameters referenced but never set
  if (p->expr_max_input >= n_input_cols) return GJX_ERR_INVALID;
  if (n == 0) return GJX_OK;
  KeySrc k = key_src(pk);
  // Specialised straight-line kernel for this site table (compiled once per plan and RNG scheme).
  if (gjx_jit::enabled()) {
    // the paired / quad forms write the 2 / 4 adjacent particles of a lane with one 8- / 16-byte store: n a multiple
    // of 2 / 4, columns (and the distance between passes) aligned alike
    uintptr_t al = (uintptr_t)logw | (uintptr_t)score | (uintptr_t)(4 * pass_stride) | (uintptr_t)(4 * n);
    for (int c = 0; c < n_value_cols; ++c) al |= (uintptr_t)value_cols[c];
    const int lane_particles = (al & 15) == 0 ? 4 : ((al & 7) == 0 ? 2 : 1);
    gjx_jit::Compiled& c = plan_compiled(const_cast<gjx_plan*>(p), pk, lane_particles);
    if (c.state != 1 && (!jit_fallback_allowed() || p->has_expr || p->scopes.n_scopes > 0)) return GJX_ERR_JIT;  // loud: never a silent 7x slower route
    if (c.state == 1) {
      uint64_t nn = n;
      LseTail tail{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0f};
      if (lse) tail = LseTail{lse->e, lse->q, lse->lse, lse->record, lse->tickets, lse->lse_shifted, lse->shift};
      PassBatch bt;
      memset(&bt, 0, sizeof bt);
      bt.n_pass = (uint32_t)n_pass;
      bt.rows_per_pass = (uint32_t)nrows_of(n);
      bt.pass_stride = pass_stride;
      bt.row_stride = row_stride;
      for (int32_t b = 0; b < n_pass; ++b) { bt.parent[b][0] = pk[b].parent[0]; bt.parent[b][1] = pk[b].parent[1]; }
      PlanParams prm = p->prm;
      PlanTables tabs = c.tabs;
      void* args[] = {&k, &cols, &score, &logw, &nn, &max_partials, &row_e, &row_s, &tail, &bt, &prm, &tabs};
      uint64_t rows = ((uint64_t)n_pass * nrows_of(n) + c.rows_per_block - 1) / c.rows_per_block;
      static const uint64_t grid_cap = [] {
        const char* e = std::getenv("GJX_IMPORTANCE_GRID");
        return e ? (uint64_t)strtoull(e, nullptr, 10) : 0ull;
      }();
      if (grid_cap && rows > grid_cap) rows = grid_cap;  // the kernel strides over rows
      if (hipModuleLaunchKernel(c.fn, (unsigned)(rows > 0x7fffffffull ? 0x7fffffffull : rows), 1, 1, (unsigned)c.block, 1, 1, 0,
                                S(s), args, nullptr) != hipSuccess)
        return GJX_ERR_LAUNCH;
      return launch_status();
    }
  }
  if (p->has_expr || p->scopes.n_scopes > 0) return GJX_ERR_UNSUPPORTED;  // programs and nested calls are compiled, never interpreted
  {
    const int rc = plan_device_table(const_cast<gjx_plan*>(p));
    if (rc) return rc;
  }
  // generic route (table interpreter): one launch per pass
  const size_t lds = sizeof(uint32_t) * (size_t)(p->n_slots > 0 ? p->n_slots : 1) * kImpTile;
  const int m = p->dist_mask;
  for (int32_t b = 0; b < n_pass; ++b) {
    KeySrc kb = key_src(pk + b);
    RunCols cb = cols;
    for (int c = 0; c < n_value_cols; ++c)
      if (cb.out[c]) cb.out[c] = (char*)cb.out[c] + 4 * (size_t)b * pass_stride;
    float* sc_b = score ? score + (size_t)b * pass_stride : nullptr;
    float* lw_b = logw + (size_t)b * pass_stride;
    float* mp_b = max_partials ? max_partials + (size_t)b * row_stride : nullptr;
    int32_t* re_b = row_e ? row_e + (size_t)b * row_stride : nullptr;
    uint64_t* rs_b = row_s ? row_s + (size_t)b * row_stride : nullptr;
#define GJX_LAUNCH_IMPORTANCE(IMPL, MASK) \
  k_importance<IMPL, MASK><<<(unsigned)((n + kImpTile - 1) / kImpTile), kBlock, lds, S(s)>>>(p->dev, p->n_sites, kb, cb, sc_b, lw_b, n, mp_b, re_b, rs_b, p->prm)
    if (pk->impl == 0) {
      if ((m & ~kMaskNormal) == 0) GJX_LAUNCH_IMPORTANCE(0, kMaskNormal);
      else if ((m & ~kMaskReal) == 0) GJX_LAUNCH_IMPORTANCE(0, kMaskReal);
      else GJX_LAUNCH_IMPORTANCE(0, kMaskAll);
    } else {
      if ((m & ~kMaskNormal) == 0) GJX_LAUNCH_IMPORTANCE(1, kMaskNormal);
      else if ((m & ~kMaskReal) == 0) GJX_LAUNCH_IMPORTANCE(1, kMaskReal);
      else GJX_LAUNCH_IMPORTANCE(1, kMaskAll);
    }
#undef GJX_LAUNCH_IMPORTANCE
  }
  // generic route: the fold is a second (one-workgroup) launch
  if (lse) k_lse_rows<<<1, kBlock, 0, S(s)>>>(row_e, row_s, nrows_of(n), 0, lse->e, lse->q, lse->lse, lse->record);
  return launch_status();
}

int gjx_importance_run(const gjx_plan* p, const gjx_keys* pk, const float* const* input_cols,
                       int n_input_cols, void* const* value_cols, int n_value_cols, float* score,
                       float* logw, uint64_t n, float* max_partials, int32_t* row_e, uint64_t* row_s,
                       const gjx_lse_out* lse, gjx_stream s) {
  return importance_launch(p, pk, 1, 0, 0, input_cols, n_input_cols, value_cols, n_value_cols, score, logw, n,
                           max_partials, row_e, row_s, lse, s);
}

int gjx_importance_estimate(const gjx_estimate_io* io, uint32_t k0, uint32_t k1, uint64_t lane, float* out, float shift,
                            gjx_stream s) {
  if (!io || !io->plan || !io->row_e || !io->row_s || !io->lse.tickets || !out || (io->impl != 0 && io->impl != 1) ||
      (io->impl == 0 && lane != 0))
    return GJX_ERR_INVALID;
  for (int q = 0; q < io->plan->n_sites; ++q)
    if (io->plan->host[q].out_col >= 0) return GJX_ERR_INVALID;  // an estimate-only plan stores no value column
  // key, sub = split(key); key, sub = split(sub): the second child twice (on the host: two cipher blocks at most)
  Key k{k0, k1, (uint32_t)lane, (uint32_t)(lane >> 32)};
  k = io->impl == 0 ? split_at<0>(split_at<0>(k, 1), 1) : split_at<1>(split_at<1>(k, 1), 1);
  gjx_keys pk;
  memset(&pk, 0, sizeof pk);
  pk.impl = io->impl;
  pk.mode = 1;  // the particle keys: split(sub, K), lazily
  pk.parent[0] = k.k0; pk.parent[1] = k.k1;
  pk.parent_lane = ((uint64_t)k.l1 << 32) | k.l0;
  gjx_lse_out lse = io->lse;
  lse.lse_shifted = out;
  lse.shift = shift;
  return importance_launch(io->plan, &pk, 1, 0, 0, io->input_cols, io->n_input_cols, nullptr, 0, nullptr, nullptr, io->n, nullptr,
                           io->row_e, io->row_s, &lse, s);
}

int gjx_importance_run_batch(const gjx_plan* p, const gjx_keys* pk, int32_t n_pass, uint64_t pass_stride,
                             uint64_t row_stride, const float* const* input_cols, int n_input_cols,
                             void* const* value_cols, int n_value_cols, float* score, float* logw, uint64_t n,
                             float* max_partials, int32_t* row_e, uint64_t* row_s, gjx_stream s) {
  if (!pk || n_pass < 1 || n_pass > kMaxPasses || pass_stride < n || row_stride < nrows_of(n)) return GJX_ERR_INVALID;
  // one launch serves lazy batches that differ only in their (lane-0) parent key
  bool fused = true;
  for (int32_t b = 0; b < n_pass && fused; ++b)
    fused = pk[b].impl == pk[0].impl && pk[b].mode == 1 && pk[b].parent_lane == 0 && pk[b].first == pk[0].first &&
            !pk[b].has_fold;
  if (fused || n_pass == 1)
    return importance_launch(p, pk, n_pass, pass_stride, row_stride, input_cols, n_input_cols, value_cols, n_value_cols,
                             score, logw, n, max_partials, row_e, row_s, nullptr, s);
  for (int32_t b = 0; b < n_pass; ++b) {  // anything else: pass by pass
    void* vc[GJX_MAX_SITES];
    for (int c = 0; c < n_value_cols; ++c) vc[c] = value_cols[c] ? (char*)value_cols[c] + 4 * (size_t)b * pass_stride : nullptr;
    const int rc = importance_launch(p, pk + b, 1, 0, 0, input_cols, n_input_cols, vc, n_value_cols,
                                     score ? score + (size_t)b * pass_stride : nullptr, logw + (size_t)b * pass_stride, n,
                                     max_partials ? max_partials + (size_t)b * row_stride : nullptr,
                                     row_e ? row_e + (size_t)b * row_stride : nullptr,
                                     row_s ? row_s + (size_t)b * row_stride : nullptr, nullptr, s);
    if (rc) return rc;
  }
  return GJX_OK;
}

// ---- weights -------------------------------------------------------------------------------------
int gjx_max_f32(const float* x, uint64_t n, const float* max_partials_in, float* out_max, void* ws,
                size_t ws_bytes, gjx_stream s) {
  if ((!x && !max_partials_in) || !out_max || n == 0) return GJX_ERR_INVALID;
  const uint64_t nt = ntiles_of(n);
  const float* partials = max_partials_in;
  uint64_t np = nrows_of(n);  // caller-provided partials are per 256-particle row
  if (!partials) {
    Carver cv{(char*)ws, ws ? ws_bytes : 0};
    float* mp = cv.take<float>(nrows_of(n));
    if (!cv.ok) return GJX_ERR_WORKSPACE;
    k_max_partials<<<grid_for(n), kBlock, 0, S(s)>>>(x, n, mp);
    partials = mp;
    np = nt;
  }
  k_reduce_max<<<1, kBlock, 0, S(s)>>>(partials, np, out_max);
  return launch_status();
}
int gjx_expsum_fix(const float* x, uint64_t n, const float* max_dev, int frac_bits_, uint64_t* out_q,
                   void* ws, size_t ws_bytes, gjx_stream s) {
  if (!x || !max_dev || !out_q || n == 0 || frac_bits_ < 1 || frac_bits_ > 40) return GJX_ERR_INVALID;
  const uint64_t nt = ntiles_of(n);
  Carver cv{(char*)ws, ws ? ws_bytes : 0};
  (void)cv.take<float>(nrows_of(n));
  uint64_t* qp = cv.take<uint64_t>(nt);
  if (!cv.ok) return GJX_ERR_WORKSPACE;
  k_expsum_partials<<<grid_for(n), kBlock, 0, S(s)>>>(x, n, max_dev, nullptr, 0, nullptr, frac_bits_, qp);
  k_reduce_sum<<<1, kBlock, 0, S(s)>>>(qp, nt, out_q, 0, max_dev, frac_bits_, nullptr, nullptr);
  return launch_status();
}
int gjx_row_stats(const float* x, uint64_t n, int32_t* row_e, uint64_t* row_s, gjx_stream s) {
  if (!x || !row_e || !row_s || n == 0) return GJX_ERR_INVALID;
  const uint64_t rows = nrows_of(n);
  k_row_stats<<<(unsigned)(rows > 0x7fffffffull ? 0x7fffffffull : rows), kBlock, 0, S(s)>>>(x, n, row_e, row_s);
  return launch_status();
}
int gjx_lse_rows(const int32_t* row_e, const uint64_t* row_s, uint64_t n_rows, int32_t* out_e,
                 uint64_t* out_q, float* out_lse, uint64_t* out_record, gjx_stream s) {
  if (!row_e || !row_s || n_rows == 0) return GJX_ERR_INVALID;
  k_lse_rows<<<1, kBlock, 0, S(s)>>>(row_e, row_s, n_rows, 0, out_e, out_q, out_lse, out_record);
  return launch_status();
}
int gjx_lse_rows_batch(const int32_t* row_e, const uint64_t* row_s, uint64_t n_rows, int32_t n_batch,
                       uint64_t batch_stride, int32_t* out_e, uint64_t* out_q, float* out_lse,
                       uint64_t* out_record, gjx_stream s) {
  if (!row_e || !row_s || n_rows == 0 || n_batch < 1 || batch_stride < n_rows) return GJX_ERR_INVALID;
  k_lse_rows<<<(unsigned)n_batch, kBlock, 0, S(s)>>>(row_e, row_s, n_rows, batch_stride, out_e, out_q, out_lse,
                                                     out_record);
  return launch_status();
}
int gjx_lse_combine(const uint64_t* records, int32_t n_records, uint64_t record_stride, int32_t n_batch,
                    uint64_t batch_stride, int32_t* out_e, uint64_t* out_q, float* out_lse,
                    uint64_t* out_record, gjx_stream s) {
  if (!records || n_records < 1 || n_batch < 1 || record_stride < GJX_LSE_RECORD_WORDS) return GJX_ERR_INVALID;
  k_lse_combine<<<(unsigned)n_batch, kWave, 0, S(s)>>>(records, n_records, record_stride, batch_stride, out_e,
                                                       out_q, out_lse, out_record);
  return launch_status();
}
int gjx_lse_finish(const float* max_dev, const uint64_t* q_dev, int frac_bits_, float* out_lse,
                   gjx_stream s) {
  if (!max_dev || !q_dev || !out_lse) return GJX_ERR_INVALID;
  k_lse_finish<<<1, 1, 0, S(s)>>>(max_dev, q_dev, frac_bits_, out_lse);
  return launch_status();
}
int gjx_logsumexp_f32(const float* x, uint64_t n, const float* max_partials_in, float* out_lse,
                      float* out_max, uint64_t* out_q, void* ws, size_t ws_bytes, gjx_stream s) {
  if (!x || n == 0) return GJX_ERR_INVALID;
  const uint64_t nt = ntiles_of(n);
  Carver cv{(char*)ws, ws ? ws_bytes : 0};
  float* mp = cv.take<float>(nrows_of(n));
  uint64_t* qp = cv.take<uint64_t>(nt);
  float* m = cv.take<float>(1);
  if (!cv.ok) return GJX_ERR_WORKSPACE;
  const int frac = frac_bits(n);
  uint64_t np = nrows_of(n);
  if (!max_partials_in) {
    k_max_partials<<<grid_for(n), kBlock, 0, S(s)>>>(x, n, mp);
    max_partials_in = mp;
    np = nt;
  }
  k_expsum_partials<<<grid_for(n), kBlock, 0, S(s)>>>(x, n, nullptr, max_partials_in, np, m, frac, qp);
  k_reduce_sum<<<1, kBlock, 0, S(s)>>>(qp, nt, out_q, 0, m, frac, out_lse, out_max);
  return launch_status();
}

// shared front half of the resampling entry points: (max, tile sums) of logw
static int weights_prepare(const float* logw, uint64_t n, Carver& cv, float** m_out,
                           uint64_t** tiles_out, hipStream_t st) {
  const uint64_t nt = ntiles_of(n);
  float* mp = cv.take<float>(nt);
  uint64_t* tiles = cv.take<uint64_t>(nt);
  float* m = cv.take<float>(1);
  if (!cv.ok) return GJX_ERR_WORKSPACE;
  k_max_partials<<<grid_for(n), kBlock, 0, st>>>(logw, n, mp);
  k_tile_sums_block<<<(unsigned)nt, kBlock, 0, st>>>(logw, n, mp, nt, frac_bits(n), tiles, m);
  *m_out = m;
  *tiles_out = tiles;
  return GJX_OK;
}

// A scalar key (categorical draw / resampling offset) is resolved on the host side of the call.
static int scalar_key(const gjx_keys* key, Key* out) {
  const Key parent{key->parent[0], key->parent[1], (uint32_t)key->parent_lane, (uint32_t)(key->parent_lane >> 32)};
  if (key->mode == 2) { *out = parent; return GJX_OK; }
  if (key->mode != 1) return GJX_ERR_UNSUPPORTED;  // device-resident scalar keys: not needed by the host API
  *out = key->impl == 0 ? split_at<0>(parent, key->first) : split_at<1>(parent, key->first);
  return GJX_OK;
}

int gjx_categorical_index(const gjx_keys* key, const float* logits, uint64_t n, int64_t* out_idx,
                          int mode, void* ws, size_t ws_bytes, gjx_stream s) {
  if (!keys_ok(key) || !logits || !out_idx || n == 0 || (mode != 0 && mode != 1)) return GJX_ERR_INVALID;
  Key k;
  int rc = scalar_key(key, &k);
  if (rc) return rc;
  const uint64_t nt = ntiles_of(n);
  Carver cv{(char*)ws, ws ? ws_bytes : 0};
  if (mode == 0) {
    if (n > 0xffffffffull) return GJX_ERR_UNSUPPORTED;
    float* pv = cv.take<float>(nt);
    int64_t* pi = cv.take<int64_t>(nt);
    if (!cv.ok) return GJX_ERR_WORKSPACE;
    GJX_DISPATCH_IMPL(key->impl, k_gumbel_partials,
                      <<<grid_for(n), kBlock, 0, S(s)>>>(k, key->has_fold, key->fold, logits, n, pv, pi));
    k_argmax_final<<<1, kBlock, 0, S(s)>>>(pv, pi, nt, out_idx);
    return launch_status();
  }
  float* m;
  uint64_t* tiles;
  rc = weights_prepare(logits, n, cv, &m, &tiles, S(s));
  if (rc) return rc;
  uint64_t* cdf = cv.take<uint64_t>(n);
  if (!cv.ok) return GJX_ERR_WORKSPACE;
  k_cdf<<<(unsigned)nt, kBlock, 0, S(s)>>>(logits, n, m, tiles, nt, frac_bits(n), cdf);
  GJX_DISPATCH_IMPL(key->impl, k_multinomial,
                    <<<1, kBlock, 0, S(s)>>>(k, key->has_fold, key->fold, cdf, n, 1, nullptr, out_idx));
  return launch_status();
}

int gjx_categorical_index_batch(const gjx_keys* keys, int32_t n_batch, const float* logits, uint64_t n, uint64_t stride,
                                int64_t* out_idx, gjx_stream s) {
  if (!keys || !logits || !out_idx || n_batch < 1 || n_batch > kMaxDrawBatch || n == 0 || n > (uint64_t)kTile || stride < n)
    return GJX_ERR_INVALID;
  DrawKeys dk;
  memset(&dk, 0, sizeof dk);
  for (int b = 0; b < n_batch; ++b) {
    if (!keys_ok(&keys[b]) || keys[b].impl != keys[0].impl) return GJX_ERR_INVALID;
    const int rc = scalar_key(&keys[b], &dk.key[b]);
    if (rc) return rc;
    dk.fold[b] = keys[b].fold;
    dk.has_fold[b] = keys[b].has_fold ? 1u : 0u;
  }
  GJX_DISPATCH_IMPL(keys[0].impl, k_gumbel_batch, <<<(unsigned)n_batch, kBlock, 0, S(s)>>>(dk, logits, n, stride, out_idx));
  return launch_status();
}

int gjx_tile_weights(const float* x, uint64_t n, uint32_t* qw, gjx_tile_rec* recs, gjx_tile_sub* subs, gjx_tile_ess* ess,
                     gjx_stream s) {
  if (!x || !qw || !recs || !subs || n == 0 || n > 0x7fffffffull) return GJX_ERR_INVALID;
  k_tile_weights<<<(unsigned)ntiles_of(n), kBlock, 0, S(s)>>>(x, n, qw, reinterpret_cast<TileRec*>(recs), reinterpret_cast<TileSub*>(subs),
                                                              reinterpret_cast<TileEss*>(ess));
  return launch_status();
}
int gjx_tile_merge(const gjx_tile_rec* recs, uint64_t n_tiles, int32_t* out_e, uint64_t* out_q, gjx_stream s) {
  if (!recs || n_tiles == 0) return GJX_ERR_INVALID;
  k_scan_records<<<1, kBlock, 0, S(s)>>>(reinterpret_cast<const TileRec*>(recs), nullptr, n_tiles, nullptr, out_e, out_q, 0);
  return launch_status();
}

// the comb offset of a resampling key: the top 53 bits of its 64-bit draw (sub-stream 0), evaluated on the host — it is
// launch-uniform, and in the kernel it cost every wave a cipher block
static double comb_offset(int impl, Key k, int has_fold, uint32_t fold) {
  if (impl == 0) return u0_from_bits(Stream<0>(k, has_fold != 0, fold).bits64(0));
  return u0_from_bits(Stream<1>(k, has_fold != 0, fold).bits64(0));
}
// test knob: GJX_SMC_SCAN_MAX=0 sends every output tile through the per-slot search (same ancestors either way)
static int scan_max_knob() {
  static const int v = [] {
    const char* e = std::getenv("GJX_SMC_SCAN_MAX");
    const int x = e ? atoi(e) : kScanMax;
    return x < 0 ? 0 : (x > 64 ? 64 : x);
  }();
  return v;
}

int gjx_resample_systematic(const gjx_keys* key, const float* logw, uint64_t n, uint64_t n_out,
                            int32_t* ancestors, int32_t* out_e, uint64_t* out_q, void* ws,
                            size_t ws_bytes, gjx_stream s) {
  if (!keys_ok(key) || !logw || !ancestors || n == 0 || n_out == 0 || n > 0x7fffffffull ||
      n_out > 0x7fffffffull)
    return GJX_ERR_INVALID;
  Key k;
  int rc = scalar_key(key, &k);
  if (rc) return rc;
  const uint64_t nt = ntiles_of(n);
  Carver cv{(char*)ws, ws ? ws_bytes : 0};
  uint32_t* qw = cv.take<uint32_t>(n);
  TileRec* recs = cv.take<TileRec>(nt);
  TileSub* subs = cv.take<TileSub>(nt);
  uint64_t* prefix = nt > (uint64_t)kMaxLdsTiles ? cv.take<uint64_t>(prefix_words(nt)) : nullptr;
  if (!cv.ok) return GJX_ERR_WORKSPACE;
  k_tile_weights<<<(unsigned)nt, kBlock, 0, S(s)>>>(logw, n, qw, recs, subs, nullptr);
  ResampleArgs A;
  A.qw = qw; A.recs = recs; A.subs = subs; A.n = n; A.ntiles = nt;
  A.n_out = n_out; A.out_lo = 0; A.out_hi = (int64_t)n_out;
  A.u0 = comb_offset(key->impl, k, key->has_fold, key->fold);
  A.e_out = out_e; A.q_out = out_q;
  if (prefix && !launch_group_records(A, prefix, S(s))) {
    k_scan_records<<<1, kBlock, 0, S(s)>>>(recs, nullptr, nt, prefix, nullptr, nullptr, 0);
    A.prefix = prefix;
  }
  A.scan_max = scan_max_knob();
  AncestorOnly P{ancestors};
  const unsigned grid = (unsigned)ntiles_of(n_out);
  if (key->impl == 0) k_resample<0, AncestorOnly><<<grid, kBlock, 0, S(s)>>>(A, P);
  else k_resample<1, AncestorOnly><<<grid, kBlock, 0, S(s)>>>(A, P);
  return launch_status();
}

int gjx_resample_multinomial(const gjx_keys* key, const float* logw, uint64_t n, uint64_t n_out,
                             int32_t* ancestors, float* out_max, uint64_t* out_q, void* ws,
                             size_t ws_bytes, gjx_stream s) {
  if (!keys_ok(key) || !logw || !ancestors || n == 0 || n_out == 0 || n > 0x7fffffffull)
    return GJX_ERR_INVALID;
  Key k;
  int rc = scalar_key(key, &k);
  if (rc) return rc;
  Carver cv{(char*)ws, ws ? ws_bytes : 0};
  float* m;
  uint64_t* tiles;
  rc = weights_prepare(logw, n, cv, &m, &tiles, S(s));
  if (rc) return rc;
  uint64_t* cdf = cv.take<uint64_t>(n);
  if (!cv.ok) return GJX_ERR_WORKSPACE;
  const uint64_t nt = ntiles_of(n);
  k_cdf<<<(unsigned)nt, kBlock, 0, S(s)>>>(logw, n, m, tiles, nt, frac_bits(n), cdf);
  GJX_DISPATCH_IMPL(key->impl, k_multinomial,
                    <<<grid_for(n_out), kBlock, 0, S(s)>>>(k, key->has_fold, key->fold, cdf, n, n_out, ancestors, nullptr));
  if (out_max) (void)hipMemcpyAsync(out_max, m, sizeof(float), hipMemcpyDeviceToDevice, S(s));
  if (out_q) (void)hipMemcpyAsync(out_q, cdf + (n - 1), sizeof(uint64_t), hipMemcpyDeviceToDevice, S(s));
  return launch_status();
}

int gjx_gather_cols(const int32_t* ancestors, uint64_t n_out, const void* const* src_cols,
                    void* const* dst_cols, int n_cols, gjx_stream s) {
  if (!ancestors || !src_cols || !dst_cols || n_cols < 0) return GJX_ERR_INVALID;
  if (n_out == 0) return GJX_OK;
  for (int c0 = 0; c0 < n_cols; c0 += 16) {
    GatherCols g;
    memset(&g, 0, sizeof(g));
    const int nc = n_cols - c0 < 16 ? n_cols - c0 : 16;
    for (int c = 0; c < nc; ++c) {
      g.src[c] = (const uint32_t*)src_cols[c0 + c];
      g.dst[c] = (uint32_t*)dst_cols[c0 + c];
      if (!g.src[c] || !g.dst[c]) return GJX_ERR_INVALID;
    }
    k_gather<<<grid_for(n_out), kBlock, 0, S(s)>>>(ancestors, n_out, g, nc);
  }
  return launch_status();
}

// ---- fused bootstrap SMC ---------------------------------------------------------------------------
static bool cfg_adaptive(const gjx_smc_config* c) { return c->ess_threshold > 0.0f && c->ess_threshold < 1.0f; }
static bool cfg_ok(const gjx_smc_config* c) {
  return c && (c->impl == 0 || c->impl == 1) && c->n_total > 0 && c->n_local > 0 &&
         c->first_slot + c->n_local <= c->n_total && c->n_steps > 0 && c->step_keys &&
         c->resample_keys && (c->first_slot % kTile) == 0 && c->n_total <= 0x7fffffffull &&
         !(c->ess_threshold < 0.0f);
}
static_assert(sizeof(gjx_tile_rec) == sizeof(TileRec) && sizeof(gjx_tile_sub) == sizeof(TileSub) && sizeof(gjx_tile_ess) == sizeof(TileEss), "gjx.h tile records");

uint64_t gjx_hmm_alias_words(int32_t n_states) {
  return n_states > 0 ? (uint64_t)n_states * (uint64_t)n_states : 0;
}
int gjx_hmm_prepare(const gjx_hmm* mdl, uint32_t* trans_cdf, float* obs_logp, gjx_stream s) {
  if (!mdl || !trans_cdf || !obs_logp || mdl->n_states <= 0 || mdl->n_states > 256 ||
      !mdl->trans_logits || !mdl->obs_logits)
    return GJX_ERR_INVALID;
  k_hmm_prepare<<<mdl->n_states, 128, 0, S(s)>>>(mdl->trans_logits, mdl->obs_logits, mdl->n_states, trans_cdf, obs_logp);
  return launch_status();
}

// What the whole-run drivers add to a step (the public per-step entry points pass the default): the filter batch
// (this step's keys and the strides of the per-filter arrays; n_filters <= 1 otherwise) and where the step's
// resampling flag goes.
struct StepCtx {
  FilterBatch fb;
  int32_t* resampled_out = nullptr;  // this step's entry of cfg->resampled_out
  const StepParams* sp = nullptr;    // a replayed run (RunGraphs): this step's entry of the device parameter block
  const float* rp = nullptr;         // ... and the run's model parameters
};

// a population a step READS (t >= 1) / WRITES: the pointers its configuration needs
static bool pop_ok(const gjx_smc_pop* p, int n_state, bool adaptive, bool reading, uint64_t nt) {
  if (!p || !p->qw || !p->recs || !p->subs) return false;
  for (int k = 0; k < n_state; ++k)
    if (!p->state[k]) return false;
  if (adaptive && (!p->logw || !p->ess)) return false;
  if (reading && nt > (uint64_t)kMaxLdsTiles && !p->prefix) return false;
  if ((((uintptr_t)p->recs | (uintptr_t)p->subs | (uintptr_t)p->ess) & 15) != 0) return false;
  return true;
}

// ---- the peer transport (gjx.h: gjx_smc_peers) ---------------------------------------------------------------------------
static bool peers_desc_ok(const gjx_smc_peers* p) {
  return p && p->world >= 2 && p->world <= GJX_MAX_PEERS && p->rank >= 0 && p->rank < p->world && p->flags && p->error &&
         p->delta[p->rank] == 0;
}
static bool peers_ok(const gjx_smc_config* c) {
  const gjx_smc_peers* p = c->peers;
  if (!peers_desc_ok(p) || c->n_filters > 1) return false;
  const uint64_t w = (uint64_t)p->world;
  return c->n_total % (w * kTile) == 0 && c->n_local == c->n_total / w && c->first_slot == (uint64_t)p->rank * c->n_local;
}
static PeerMap peer_map_of(const gjx_smc_peers* p, uint64_t n_total) {
  PeerMap pm;
  pm.world = p->world;
  pm.tiles_per_rank = n_total ? (uint32_t)(n_total / (uint64_t)p->world / kTile) : 1u;
  for (int o = 0; o < p->world; ++o) pm.delta[o] = p->delta[o];
  pm.flags = p->flags;
  pm.error = p->error;
  pm.wait_value = p->wait_value;
  pm.timeout_ticks = (uint64_t)(p->timeout_ms ? p->timeout_ms : 10000u) * 100000ull;  // s_memrealtime: 100 MHz
  pm.rank = p->rank;
  pm.sig_recs = reinterpret_cast<const TileRec*>(p->signal_recs);
  pm.sig_ess = reinterpret_cast<const TileEss*>(p->signal_ess);
  pm.sig_first = p->signal_first_tile; pm.sig_n = p->signal_n_tiles; pm.sig_value = p->signal_value;
  return pm;
}
// The arguments of a step's resample launch; for populations beyond kMaxLdsTiles the records of `prev` are merged
// into prev->prefix first (one small launch).
static int smc_resample_args(const gjx_smc_config* cfg, int t, const gjx_smc_pop* prev, const gjx_smc_pop* out,
                             int32_t* prev_e_out, uint64_t* prev_q_out, const StepCtx& ctx, gjx_stream s, ResampleArgs* Ap) {
  ResampleArgs& A = *Ap;
  const bool ad = cfg_adaptive(cfg);
  A.fb = ctx.fb;
  A.sp = ctx.sp; A.rp = ctx.rp;
  A.qw = prev->qw;
  A.lw = ad ? prev->logw : nullptr;
  A.recs = reinterpret_cast<const TileRec*>(prev->recs);
  A.subs = reinterpret_cast<const TileSub*>(prev->subs);
  A.ess = ad ? reinterpret_cast<const TileEss*>(prev->ess) : nullptr;
  A.n = cfg->n_total; A.ntiles = ntiles_of(cfg->n_total); A.n_out = cfg->n_total;
  A.out_lo = (int64_t)cfg->first_slot; A.out_hi = (int64_t)(cfg->first_slot + cfg->n_local);
  A.u0 = comb_offset(cfg->impl, Key{cfg->resample_keys[2 * t], cfg->resample_keys[2 * t + 1]}, 0, 0);
  A.e_out = prev_e_out; A.q_out = prev_q_out;
  if (ad) A.ess_thr = (double)cfg->ess_threshold * (double)cfg->n_total;
  A.resampled_out = ctx.resampled_out ? ctx.resampled_out : (cfg->resampled_out && !(cfg->n_filters > 1) ? cfg->resampled_out + t : nullptr);
  A.qw_out = out->qw; A.logw_out = out->logw;
  A.recs_out = reinterpret_cast<TileRec*>(out->recs);
  A.subs_out = reinterpret_cast<TileSub*>(out->subs);
  A.ess_out = ad ? reinterpret_cast<TileEss*>(out->ess) : nullptr;
  A.scan_max = scan_max_knob();
#ifdef GJX_PROFILE_HOOKS
  static const int dbg_stop = [] { const char* e = std::getenv("GJX_SMC_DEBUG_STOP"); return e ? atoi(e) : 0; }();
  A.debug_stop = dbg_stop;
#endif
  static const int xcd_map = [] { const char* e = std::getenv("GJX_SMC_XCD_MAP"); return e ? atoi(e) : 1; }();
  A.xcd_map = xcd_map;
  static const int wt_knob = [] { const char* e = std::getenv("GJX_SMC_WT"); return e ? atoi(e) : -1; }();
  A.wt_stores = wt_knob >= 0 ? wt_knob : (ctx.fb.n_filters > 1 ? 0 : 1);  // (measured: store16_out, gjx_device.hpp)
  // The merged prefix by ONE small launch (a workgroup per filter) instead of in every workgroup: required beyond
  // kMaxLdsTiles, and worth it from a few filters per launch (the whole-run drivers provide prev->prefix then), where its
  // ~4 us are shared by all filters while every one of the F x tiles workgroups saves the merge of its filter's records.
  if (cfg->peers) {
    if (!peers_ok(cfg) || ctx.fb.n_filters > 1) return GJX_ERR_INVALID;
    A.pm = peer_map_of(cfg->peers, cfg->n_total);
  }
  static const int wave_route = [] { const char* e = std::getenv("GJX_SMC_WAVE_ROUTE"); return e ? atoi(e) : 1; }();
  A.wave_route = wave_route;
  // (r04: filters of up to 256 tiles merge their records inside every wave — no prefix launch even for a batch of filters)
  const bool in_wave = wave_route != 0 && A.ntiles <= (uint64_t)(kWave * (kMaxLdsTiles / kBlock));
  if (A.ntiles > (uint64_t)kMaxLdsTiles || (ctx.fb.n_filters > 1 && prev->prefix && !in_wave)) {
    if (ctx.fb.n_filters > 1 && A.ntiles > (uint64_t)kMaxLdsTiles) return GJX_ERR_UNSUPPORTED;
    const unsigned nf = ctx.fb.n_filters > 1 ? ctx.fb.n_filters : 1u;
    if (nf == 1 && A.ntiles > (uint64_t)kMaxLdsTiles && launch_group_records(A, prev->prefix, S(s))) {
      A.pm.wait_value = 0;  // (peers: the group launch has waited for them; the step launch behind it need not)
      A.pm.sig_value = 0;   // (... and has delivered the deferred signal)
      return GJX_OK;        // (r04: the grouped route — no whole-population scan, no prefix array)
    }
    if (cfg->peers && A.pm.sig_value != 0) {  // (a deferred signal and no group launch to carry it: its own launch after all)
      k_peer_signal<<<(unsigned)A.pm.world, kBlock, 0, S(s)>>>(A.pm, A.pm.rank, A.pm.sig_recs, A.pm.sig_ess, A.pm.sig_first, A.pm.sig_n, A.pm.sig_value);
      A.pm.sig_value = 0;
    }
    if (nf == 1 && A.ntiles <= (uint64_t)kBigBlock * kBigPer) {
      // (peers: the records in this rank's arena are complete only once every peer has arrived — the merge launch reads them
      // before the step's own wait, so it waits itself first)
      if (A.ess) k_scan_records_big<true><<<1, kBigBlock, 0, S(s)>>>(A.recs, A.ess, A.ntiles, prev->prefix, A.pm);
      else k_scan_records_big<false><<<1, kBigBlock, 0, S(s)>>>(A.recs, A.ess, A.ntiles, prev->prefix, A.pm);
    } else {
      if (cfg->peers) k_peer_wait<<<1, kWave, 0, S(s)>>>(A.pm);
      k_scan_records<<<nf, kBlock, 0, S(s)>>>(A.recs, A.ess, A.ntiles, prev->prefix, nullptr, nullptr, 0);
    }
    A.prefix = prev->prefix;
  }
  if (cfg->peers && A.pm.sig_value != 0) {  // (a deferred signal on a route without a waiting launch in front: its own launch)
    k_peer_signal<<<(unsigned)A.pm.world, kBlock, 0, S(s)>>>(A.pm, A.pm.rank, A.pm.sig_recs, A.pm.sig_ess, A.pm.sig_first, A.pm.sig_n, A.pm.sig_value);
    A.pm.sig_value = 0;
  }
  return GJX_OK;
}
static EmitOut emit_out_of(const gjx_smc_config* cfg, const gjx_smc_pop* out) {
  return EmitOut{out->qw, out->logw, reinterpret_cast<TileRec*>(out->recs), reinterpret_cast<TileSub*>(out->subs),
                 cfg_adaptive(cfg) ? reinterpret_cast<TileEss*>(out->ess) : nullptr};
}

static int lgssm_step(const gjx_smc_config* cfg, const gjx_lgssm* mdl, int t, float y_t, const gjx_smc_pop* prev,
                      const gjx_smc_pop* out, int32_t* prev_e_out, uint64_t* prev_q_out, int32_t* ancestors_out,
                      gjx_stream s, const StepCtx& ctx) {
  if (!cfg_ok(cfg) || !mdl || t < 0 || t >= cfg->n_steps) return GJX_ERR_INVALID;
  const bool ad = cfg_adaptive(cfg);
  const uint64_t nt = ntiles_of(cfg->n_total);
  if (!pop_ok(out, 1, ad, false, nt) || (t > 0 && (!pop_ok(prev, 1, ad, true, nt) || prev->recs == out->recs))) return GJX_ERR_INVALID;
  const Key sk{cfg->step_keys[2 * t], cfg->step_keys[2 * t + 1]};
  const float rs = normal_rs(mdl->r), lognorm = normal_lognorm(mdl->r);
  const unsigned ntl = (unsigned)ntiles_of(cfg->n_local), nf = ctx.fb.n_filters > 1 ? ctx.fb.n_filters : 1u;
  if (t == 0) {
    GJX_DISPATCH_IMPL(cfg->impl, k_lgssm_init,
                      <<<ntl * nf, kBlock, 0, S(s)>>>(ctx.fb, sk, cfg->first_slot, cfg->n_local, mdl->x0_loc, mdl->x0_scale, y_t, rs, lognorm, (float*)out->state[0], ancestors_out, emit_out_of(cfg, out), ctx.sp, ctx.rp));
    return launch_status();
  }
  ResampleArgs A;
  int rc = smc_resample_args(cfg, t, prev, out, prev_e_out, prev_q_out, ctx, s, &A);
  if (rc) return rc;
  const unsigned grid = ntl * nf;
#define GJX_LAUNCH_LGSSM(I, PEERS_)                                                                                                        \
  do {                                                                                                                                     \
    LgssmPolicy<I, PEERS_> P{(const float*)prev->state[0], (float*)out->state[0], ancestors_out, sk, mdl->a, mdl->q, y_t, rs, lognorm, {}}; \
    P.wt = A.wt_stores;                                                                                                                    \
    if (ad) k_resample<I, LgssmPolicy<I, PEERS_>, true><<<grid, kBlock, 0, S(s)>>>(A, P);                                                  \
    else k_resample<I, LgssmPolicy<I, PEERS_>, false><<<grid, kBlock, 0, S(s)>>>(A, P);                                                    \
  } while (0)
  if (cfg->peers) {
    if (cfg->impl == 0) GJX_LAUNCH_LGSSM(0, true);
    else GJX_LAUNCH_LGSSM(1, true);
  } else {
    if (cfg->impl == 0) GJX_LAUNCH_LGSSM(0, false);
    else GJX_LAUNCH_LGSSM(1, false);
  }
#undef GJX_LAUNCH_LGSSM
  return launch_status();
}

static int hmm_step(const gjx_smc_config* cfg, const gjx_hmm* mdl, int t, int32_t y_t, const gjx_smc_pop* prev,
                    const gjx_smc_pop* out, int32_t* prev_e_out, uint64_t* prev_q_out, const uint32_t* trans_cdf,
                    const float* obs_logp, int32_t* ancestors_out, gjx_stream s, const StepCtx& ctx) {
  if (!cfg_ok(cfg) || !mdl || t < 0 || t >= cfg->n_steps || !trans_cdf || !obs_logp || y_t < 0 || y_t >= mdl->n_states ||
      mdl->n_states > 256 || mdl->init_state < 0 || mdl->init_state >= mdl->n_states)
    return GJX_ERR_INVALID;
  const bool ad = cfg_adaptive(cfg);
  const uint64_t nt = ntiles_of(cfg->n_total);
  if (!pop_ok(out, 1, ad, false, nt) || (t > 0 && (!pop_ok(prev, 1, ad, true, nt) || prev->recs == out->recs))) return GJX_ERR_INVALID;
  const Key sk{cfg->step_keys[2 * t], cfg->step_keys[2 * t + 1]};
  const unsigned ntl = (unsigned)ntiles_of(cfg->n_local), nf = ctx.fb.n_filters > 1 ? ctx.fb.n_filters : 1u;
  if (t == 0) {
    GJX_DISPATCH_IMPL(cfg->impl, k_hmm_init,
                      <<<ntl * nf, kBlock, 0, S(s)>>>(ctx.fb, sk, cfg->first_slot, cfg->n_local, trans_cdf, obs_logp, mdl->n_states, mdl->init_state, y_t, (int32_t*)out->state[0], ancestors_out, emit_out_of(cfg, out), ctx.sp));
    return launch_status();
  }
  ResampleArgs A;
  int rc = smc_resample_args(cfg, t, prev, out, prev_e_out, prev_q_out, ctx, s, &A);
  if (rc) return rc;
  const unsigned grid = ntl * nf;
#define GJX_LAUNCH_HMM(I, PEERS_)                                                                                                          \
  do {                                                                                                                                     \
    HmmPolicy<I, PEERS_> P{(const int32_t*)prev->state[0], (int32_t*)out->state[0], ancestors_out, sk, trans_cdf, obs_logp, mdl->n_states, y_t, {}, {}, 0.0f, nullptr}; \
    P.wt = A.wt_stores;                                                                                                                    \
    if (ad) k_resample<I, HmmPolicy<I, PEERS_>, true><<<grid, kBlock, 0, S(s)>>>(A, P);                                                    \
    else k_resample<I, HmmPolicy<I, PEERS_>, false><<<grid, kBlock, 0, S(s)>>>(A, P);                                                      \
  } while (0)
  if (cfg->peers) {
    if (cfg->impl == 0) GJX_LAUNCH_HMM(0, true);
    else GJX_LAUNCH_HMM(1, true);
  } else {
    if (cfg->impl == 0) GJX_LAUNCH_HMM(0, false);
    else GJX_LAUNCH_HMM(1, false);
  }
#undef GJX_LAUNCH_HMM
  return launch_status();
}

int gjx_smc_lgssm_step(const gjx_smc_config* cfg, const gjx_lgssm* mdl, int t, float y_t, const gjx_smc_pop* prev,
                       const gjx_smc_pop* out, int32_t* prev_e_out, uint64_t* prev_q_out, int32_t* ancestors_out,
                       gjx_stream s) {
  if (cfg && cfg->n_filters > 1) return GJX_ERR_INVALID;
  return lgssm_step(cfg, mdl, t, y_t, prev, out, prev_e_out, prev_q_out, ancestors_out, s, StepCtx{});
}
int gjx_smc_hmm_step(const gjx_smc_config* cfg, const gjx_hmm* mdl, int t, int32_t y_t, const gjx_smc_pop* prev,
                     const gjx_smc_pop* out, int32_t* prev_e_out, uint64_t* prev_q_out, const uint32_t* trans_cdf,
                     const float* obs_logp, int32_t* ancestors_out, gjx_stream s) {
  if (cfg && cfg->n_filters > 1) return GJX_ERR_INVALID;
  return hmm_step(cfg, mdl, t, y_t, prev, out, prev_e_out, prev_q_out, trans_cdf, obs_logp, ancestors_out, s, StepCtx{});
}

int gjx_smc_finish(const gjx_smc_config* cfg, const gjx_tile_rec* recs, int32_t* e_out, uint64_t* q_out, gjx_stream s) {
  if (!cfg_ok(cfg) || !recs || cfg->n_filters > 1) return GJX_ERR_INVALID;
  k_scan_records<<<1, kBlock, 0, S(s)>>>(reinterpret_cast<const TileRec*>(recs), nullptr, ntiles_of(cfg->n_total), nullptr, e_out, q_out, 0);
  return launch_status();
}
// One message per rank and step for ESS-adaptive filters on a collective transport: [records of the rank's tiles | their ESS
// sums].  pack: this rank's block -> its slot; unpack: every other rank's slot -> the dense arrays.
__global__ __launch_bounds__(kBlock) void k_records_pack(TileRec* recs, TileEss* ess, uint4* stage, uint64_t tl, int world, int rank,
                                                         int unpack) {
  const uint64_t per = 2 * tl;  // 16-byte words per slot
  for (uint64_t i = blockIdx.x * (uint64_t)kBlock + threadIdx.x; i < per * (uint64_t)world; i += (uint64_t)gridDim.x * kBlock) {
    const int r = (int)(i / per);
    const uint64_t w = i - (uint64_t)r * per;
    if ((r == rank) == (unpack != 0)) continue;  // pack: own slot only; unpack: the others
    uint4* arr = w < tl ? reinterpret_cast<uint4*>(recs) + (uint64_t)r * tl + w : reinterpret_cast<uint4*>(ess) + (uint64_t)r * tl + (w - tl);
    if (unpack) *arr = stage[i];
    else stage[i] = *arr;
  }
}
int gjx_smc_records_pack(const gjx_smc_config* cfg, int world, int unpack, gjx_tile_rec* recs, gjx_tile_ess* ess, void* stage,
                         gjx_stream s) {
  if (!cfg_ok(cfg) || !recs || !ess || !stage || world < 1 || world > 64 || cfg->n_local == 0 || cfg->n_local % kTile ||
      cfg->n_total != cfg->n_local * (uint64_t)world || cfg->first_slot % cfg->n_local)
    return GJX_ERR_INVALID;
  const uint64_t tl = cfg->n_local / kTile;
  const uint64_t words = 2 * tl * (uint64_t)world;
  const unsigned grid = (unsigned)((words + kBlock - 1) / kBlock < 64 ? (words + kBlock - 1) / kBlock : 64);
  k_records_pack<<<grid, kBlock, 0, S(s)>>>(reinterpret_cast<TileRec*>(recs), reinterpret_cast<TileEss*>(ess), reinterpret_cast<uint4*>(stage), tl,
                                            world, (int)(cfg->first_slot / cfg->n_local), unpack);
  return launch_status();
}
int gjx_smc_peer_signal(const gjx_smc_peers* peers, const gjx_tile_rec* recs, const gjx_tile_ess* ess, uint64_t first_tile,
                        uint64_t n_tiles, uint64_t value, gjx_stream s) {
  if (!peers_desc_ok(peers) || (n_tiles > 0 && !recs) || ((((uintptr_t)recs | (uintptr_t)ess) & 15) != 0)) return GJX_ERR_INVALID;
  k_peer_signal<<<(unsigned)peers->world, kBlock, 0, S(s)>>>(peer_map_of(peers, 0), peers->rank, reinterpret_cast<const TileRec*>(recs),
                                                             reinterpret_cast<const TileEss*>(ess), first_tile, n_tiles, value);
  return launch_status();
}
int gjx_smc_peer_signal_fused(const gjx_smc_config* cfg) {
  // (the group-record route: one filter beyond kMaxLdsTiles tiles, within kMaxGroups groups, not switched off)
  if (!cfg || !cfg->peers || cfg->n_filters > 1) return 0;
  static const bool allow = [] { const char* e = std::getenv("GJX_SMC_BIG_ROUTE"); return !(e && e[0] == 'p'); }();
  const uint64_t nt = ntiles_of(cfg->n_total);
  return allow && nt > (uint64_t)kMaxLdsTiles && (nt + kGroupTiles - 1) / kGroupTiles <= (uint64_t)kMaxGroups ? 1 : 0;
}
int gjx_smc_peer_wait(const gjx_smc_peers* peers, uint64_t value, gjx_stream s) {
  if (!peers_desc_ok(peers)) return GJX_ERR_INVALID;
  PeerMap pm = peer_map_of(peers, 0);
  pm.wait_value = value;
  k_peer_wait<<<1, kWave, 0, S(s)>>>(pm);
  return launch_status();
}
int gjx_smc_source_ranges(const gjx_smc_config* cfg, const gjx_tile_rec* recs, const gjx_tile_ess* ess, int world, int64_t ticket,
                          int64_t* out_ranges, gjx_stream s) {
  if (!cfg_ok(cfg) || !recs || !out_ranges || world < 1 || world > kMaxRangeBlocks || cfg->n_total % (uint64_t)world)
    return GJX_ERR_INVALID;
  if (cfg_adaptive(cfg) && !ess) return GJX_ERR_INVALID;
  k_source_ranges<<<1, kBlock, 0, S(s)>>>(reinterpret_cast<const TileRec*>(recs), reinterpret_cast<const TileEss*>(ess),
                                          cfg_adaptive(cfg) ? (double)cfg->ess_threshold * (double)cfg->n_total : 0.0,
                                          ntiles_of(cfg->n_total), cfg->n_total, world, ticket, out_ranges);
  return launch_status();
}

// What the whole-run drivers (fixed models and plans) share: the ping-pong populations carved from the caller's
// workspace — the LAST step lands in the caller's state / log-weight buffers — and the per-step context (this step's
// keys of every filter, where the step's resampling flag goes).
struct RunCommon {
  unsigned F = 1;
  uint64_t nt = 0, stride = 0;
  bool adaptive = false;
  gjx_smc_pop pop[2];
  float* logw_final = nullptr;  // the caller's log-weight buffer (written by the last step)
  int last = 0;
  FilterBatch fb;
};
static int run_common_init(const gjx_smc_config* cfg, Carver& cv, RunCommon& rc, int n_state, void* const* state_out,
                           float* logw_out, gjx_stream s, bool clear_flags = true) {
  const uint64_t N = cfg->n_total;
  rc.nt = ntiles_of(N);
  rc.F = cfg->n_filters > 1 ? (unsigned)cfg->n_filters : 1u;
  rc.stride = rc.F > 1 ? cfg->filter_stride : N;
  rc.adaptive = cfg_adaptive(cfg);
  if (rc.F > kMaxFilters || (rc.F > 1 && (rc.stride != rc.nt * kTile || rc.nt > (uint64_t)kMaxLdsTiles))) return GJX_ERR_UNSUPPORTED;
  if (rc.adaptive && !cfg->resampled_out) return GJX_ERR_INVALID;
  const int T = cfg->n_steps;
  rc.last = (T - 1) & 1;
  memset(rc.pop, 0, sizeof rc.pop);
  const size_t cells = (size_t)rc.F * rc.stride;
  for (int k = 0; k < n_state; ++k) {
    rc.pop[rc.last].state[k] = state_out[k];
    rc.pop[rc.last ^ 1].state[k] = cv.take<uint32_t>(cells);
  }
  for (int i = 0; i < 2; ++i) {
    rc.pop[i].qw = cv.take<uint32_t>(cells);
    rc.pop[i].recs = reinterpret_cast<gjx_tile_rec*>(cv.take<TileRec>((size_t)rc.F * rc.nt));
    rc.pop[i].subs = reinterpret_cast<gjx_tile_sub*>(cv.take<TileSub>((size_t)rc.F * rc.nt));
    rc.pop[i].ess = rc.adaptive ? reinterpret_cast<gjx_tile_ess*>(cv.take<TileEss>((size_t)rc.F * rc.nt)) : nullptr;
    static const unsigned prefix_filters = [] {
      const char* e = std::getenv("GJX_SMC_PREFIX_FILTERS");  // tuning knob: filters per launch from which the merge is a launch of its own
      return e ? (unsigned)atoi(e) : 4u;
    }();
    rc.pop[i].prefix = (rc.nt > (uint64_t)kMaxLdsTiles || rc.F >= prefix_filters) ? cv.take<uint64_t>((size_t)rc.F * prefix_words(rc.nt)) : nullptr;
  }
  // log-weights: an adaptive filter carries them from step to step; otherwise only the last step's are stored
  rc.logw_final = logw_out;
  if (rc.adaptive) {
    rc.pop[rc.last].logw = logw_out;
    rc.pop[rc.last ^ 1].logw = cv.take<float>(cells);
  }
  if (!cv.ok) return GJX_ERR_WORKSPACE;
  if (rc.F > 1) {
    rc.fb.n_filters = rc.F; rc.fb.tiles = (uint32_t)rc.nt; rc.fb.stride = rc.stride; rc.fb.mq_stride = (uint64_t)T;
  }
  const size_t nmq = (size_t)rc.F * (size_t)T;
  if (clear_flags && cfg->resampled_out && hipMemsetAsync(cfg->resampled_out, 0, nmq * sizeof(int32_t), S(s)) != hipSuccess) return GJX_ERR_LAUNCH;
  return GJX_OK;
}
// the populations of step t: written (`out`) and read (`prev`), and the step's context
static StepCtx run_step_ctx(const gjx_smc_config* cfg, RunCommon& rc, int t, gjx_smc_pop* out) {
  const int T = cfg->n_steps;
  for (unsigned f = 0; f < rc.F && rc.F > 1; ++f) {  // this step's keys of every filter ([F, T, 2] host arrays)
    const uint32_t* sk = cfg->step_keys + 2 * ((size_t)f * T + t);
    const uint32_t* rk = cfg->resample_keys + 2 * ((size_t)f * T + t);
    rc.fb.step_key[f] = Key{sk[0], sk[1]};
    rc.fb.u0[f] = comb_offset(cfg->impl, Key{rk[0], rk[1]}, 0, 0);
  }
  *out = rc.pop[t & 1];
  if (!rc.adaptive) out->logw = t == T - 1 ? rc.logw_final : nullptr;
  StepCtx ctx;
  ctx.fb = rc.fb;
  ctx.resampled_out = cfg->resampled_out ? cfg->resampled_out + t : nullptr;
  return ctx;
}
// the closing (e, Q) pairs of a run: the merge of the last step's records, one workgroup per filter
static int run_finish(const gjx_smc_config* cfg, RunCommon& rc, int32_t* out_e, uint64_t* out_q, gjx_stream s) {
  const int T = cfg->n_steps;
  k_scan_records<<<rc.F, kBlock, 0, S(s)>>>(reinterpret_cast<const TileRec*>(rc.pop[rc.last].recs), nullptr, rc.nt, nullptr,
                                            out_e + (T - 1), out_q + (T - 1), (uint64_t)T);
  return launch_status();
}

// ---- importance over a Scan model: the T-step walk of every particle in one launch ------------------
struct gjx_scan_plan {
  int n_state, n_obs, n_step;
  uint32_t flags;
  CSite step[GJX_MAX_SITES];
  CArg next_state[GJX_SMC_MAX_STATE];
  gjx_jit::Compiled jit[3];  // THREEFRY, PHILOX one particle per lane, PHILOX four per lane (GenScan::quad)
  gjx_jit::ScopeInfo scopes;  // nested calls inside the step kernel (gjx_scan_plan_create_scoped)
  std::vector<void*> dev_owned;  // per-row tables of categorical sites
  std::mutex mu;
  ExprStore step_expr;  // GJX_ARG_EXPR programs
  StateExprStore next_state_expr;
};
int gjx_scan_plan_create(const gjx_scan_model* m, uint32_t flags, gjx_scan_plan** out) {
  if (!m || !out || (flags & ~(uint32_t)GJX_PLAN_FAST_MATH) || m->n_state < 1 || m->n_state > GJX_SMC_MAX_STATE ||
      m->n_obs < 0 || m->n_obs > GJX_SMC_MAX_OBS || !m->step_sites || m->n_step_sites <= 0 || m->n_step_sites > GJX_MAX_SITES)
    return GJX_ERR_INVALID;
  gjx_scan_plan* p = new (std::nothrow) gjx_scan_plan;
  if (!p) return GJX_ERR_LAUNCH;
  p->n_state = m->n_state; p->n_obs = m->n_obs; p->n_step = m->n_step_sites; p->flags = flags;
  bool ok = true;
  for (int s = 0; ok && s < p->n_step; ++s) ok = convert_site(m->step_sites[s], s, p->step[s], m->n_state, m->n_obs, true);
  for (int k = 0; ok && k < p->n_state; ++k) {
    ok = arg_ok(m->next_state[k], p->n_step, m->n_state, m->n_obs, true) && m->next_state[k].kind != GJX_ARG_TABLE;
    p->next_state[k] = carg(m->next_state[k]);
  }
  if (!ok) {
    delete p;
    return GJX_ERR_INVALID;
  }
  (void)expr_adopt(p->step, p->n_step, &p->step_expr);
  state_expr_adopt(p->next_state, p->n_state, &p->next_state_expr);
  *out = p;
  return GJX_OK;
}
int gjx_scan_plan_create_scoped(const gjx_scan_model* m, const gjx_scope* scopes, int n_scopes, uint32_t flags,
                                gjx_scan_plan** out) {
  gjx_scan_plan* p = nullptr;
  const int rc = gjx_scan_plan_create(m, flags, &p);
  if (rc) return rc;
  if (!gjx_jit::derive_scopes(m->step_sites, m->n_step_sites, scopes, n_scopes, p->scopes)) {
    gjx_scan_plan_destroy(p);
    return GJX_ERR_INVALID;
  }
  *out = p;
  return GJX_OK;
}
int gjx_scan_plan_destroy(gjx_scan_plan* p) {
  if (!p) return GJX_OK;
  for (auto& c : p->jit) gjx_jit::release(&c);
  free_owned(p->dev_owned);
  delete p;
  return GJX_OK;
}
static std::string scan_plan_source(const gjx_scan_plan* plan, int impl, const char** kname = nullptr, PlanTables* tabs = nullptr,
                                    bool quad = false, int* block = nullptr) {
  gjx_jit::TableScope ts;
  gjx_jit::GenScan<CSite, CArg> g;
  g.impl = impl; g.sites = plan->step; g.n_sites = plan->n_step; g.next_state = plan->next_state;
  g.n_state = plan->n_state; g.n_obs = plan->n_obs; g.fast_math = (plan->flags & GJX_PLAN_FAST_MATH) != 0;
  g.quad = quad;
  g.sc = plan->scopes.n_scopes > 0 ? &plan->scopes : nullptr;
  if (kname) *kname = g.kname();
  std::string src = g.run();
  if (tabs) *tabs = ts.reg.tables();
  if (block) *block = g.block;
  return src;
}
int gjx_scan_plan_compile_check(const gjx_scan_plan* p, int impl) {
  if (!p || (impl != 0 && impl != 1)) return GJX_ERR_INVALID;
  if (std::getenv("GJX_PLAN_JIT_DUMP")) fprintf(stderr, "%s\n", scan_plan_source(p, impl).c_str());
  if (impl == 1 && !gjx_jit::compile_only(scan_plan_source(p, impl, nullptr, nullptr, true))) return GJX_ERR_UNSUPPORTED;
  return gjx_jit::compile_only(scan_plan_source(p, impl)) ? GJX_OK : GJX_ERR_UNSUPPORTED;
}
int gjx_scan_run(gjx_scan_plan* p, const gjx_scan_io* io, gjx_stream s) {
  if (!p || !io || !keys_ok(io->particle_keys) || io->particle_keys->has_fold || !io->logw || io->n_steps < 1 ||
      io->col_stride < io->n || (p->n_obs > 0 && !io->obs) || !io->carry0 || io->n_value_cols < 0 ||
      io->n_value_cols > GJX_MAX_SITES || ((io->row_e == nullptr) != (io->row_s == nullptr)) ||
      (io->lse && (!io->row_e || !io->lse->tickets)))
    return GJX_ERR_INVALID;
  /* PHILOX step keys put t + 1 above bit 40 of the lane */
  if (io->particle_keys->impl == 1 &&
      (io->n_steps >= (1 << 24) - 1 || (io->particle_keys->mode == 1 && io->particle_keys->first + io->n >= (1ull << 40))))
    return GJX_ERR_INVALID;
  RunCols cols;
  memset(&cols, 0, sizeof(cols));
  for (int c = 0; c < io->n_value_cols; ++c) cols.out[c] = io->value_cols[c];
  for (int q = 0; q < p->n_step; ++q) {
    const CSite& st = p->step[q];
    if (st.out_col >= io->n_value_cols || (st.out_col >= 0 && !cols.out[st.out_col])) return GJX_ERR_INVALID;
  }
  if (io->n == 0) return GJX_OK;
  const int impl = io->particle_keys->impl;
  // Scan plans exist only as specialised kernels: a failed compilation is an error, never a slower route.
  if (!gjx_jit::enabled()) return GJX_ERR_UNSUPPORTED;
  // PHILOX children of a lane-0 key from an even first index, n and every column a multiple of four elements: the quad form
  // (four adjacent particles per lane share two pair blocks and two Box-Muller transforms per site and step)
  bool quad = false;
  {
    const gjx_keys* pk = io->particle_keys;
    uintptr_t al = (uintptr_t)io->logw | (uintptr_t)io->score | (uintptr_t)(4 * io->col_stride) | (uintptr_t)(4 * io->n);
    for (int c = 0; c < io->n_value_cols; ++c) al |= (uintptr_t)io->value_cols[c];
    for (int d = 0; d < p->n_state; ++d) al |= io->carry_out ? (uintptr_t)io->carry_out[d] : 0;
    static const bool allow = [] { const char* e = std::getenv("GJX_SCAN_QUAD"); return !(e && e[0] == '0'); }();
    quad = allow && impl == 1 && pk->mode == 1 && pk->parent_lane == 0 && (pk->first & 1) == 0 && (al & 15) == 0;
  }
  gjx_jit::Compiled& c = p->jit[quad ? 2 : impl];
  if (c.state == 0) {
    std::lock_guard<std::mutex> lock(p->mu);
    if (c.state == 0) {
      cat_tables_prepare(p->step, p->n_step, &p->dev_owned);
      const char* kname = nullptr;
      int block = 256;
      const std::string src = scan_plan_source(p, impl, &kname, &c.tabs, quad, &block);
      c.block = block;
      if (std::getenv("GJX_PLAN_JIT_DUMP")) fprintf(stderr, "%s\n", src.c_str());
      c.state = gjx_jit::compile(src, impl, &c, kname) ? 1 : -1;
    }
  }
  if (c.state != 1) return GJX_ERR_JIT;
  KeySrc k = key_src(io->particle_keys);
  ScanArgs sa;
  memset(&sa, 0, sizeof sa);
  sa.obs = io->obs; sa.n = io->n; sa.col_stride = io->col_stride; sa.n_steps = io->n_steps;
  for (int d = 0; d < p->n_state; ++d) {
    sa.carry0[d] = io->carry0[d];
    sa.carry0_cols[d] = io->carry0_cols ? io->carry0_cols[d] : nullptr;
    sa.carry_out[d] = io->carry_out ? io->carry_out[d] : nullptr;
  }
  LseTail tail{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0f};
  if (io->lse) tail = LseTail{io->lse->e, io->lse->q, io->lse->lse, io->lse->record, io->lse->tickets, io->lse->lse_shifted, io->lse->shift};
  float* score = io->score; float* logw = io->logw; float* mp = io->max_partials;
  int32_t* row_e = io->row_e; uint64_t* row_s = io->row_s;
  PlanTables tabs = c.tabs;
  void* args[] = {&k, &cols, &sa, &score, &logw, &mp, &row_e, &row_s, &tail, &tabs};
  const uint64_t rows = nrows_of(io->n);
  if (hipModuleLaunchKernel(c.fn, (unsigned)(rows > 0x7fffffffull ? 0x7fffffffull : rows), 1, 1, (unsigned)c.block, 1, 1, 0, S(s), args,
                            nullptr) != hipSuccess)
    return GJX_ERR_LAUNCH;
  return launch_status();
}

// ---- bootstrap SMC for a user model: generated policy in the fused resample kernel -----------------
struct gjx_smc_plan {
  int n_state, n_obs, n_init, n_step;
  CSite init[GJX_MAX_SITES], step[GJX_MAX_SITES];
  CArg init_state[GJX_SMC_MAX_STATE], next_state[GJX_SMC_MAX_STATE];
  gjx_jit::CompiledSmc jit[2];
  gjx_jit::CompiledSmc jit_peers[2];  // r04: the step kernels of the peer transport (compiled on first use)
  std::vector<void*> dev_owned;  // per-row tables of categorical sites
  std::mutex mu;
  ExprStore init_expr, step_expr;  // GJX_ARG_EXPR programs of the two tables
  StateExprStore init_state_expr, next_state_expr;
  gjx_jit::ScopeInfo init_scopes, step_scopes;        // nested calls (gjx_smc_plan_create_scoped): generated kernels only
  bool has_expr = false;                              // any program?  Then the filter runs as generated kernels only
  CSite* dev_init = nullptr; CSite* dev_step = nullptr;  // the interpreter's device copies of the tables (made on first use)
};

int gjx_smc_plan_create(const gjx_smc_model* m, gjx_smc_plan** out) {
  if (!m || !out || m->n_state < 1 || m->n_state > GJX_SMC_MAX_STATE || m->n_obs < 0 || m->n_obs > GJX_SMC_MAX_OBS ||
      !m->init_sites || !m->step_sites || m->n_init_sites <= 0 || m->n_init_sites > GJX_MAX_SITES ||
      m->n_step_sites <= 0 || m->n_step_sites > GJX_MAX_SITES)
    return GJX_ERR_INVALID;
  gjx_smc_plan* p = new (std::nothrow) gjx_smc_plan;
  if (!p) return GJX_ERR_LAUNCH;
  p->n_state = m->n_state; p->n_obs = m->n_obs; p->n_init = m->n_init_sites; p->n_step = m->n_step_sites;
  bool ok = true;
  for (int s = 0; ok && s < p->n_init; ++s) ok = convert_site(m->init_sites[s], s, p->init[s], m->n_state, m->n_obs, false);
  for (int s = 0; ok && s < p->n_step; ++s) ok = convert_site(m->step_sites[s], s, p->step[s], m->n_state, m->n_obs, true);
  for (int k = 0; ok && k < p->n_state; ++k) {
    ok = arg_ok(m->init_state[k], p->n_init, m->n_state, m->n_obs, false) && m->init_state[k].kind != GJX_ARG_TABLE &&
         arg_ok(m->next_state[k], p->n_step, m->n_state, m->n_obs, true) && m->next_state[k].kind != GJX_ARG_TABLE;
    p->init_state[k] = carg(m->init_state[k]);
    p->next_state[k] = carg(m->next_state[k]);
  }
  if (!ok) {
    delete p;
    return GJX_ERR_INVALID;
  }
  p->has_expr = expr_adopt(p->init, p->n_init, &p->init_expr);
  p->has_expr = expr_adopt(p->step, p->n_step, &p->step_expr) || p->has_expr;
  state_expr_adopt(p->init_state, p->n_state, &p->init_state_expr);
  state_expr_adopt(p->next_state, p->n_state, &p->next_state_expr);
  for (int k = 0; k < p->n_state; ++k)
    p->has_expr = p->has_expr || p->init_state[k].kind == GJX_ARG_EXPR || p->next_state[k].kind == GJX_ARG_EXPR;
  *out = p;
  return GJX_OK;
}
int gjx_smc_plan_create_scoped(const gjx_smc_model* m, const gjx_scope* init_scopes, int n_init_scopes,
                               const gjx_scope* step_scopes, int n_step_scopes, gjx_smc_plan** out) {
  gjx_smc_plan* p = nullptr;
  const int rc = gjx_smc_plan_create(m, &p);
  if (rc) return rc;
  if (!gjx_jit::derive_scopes(m->init_sites, m->n_init_sites, init_scopes, n_init_scopes, p->init_scopes) ||
      !gjx_jit::derive_scopes(m->step_sites, m->n_step_sites, step_scopes, n_step_scopes, p->step_scopes)) {
    gjx_smc_plan_destroy(p);
    return GJX_ERR_INVALID;
  }
  *out = p;
  return GJX_OK;
}
int gjx_smc_plan_destroy(gjx_smc_plan* p) {
  if (!p) return GJX_OK;
  for (auto& c : p->jit) gjx_jit::release_smc(&c);  // compiled modules are owned by the process-wide (bounded) cache
  for (auto& c : p->jit_peers) gjx_jit::release_smc(&c);
  free_owned(p->dev_owned);
  if (p->dev_init) (void)hipFree(p->dev_init);
  if (p->dev_step) (void)hipFree(p->dev_step);
  delete p;
  return GJX_OK;
}
int gjx_jit_compile_source(const char* source) {
  if (!source) return GJX_ERR_INVALID;
  return gjx_jit::compile_only(source) ? GJX_OK : GJX_ERR_JIT;
}
int gjx_jit_stats(uint64_t* compiles, uint64_t* cached_modules, uint64_t* evictions) {
  gjx_jit::ModuleCache& mc = gjx_jit::ModuleCache::get();
  std::lock_guard<std::mutex> lock(mc.mu);
  if (compiles) *compiles = mc.compiles;
  if (cached_modules) *cached_modules = mc.map.size();
  if (evictions) *evictions = mc.evictions;
  return GJX_OK;
}

int gjx_jit_routes(uint64_t* child_compiles, uint64_t* inproc_compiles, uint64_t* child_failures, uint64_t* spawn_failures) {
  gjx_jit::RouteCounters& r = gjx_jit::routes();
  if (child_compiles) *child_compiles = r.child.load();
  if (inproc_compiles) *inproc_compiles = r.inproc.load();
  if (child_failures) *child_failures = r.child_failures.load();
  if (spawn_failures) *spawn_failures = r.spawn_failures.load();
  return GJX_OK;
}

static std::string smc_plan_source(const gjx_smc_plan* plan, int impl, PlanTables* tabs = nullptr, bool peers = false) {
  gjx_jit::TableScope ts;
  gjx_jit::GenSmc<CSite, CArg> g;
  g.peers = peers;
  g.impl = impl; g.init_sites = plan->init; g.n_init = plan->n_init; g.step_sites = plan->step;
  g.n_step = plan->n_step; g.init_state = plan->init_state; g.next_state = plan->next_state; g.n_state = plan->n_state;
  g.sc_init = plan->init_scopes.n_scopes > 0 ? &plan->init_scopes : nullptr;
  g.sc_step = plan->step_scopes.n_scopes > 0 ? &plan->step_scopes : nullptr;
  std::string src = g.run();
  if (tabs) *tabs = ts.reg.tables();
  return src;
}
int gjx_smc_plan_compile_check(const gjx_smc_plan* p, int impl) {
  if (!p || (impl != 0 && impl != 1)) return GJX_ERR_INVALID;
  if (std::getenv("GJX_PLAN_JIT_DUMP")) fprintf(stderr, "%s\n", smc_plan_source(p, impl).c_str());
  return gjx_jit::compile_only(smc_plan_source(p, impl)) ? GJX_OK : GJX_ERR_UNSUPPORTED;
}

// The interpreter's device copies of the site tables (GJX_PLAN_JIT=0 / a failed compilation with the fallback allowed).
static int smc_plan_interp_tables(gjx_smc_plan* plan) {
  std::lock_guard<std::mutex> lock(plan->mu);
  if (plan->dev_init && plan->dev_step) return GJX_OK;
  CSite* di = nullptr; CSite* ds = nullptr;
  if (hipMalloc((void**)&di, sizeof(CSite) * (size_t)plan->n_init) != hipSuccess) return GJX_ERR_LAUNCH;
  if (hipMalloc((void**)&ds, sizeof(CSite) * (size_t)plan->n_step) != hipSuccess) { (void)hipFree(di); return GJX_ERR_LAUNCH; }
  if (hipMemcpy(di, plan->init, sizeof(CSite) * (size_t)plan->n_init, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(ds, plan->step, sizeof(CSite) * (size_t)plan->n_step, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(di); (void)hipFree(ds);
    return GJX_ERR_LAUNCH;
  }
  plan->dev_init = di; plan->dev_step = ds;
  return GJX_OK;
}
static gjx_jit::CompiledSmc* smc_plan_compiled(gjx_smc_plan* plan, int impl, bool peers = false) {
  if (!gjx_jit::enabled()) return nullptr;  // (GJX_PLAN_JIT=0: the table-walking policy, smc_plan_route)
  gjx_jit::CompiledSmc& c = peers ? plan->jit_peers[impl] : plan->jit[impl];
  if (c.state == 0) {
    std::lock_guard<std::mutex> lock(plan->mu);
    if (c.state == 0) {
      if (plan->jit[impl].state == 0 && plan->jit_peers[impl].state == 0) {  // (the derived tables: once per plan)
        cat_tables_prepare(plan->init, plan->n_init, &plan->dev_owned);
        cat_tables_prepare(plan->step, plan->n_step, &plan->dev_owned);
      }
      c.state = gjx_jit::compile_smc(smc_plan_source(plan, impl, &c.tabs, peers), &c) ? 1 : -1;
    }
  }
  return c.state == 1 ? &c : nullptr;
}

// One step of a plan-driven filter: the generated init kernel (t == 0) or the generated policy inside the fused
// resample kernel, for the output slots [first_slot, first_slot + n_local) of cfg.
// Which route a generated filter takes: its compiled kernels; or, with the compiler switched off (GJX_PLAN_JIT=0) or failed
// and GJX_PLAN_JIT_FALLBACK=1, the table-walking policy (k_smc_interp_*: same bits, several times slower) — unless the
// model holds programs (compiled, never interpreted).  *c_out = nullptr means the interpreter.
static int smc_plan_route(gjx_smc_plan* plan, int impl, gjx_jit::CompiledSmc** c_out, bool peers = false) {
  *c_out = smc_plan_compiled(plan, impl, peers);
  if (*c_out) return GJX_OK;
  if (peers) return gjx_jit::enabled() ? GJX_ERR_JIT : GJX_ERR_UNSUPPORTED;  // (the table-walking policy has no peer form)
  const bool off = !gjx_jit::enabled();
  if (!off && !jit_fallback_allowed()) return GJX_ERR_JIT;  // loud: never a silent slower route
  if (plan->has_expr || plan->init_scopes.n_scopes > 0 || plan->step_scopes.n_scopes > 0) return off ? GJX_ERR_UNSUPPORTED : GJX_ERR_JIT;
  return smc_plan_interp_tables(plan);
}
static int smc_plan_step(const gjx_smc_config* cfg, gjx_smc_plan* plan, gjx_jit::CompiledSmc* cp, int t,
                         const float* obs_t, const gjx_smc_pop* prev, const gjx_smc_pop* out, int32_t* prev_e_out,
                         uint64_t* prev_q_out, int32_t* ancestors_out, gjx_stream s, const StepCtx& ctx) {
  const bool ad = cfg_adaptive(cfg);
  const uint64_t nt = ntiles_of(cfg->n_total);
  if (!pop_ok(out, plan->n_state, ad, false, nt) ||
      (t > 0 && (!pop_ok(prev, plan->n_state, ad, true, nt) || prev->recs == out->recs)))
    return GJX_ERR_INVALID;
  const unsigned ntl = (unsigned)ntiles_of(cfg->n_local), nf = ctx.fb.n_filters > 1 ? ctx.fb.n_filters : 1u;
  PlanPolicyArgs PA;
  memset(&PA, 0, sizeof(PA));
  for (int k = 0; k < plan->n_state; ++k) {
    PA.prev_state[k] = t > 0 ? (const float*)prev->state[k] : nullptr;
    PA.state_out[k] = (float*)out->state[k];
  }
  PA.anc_out = ancestors_out;
  PA.step_key = Key{cfg->step_keys[2 * t], cfg->step_keys[2 * t + 1]};
  for (int k = 0; k < plan->n_obs; ++k) PA.obs[k] = obs_t[k];
  InterpTable IT;
  if (!cp) {  // the table-walking policy
    memset(&IT, 0, sizeof IT);
    IT.sites = t == 0 ? plan->dev_init : plan->dev_step;
    IT.n_sites = t == 0 ? plan->n_init : plan->n_step;
    IT.n_state = plan->n_state;
    for (int k = 0; k < plan->n_state; ++k) IT.state_args[k] = t == 0 ? plan->init_state[k] : plan->next_state[k];
  }
  if (t == 0) {
    uint64_t first = cfg->first_slot, nl = cfg->n_local;
    FilterBatch fb = ctx.fb;
    EmitOut em = emit_out_of(cfg, out);
    if (!cp) {
      if (cfg->impl == 0) k_smc_interp_init<0><<<ntl * nf, kBlock, 0, S(s)>>>(PA, first, nl, em, fb, IT);
      else k_smc_interp_init<1><<<ntl * nf, kBlock, 0, S(s)>>>(PA, first, nl, em, fb, IT);
      return launch_status();
    }
    PlanTables tabs = cp->tabs;
    void* args[] = {&PA, &first, &nl, &em, &fb, &tabs};
    if (hipModuleLaunchKernel(cp->init, ntl * nf, 1, 1, kBlock, 1, 1, 0, S(s), args, nullptr) != hipSuccess) return GJX_ERR_LAUNCH;
    return launch_status();
  }
  ResampleArgs A;
  int rc = smc_resample_args(cfg, t, prev, out, prev_e_out, prev_q_out, ctx, s, &A);
  if (rc) return rc;
  PA.wt = A.wt_stores;
  if (!cp) {
    if (cfg->impl == 0) k_smc_interp_step<0><<<ntl * nf, kBlock, 0, S(s)>>>(A, PA, IT);
    else k_smc_interp_step<1><<<ntl * nf, kBlock, 0, S(s)>>>(A, PA, IT);
    return launch_status();
  }
  PlanTables tabs = cp->tabs;
  void* args[] = {&A, &PA, &tabs};
  if (hipModuleLaunchKernel(ad ? cp->step_adaptive : cp->step, ntl * nf, 1, 1, kBlock, 1, 1, 0, S(s), args, nullptr) != hipSuccess) return GJX_ERR_LAUNCH;
  return launch_status();
}

int gjx_smc_plan_step(const gjx_smc_config* cfg, gjx_smc_plan* plan, int t, const float* obs_t, const gjx_smc_pop* prev,
                      const gjx_smc_pop* out, int32_t* prev_e_out, uint64_t* prev_q_out, int32_t* ancestors_out,
                      gjx_stream s) {
  if (!cfg_ok(cfg) || !plan || t < 0 || t >= cfg->n_steps || !out || (t > 0 && !prev) || (plan->n_obs > 0 && !obs_t) ||
      cfg->n_filters > 1)
    return GJX_ERR_INVALID;
  gjx_jit::CompiledSmc* c = nullptr;
  // (the peer transport: steps t >= 1 run the peer form of the generated kernels; step 0 reads no source population)
  const int route = smc_plan_route(plan, cfg->impl, &c, cfg->peers != nullptr && t > 0);
  if (route) return route;
  return smc_plan_step(cfg, plan, c, t, obs_t, prev, out, prev_e_out, prev_q_out, ancestors_out, s, StepCtx{});
}

int gjx_smc_run_plan(const gjx_smc_config* cfg, gjx_smc_plan* plan, const float* obs_host, int32_t* out_e,
                     uint64_t* out_q, float* const* state_out, float* logw_out, int32_t* ancestors_out,
                     void* ws, size_t ws_bytes, gjx_stream s) {
  if (!cfg_ok(cfg) || cfg->first_slot != 0 || cfg->n_local != cfg->n_total || !plan || !out_e || !out_q ||
      !state_out || !logw_out || (plan->n_obs > 0 && !obs_host))
    return GJX_ERR_INVALID;
  gjx_jit::CompiledSmc* cp = nullptr;
  const int route = smc_plan_route(plan, cfg->impl, &cp);
  if (route) return route;
  const int D = plan->n_state, T = cfg->n_steps;
  void* st[GJX_SMC_MAX_STATE];
  for (int k = 0; k < D; ++k) {
    if (!state_out[k]) return GJX_ERR_INVALID;
    st[k] = state_out[k];
  }
  // several filters per launch (as in smc_run): filter f's particles lie f * stride further in every array
  Carver cv{(char*)ws, ws ? ws_bytes : 0};
  RunCommon rc;
  int r = run_common_init(cfg, cv, rc, D, st, logw_out, s);
  if (r) return r;
  for (int t = 0; t < T; ++t) {
    gjx_smc_pop out;
    StepCtx ctx = run_step_ctx(cfg, rc, t, &out);
    r = smc_plan_step(cfg, plan, cp, t, plan->n_obs ? obs_host + (size_t)t * (size_t)plan->n_obs : nullptr,
                      &rc.pop[(t & 1) ^ 1], &out, t ? out_e + (t - 1) : nullptr, t ? out_q + (t - 1) : nullptr,
                      ancestors_out ? ancestors_out + (size_t)t * rc.F * rc.stride : nullptr, s, ctx);
    if (r) return r;
  }
  return run_finish(cfg, rc, out_e, out_q, s);
}

}  // extern "C"

// ---- r04: a whole run as ONE hipGraph, replayed -------------------------------------------------------------------------------
// A one-filter run is T dependent launches of ~11 us: the ~0.6 us the command processor spends between two launches of a
// stream are 5 % of it, and a graph's launches follow each other closer (tools/graph_smc.py: 11.5 -> 10.8 us per step).  What
// differs between two runs over the same buffers — keys, observations, comb offsets, the model's scalars — is read from a
// device block (StepParams), so ONE instantiated graph serves every later run of the shape: a run copies its parameters (a few
// KB) and launches the graph.  The first run of a shape takes the plain stream of launches; the second captures (about a
// millisecond, once); GJX_SMC_GRAPH=0 switches the whole thing off.  The CAPTURE runs on a stream of the entry's own (the
// legacy default stream cannot be captured); the graph is LAUNCHED on the caller's stream — launched on a second stream and
// ordered by events it ran a microsecond per step slower than the plain launches (measured: profiles/r04_ab/README.md).
// Results: the same kernels with the same values.
struct RunGraphInfo {
  int kind = 0;              // 0 LGSSM, 1 HMM
  uint64_t extra[4] = {};    // model facts that are baked into the launches
  float rp[8] = {};          // the model's scalar parameters (device copy: ResampleArgs::rp)
  int n_rp = 0;
  const void* y_host = nullptr;  // T observations (f32 or i32: the bits travel)
};
struct RunGraphEntry {
  uint64_t key[20] = {};
  hipGraphExec_t exec = nullptr;
  hipStream_t stream = nullptr;
  void* dev = nullptr;
  void* pin[2] = {nullptr, nullptr};       // pinned staging of the parameter block, used alternately
  hipEvent_t pin_ev[2] = {nullptr, nullptr};  // ... recorded behind the copy that read it
  int flip = 0;
  size_t bytes = 0;
  int seen = 0;
  bool bad = false;
  uint64_t stamp = 0;
  void destroy() {
    // (eviction only — the 17th shape of a process: a replay of this graph may still be running on some stream)
    if (exec) (void)hipDeviceSynchronize();
    if (exec) (void)hipGraphExecDestroy(exec);
    for (int i = 0; i < 2; ++i) {
      if (pin_ev[i]) (void)hipEventDestroy(pin_ev[i]);
      if (pin[i]) (void)hipHostFree(pin[i]);
    }
    if (stream) (void)hipStreamDestroy(stream);
    if (dev) (void)hipFree(dev);
    *this = RunGraphEntry{};
  }
};
struct RunGraphs {
  std::mutex mu;
  std::vector<RunGraphEntry> entries;
  uint64_t clock = 0;
  uint64_t replays = 0, captures = 0;
  static RunGraphs& get() {
    static RunGraphs* g = new RunGraphs;  // (never destroyed: HIP objects must not be torn down after the runtime)
    return *g;
  }
  static bool enabled() {
    static const bool on = [] { const char* e = std::getenv("GJX_SMC_GRAPH"); return !(e && e[0] == '0'); }();
    return on;
  }
};

template <class Step>
static int smc_run(const gjx_smc_config* cfg, const void* model, int32_t* out_e, uint64_t* out_q,
                   void* state_out, float* logw_out, int32_t* ancestors_out, void* ws,
                   size_t ws_bytes, gjx_stream s, Step step, const RunGraphInfo* gi = nullptr) {
  if (!cfg_ok(cfg) || cfg->first_slot != 0 || cfg->n_local != cfg->n_total || !model || !out_e ||
      !out_q || !state_out || !logw_out)
    return GJX_ERR_INVALID;
  const int T = cfg->n_steps;
  // the T launches of the run on stream `st`; sp / rp: the device parameter block of a replayed run (or null)
  auto run_loop = [&](gjx_stream st, const StepParams* sp, const float* rp) -> int {
    Carver cv{(char*)ws, ws ? ws_bytes : 0};
    RunCommon rc;
    void* stt[1] = {state_out};
    int r = run_common_init(cfg, cv, rc, 1, stt, logw_out, st, sp == nullptr);  // (a replayed run clears the flags in front of the graph)
    for (int t = 0; t < T && !r; ++t) {
      int32_t* anc_t = ancestors_out ? ancestors_out + (size_t)t * rc.F * rc.stride : nullptr;
      gjx_smc_pop out;
      StepCtx ctx = run_step_ctx(cfg, rc, t, &out);
      if (sp) { ctx.sp = sp + t; ctx.rp = rp; }
#ifdef GJX_PROFILE_HOOKS
      static const bool dbg_fixed = std::getenv("GJX_SMC_DEBUG_FIXED") != nullptr;  // profiling: every step reads step 0's population
      if (dbg_fixed && t > 0) {
        out = rc.pop[1];
        if (!rc.adaptive) out.logw = nullptr;
        r = step(t, &rc.pop[0], &out, out_e + (t - 1), out_q + (t - 1), anc_t, ctx, st);
        continue;
      }
#endif
      r = step(t, &rc.pop[(t & 1) ^ 1], &out, t ? out_e + (t - 1) : nullptr, t ? out_q + (t - 1) : nullptr, anc_t, ctx, st);
    }
    if (r) return r;
    return run_finish(cfg, rc, out_e, out_q, st);
  };
#ifndef GJX_PROFILE_HOOKS
  if (gi && RunGraphs::enabled() && cfg->n_filters <= 1 && !cfg->peers && T >= 2) {
    RunGraphs& G = RunGraphs::get();
    std::lock_guard<std::mutex> lock(G.mu);
    uint64_t key[20] = {(uint64_t)gi->kind, cfg->n_total, (uint64_t)T, (uint64_t)cfg->impl, (uint64_t)f2u(cfg->ess_threshold),
                        (uint64_t)(uintptr_t)cfg->resampled_out, (uint64_t)(uintptr_t)out_e, (uint64_t)(uintptr_t)out_q,
                        (uint64_t)(uintptr_t)state_out, (uint64_t)(uintptr_t)logw_out, (uint64_t)(uintptr_t)ancestors_out,
                        (uint64_t)(uintptr_t)ws, (uint64_t)ws_bytes, gi->extra[0], gi->extra[1], gi->extra[2], gi->extra[3], 0, 0, 0};
    RunGraphEntry* e = nullptr;
    for (auto& x : G.entries)
      if (memcmp(x.key, key, sizeof key) == 0) { e = &x; break; }
    if (!e) {
      if (G.entries.size() >= 16) {  // (evict the entry used longest ago)
        size_t old = 0;
        for (size_t i = 1; i < G.entries.size(); ++i)
          if (G.entries[i].stamp < G.entries[old].stamp) old = i;
        G.entries[old].destroy();
        G.entries.erase(G.entries.begin() + (long)old);
      }
      G.entries.emplace_back();
      e = &G.entries.back();
      memcpy(e->key, key, sizeof key);
    }
    e->stamp = ++G.clock;
    e->seen++;
    if (!e->bad && (e->exec || e->seen >= 2)) {
      // this run's parameters
      const size_t n_rp = 8;
      const size_t bytes = sizeof(StepParams) * ((size_t)T + 1) + sizeof(float) * n_rp;  // (one spare entry: the last step's prefetch)
      bool ok = true;
      if (!e->stream) {
        ok = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) == hipSuccess && hipMalloc(&e->dev, bytes) == hipSuccess;
        for (int i = 0; ok && i < 2; ++i)
          ok = hipHostMalloc(&e->pin[i], bytes, hipHostMallocDefault) == hipSuccess &&
               hipEventCreateWithFlags(&e->pin_ev[i], hipEventDisableTiming) == hipSuccess &&
               hipEventRecord(e->pin_ev[i], e->stream) == hipSuccess;
        e->bytes = bytes;
      }
      // (the staging buffer of two runs ago: its copy has long run — the wait returns at once in steady state)
      const int fl = e->flip;
      e->flip ^= 1;
      if (ok) ok = hipEventSynchronize(e->pin_ev[fl]) == hipSuccess;
      if (!ok) { (void)hipGetLastError(); e->bad = true; return run_loop(s, nullptr, nullptr); }
      char* host = reinterpret_cast<char*>(e->pin[fl]);
      StepParams* hp = reinterpret_cast<StepParams*>(host);
      for (int t = 0; t < T; ++t) {
        hp[t].k0 = cfg->step_keys[2 * t]; hp[t].k1 = cfg->step_keys[2 * t + 1];
        hp[t].y_bits = reinterpret_cast<const uint32_t*>(gi->y_host)[t];
        hp[t].pad = 0;
        hp[t].u0 = t ? comb_offset(cfg->impl, Key{cfg->resample_keys[2 * t], cfg->resample_keys[2 * t + 1]}, 0, 0) : 0.0;
      }
      memset(&hp[T], 0, sizeof(StepParams));
      memcpy(host + sizeof(StepParams) * ((size_t)T + 1), gi->rp, sizeof(float) * n_rp);
      const StepParams* dsp = reinterpret_cast<const StepParams*>(e->dev);
      const float* drp = reinterpret_cast<const float*>(reinterpret_cast<const char*>(e->dev) + sizeof(StepParams) * ((size_t)T + 1));
      if (ok && !e->exec) {  // capture the run's launches once
        hipGraph_t graph = nullptr;
        ok = hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        int r = ok ? run_loop(reinterpret_cast<gjx_stream>(e->stream), dsp, drp) : GJX_ERR_LAUNCH;
        if (ok) ok = hipStreamEndCapture(e->stream, &graph) == hipSuccess && r == GJX_OK && graph;
        if (ok) ok = hipGraphInstantiate(&e->exec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (graph) (void)hipGraphDestroy(graph);
        if (ok) G.captures++;
      }
      if (ok && e->exec) {
        // the graph is LAUNCHED on the caller's stream (only its capture needed a stream of its own): the parameter copy, the
        // flags' clearing and the graph are stream-ordered there like the launches they replace; nothing waits on the host
        ok = hipMemcpyAsync(e->dev, host, bytes, hipMemcpyHostToDevice, S(s)) == hipSuccess &&
             hipEventRecord(e->pin_ev[fl], S(s)) == hipSuccess &&
             (!cfg->resampled_out || hipMemsetAsync(cfg->resampled_out, 0, (size_t)T * sizeof(int32_t), S(s)) == hipSuccess);
        if (ok) ok = hipGraphLaunch(e->exec, S(s)) == hipSuccess;
        if (ok) {
          G.replays++;
          return GJX_OK;
        }
      }
      (void)hipGetLastError();
      e->bad = true;  // (anything that went wrong: this shape takes the plain stream of launches from now on)
    }
  }
#endif
  return run_loop(s, nullptr, nullptr);
}

extern "C" {

int gjx_smc_run_graph_stats(uint64_t* captures, uint64_t* replays) {
  RunGraphs& G = RunGraphs::get();
  std::lock_guard<std::mutex> lock(G.mu);
  if (captures) *captures = G.captures;
  if (replays) *replays = G.replays;
  return GJX_OK;
}

int gjx_smc_run_lgssm(const gjx_smc_config* cfg, const gjx_lgssm* model, const float* y_host,
                      int32_t* out_e, uint64_t* out_q, float* state_out, float* logw_out,
                      int32_t* ancestors_out, void* ws, size_t ws_bytes, gjx_stream s) {
  if (!y_host) return GJX_ERR_INVALID;
  auto step = [&](int t, const gjx_smc_pop* prev, const gjx_smc_pop* out, int32_t* pe, uint64_t* pq, int32_t* anc,
                  const StepCtx& ctx, gjx_stream st) { return lgssm_step(cfg, model, t, y_host[t], prev, out, pe, pq, anc, st, ctx); };
  RunGraphInfo gi;
  gi.kind = 0;
  gi.y_host = y_host;
  if (model) {
    const float rp[6] = {model->x0_loc, model->x0_scale, model->a, model->q, normal_rs(model->r), normal_lognorm(model->r)};
    memcpy(gi.rp, rp, sizeof rp);
    gi.n_rp = 6;
  }
  return smc_run(cfg, model, out_e, out_q, state_out, logw_out, ancestors_out, ws, ws_bytes, s, step, model ? &gi : nullptr);
}

int gjx_smc_run_hmm(const gjx_smc_config* cfg, const gjx_hmm* model, const int32_t* y_host,
                    int32_t* out_e, uint64_t* out_q, int32_t* state_out, float* logw_out,
                    int32_t* ancestors_out, void* ws, size_t ws_bytes, gjx_stream s) {
  if (!y_host || !model || model->n_states <= 0 || model->n_states > 256) return GJX_ERR_INVALID;
  // tables live at the tail of the workspace
  const size_t kk = (size_t)model->n_states * (size_t)model->n_states;
  const size_t cdf_bytes = pad256((size_t)gjx_hmm_alias_words(model->n_states) * 4);
  const size_t tail = cdf_bytes + pad256(kk * 4);
  if (!ws || ws_bytes < tail) return GJX_ERR_WORKSPACE;
  // aligned DOWN from the end of the workspace: with an unaligned `ws` within 255 bytes of `tail` that lands below
  // the workspace — checked as addresses, before anything is carved
  const uintptr_t w0 = (uintptr_t)ws, tail_a = (w0 + ws_bytes - tail) & ~(uintptr_t)255;
  if (tail_a < w0) return GJX_ERR_WORKSPACE;
  char* tail_p = (char*)tail_a;
  uint32_t* tcdf = (uint32_t*)tail_p;
  float* ologp = (float*)(tail_p + cdf_bytes);
  if ((char*)ologp + kk * 4 > (char*)ws + ws_bytes) return GJX_ERR_WORKSPACE;
  int rc = gjx_hmm_prepare(model, tcdf, ologp, s);
  if (rc) return rc;
  const size_t head_bytes = (size_t)(tail_p - (char*)ws);
  auto step = [&](int t, const gjx_smc_pop* prev, const gjx_smc_pop* out, int32_t* pe, uint64_t* pq, int32_t* anc,
                  const StepCtx& ctx, gjx_stream st) { return hmm_step(cfg, model, t, y_host[t], prev, out, pe, pq, tcdf, ologp, anc, st, ctx); };
  // (a replayed run does not pass through hmm_step's per-step checks: every observation is checked here first; the tables
  // were prepared above on the caller's stream, in front of the graph)
  RunGraphInfo gi;
  gi.kind = 1;
  gi.y_host = y_host;
  gi.extra[0] = (uint64_t)model->n_states; gi.extra[1] = (uint64_t)(uint32_t)model->init_state;
  bool ys_ok = cfg && cfg->n_steps > 0;
  for (int t = 0; ys_ok && t < cfg->n_steps; ++t) ys_ok = y_host[t] >= 0 && y_host[t] < model->n_states;
  return smc_run(cfg, model, out_e, out_q, state_out, logw_out, ancestors_out, ws, head_bytes, s, step, ys_ok ? &gi : nullptr);
}

}  // extern "C"

// =====================================================================================================================
// Multi-GPU: the native communicator and the sharded filter driver (gjx.h "multi-GPU"; driver in gjx_sharded.hpp)
// =====================================================================================================================
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enums only: the library itself is loaded on first use (dlopen)

#include <atomic>

#include "gjx_sharded.hpp"

namespace {

// device memory primitives of the virtual-rank transport
struct HipMem {
  void* buf = nullptr;
  size_t cap = 0;
  ~HipMem() {
    if (buf) (void)hipFree(buf);
  }
  int copy(void* dst, const void* src, size_t bytes, gjx_stream s) {
    if (!bytes) return GJX_OK;
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, S(s)) == hipSuccess ? GJX_OK : GJX_ERR_LAUNCH;
  }
  int sync(gjx_stream s) { return hipStreamSynchronize(S(s)) == hipSuccess ? GJX_OK : GJX_ERR_LAUNCH; }
};

// RCCL, loaded on first use (a process that never shards does not need the library)
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  bool ok = false;
  static Rccl& get() {
    static Rccl r = [] {
      Rccl x;
      // an RCCL already in the process (torch's) first: two copies of the library must not talk to each other
      const char* names[] = {"librccl.so", "librccl.so.1"};
      for (const char* nm : names)
        if (!x.lib) x.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
      for (const char* nm : names)
        if (!x.lib) x.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (!x.lib) {
        fprintf(stderr, "[gjx] RCCL not found (dlopen librccl.so): %s\n", dlerror());
        return x;
      }
#define GJX_NCCL_SYM(field, name) x.field = reinterpret_cast<decltype(x.field)>(dlsym(x.lib, name))
      GJX_NCCL_SYM(GetUniqueId, "ncclGetUniqueId");
      GJX_NCCL_SYM(CommInitRank, "ncclCommInitRank");
      GJX_NCCL_SYM(CommDestroy, "ncclCommDestroy");
      GJX_NCCL_SYM(CommAbort, "ncclCommAbort");
      GJX_NCCL_SYM(AllReduce, "ncclAllReduce");
      GJX_NCCL_SYM(AllGather, "ncclAllGather");
      GJX_NCCL_SYM(Send, "ncclSend");
      GJX_NCCL_SYM(Recv, "ncclRecv");
      GJX_NCCL_SYM(GroupStart, "ncclGroupStart");
      GJX_NCCL_SYM(GroupEnd, "ncclGroupEnd");
#undef GJX_NCCL_SYM
      x.ok = x.GetUniqueId && x.CommInitRank && x.CommDestroy && x.AllReduce && x.AllGather && x.Send && x.Recv &&
             x.GroupStart && x.GroupEnd;
      if (!x.ok) fprintf(stderr, "[gjx] RCCL library lacks a required symbol\n");
      return x;
    }();
    return r;
  }
};
static_assert(sizeof(ncclUniqueId) == GJX_COMM_ID_BYTES, "gjx.h GJX_COMM_ID_BYTES");

struct RcclTransport : gjx_sharded::Transport {
  ncclComm_t comm = nullptr;
  ~RcclTransport() override {
    if (comm) (void)Rccl::get().CommDestroy(comm);
  }
  // a rank that failed mid-run: abort the communicator so that peers blocked in a collective return with an error
  // instead of waiting for this rank forever
  void abort() override {
    Rccl& R = Rccl::get();
    if (comm && R.CommAbort) {
      (void)R.CommAbort(comm);
      comm = nullptr;
    }
  }
  static int st(ncclResult_t r) {
    if (r == ncclSuccess) return GJX_OK;
    fprintf(stderr, "[gjx] RCCL call failed (ncclResult %d)\n", (int)r);
    return GJX_ERR_LAUNCH;
  }
  int allgather(void* full, size_t bytes, gjx_stream s) override {  // in place: the send buffer is this rank's block
    return st(Rccl::get().AllGather((const char*)full + (size_t)rank * bytes, full, bytes, ncclUint8, comm, S(s)));
  }
  int exchange(void* const* cols, const size_t* elems, const size_t* units, int n_cols, const gjx_sharded::Seg* sends, int ns,
               const gjx_sharded::Seg* recvs, int nr, gjx_stream s) override {
    if (!ns && !nr) return GJX_OK;
    Rccl& R = Rccl::get();
    using gjx_sharded::seg_off;
    ncclResult_t r = R.GroupStart();
    for (int i = 0; i < ns && r == ncclSuccess; ++i)
      for (int c = 0; c < n_cols && r == ncclSuccess; ++c) {
        const size_t a = seg_off(sends[i].a, elems[c], units[c]), b = seg_off(sends[i].b, elems[c], units[c]);
        r = R.Send((const char*)cols[c] + a, b - a, ncclUint8, sends[i].peer, comm, S(s));
      }
    for (int i = 0; i < nr && r == ncclSuccess; ++i)
      for (int c = 0; c < n_cols && r == ncclSuccess; ++c) {
        const size_t a = seg_off(recvs[i].a, elems[c], units[c]), b = seg_off(recvs[i].b, elems[c], units[c]);
        r = R.Recv((char*)cols[c] + a, b - a, ncclUint8, recvs[i].peer, comm, S(s));
      }
    const ncclResult_t e = R.GroupEnd();
    return st(r != ncclSuccess ? r : e);
  }
  int stream_sync(gjx_stream s) override { return hipStreamSynchronize(S(s)) == hipSuccess ? GJX_OK : GJX_ERR_LAUNCH; }
};

int dev_copy(void* dst, const void* src, size_t bytes, gjx_stream s) {
  return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, S(s)) == hipSuccess ? GJX_OK : GJX_ERR_LAUNCH;
}

}  // namespace

extern "C" {

int gjx_comm_unique_id(void* id_out) {
  if (!id_out) return GJX_ERR_INVALID;
  Rccl& R = Rccl::get();
  if (!R.ok) return GJX_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (R.GetUniqueId(&id) != ncclSuccess) return GJX_ERR_LAUNCH;
  memcpy(id_out, &id, sizeof id);
  return GJX_OK;
}
int gjx_comm_init_rccl(const void* id, int rank, int world, gjx_comm** out) {
  if (!id || !out || world < 1 || world > 64 || rank < 0 || rank >= world) return GJX_ERR_INVALID;
  Rccl& R = Rccl::get();
  if (!R.ok) return GJX_ERR_UNSUPPORTED;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  RcclTransport* t = new (std::nothrow) RcclTransport;
  if (!t) return GJX_ERR_LAUNCH;
  t->rank = rank;
  t->world = world;
  if (R.CommInitRank(&t->comm, world, uid, rank) != ncclSuccess) {
    t->comm = nullptr;
    delete t;
    return GJX_ERR_LAUNCH;
  }
  gjx_comm* c = new (std::nothrow) gjx_comm;
  if (!c) {
    delete t;
    return GJX_ERR_LAUNCH;
  }
  c->t = t;
  *out = c;
  return GJX_OK;
}
int gjx_comm_group_create(int world, gjx_comm_group** out) {
  if (!out || world < 1 || world > 16) return GJX_ERR_INVALID;
  *out = new (std::nothrow) gjx_comm_group(world);
  return *out ? GJX_OK : GJX_ERR_LAUNCH;
}
int gjx_comm_group_destroy(gjx_comm_group* g) {
  delete g;
  return GJX_OK;
}
int gjx_comm_init_local(gjx_comm_group* g, int rank, gjx_comm** out) {
  if (!g || !out || rank < 0 || rank >= g->g.world) return GJX_ERR_INVALID;
  gjx_comm* c = new (std::nothrow) gjx_comm;
  if (!c) return GJX_ERR_LAUNCH;
  c->t = new (std::nothrow) gjx_sharded::LocalTransport<HipMem>(&g->g, rank);
  if (!c->t) {
    delete c;
    return GJX_ERR_LAUNCH;
  }
  *out = c;
  return GJX_OK;
}
int gjx_comm_init_callbacks(int rank, int world, gjx_allgather_fn allgather, gjx_exchange_fn exchange,
                            gjx_stream_sync_fn stream_sync, void* user, gjx_comm** out) {
  if (!out) return GJX_ERR_INVALID;
  gjx_comm* c = new (std::nothrow) gjx_comm;
  if (!c) return GJX_ERR_LAUNCH;
  const int rc = gjx_sharded::comm_init_callbacks(rank, world, allgather, exchange, stream_sync, user, &c->t);
  if (rc) {
    delete c;
    return rc;
  }
  *out = c;
  return GJX_OK;
}
int gjx_comm_init_peers(const gjx_smc_peers* peers, gjx_comm_group* group, int wait_launch, gjx_comm** out) {
  if (!out) return GJX_ERR_INVALID;
  gjx_comm* c = new (std::nothrow) gjx_comm;
  if (!c) return GJX_ERR_LAUNCH;
  // (virtual ranks of one process share a device stream: their launches execute in enqueue order)
  const int rc = gjx_sharded::comm_init_peers(peers, group ? &group->g : nullptr, wait_launch, group != nullptr, &c->t);
  if (rc) {
    delete c;
    return rc;
  }
  *out = c;
  return GJX_OK;
}
int gjx_comm_destroy(gjx_comm* c) {
  delete c;
  return GJX_OK;
}
int gjx_comm_rank(const gjx_comm* c) { return c && c->t ? c->t->rank : -1; }
int gjx_comm_world(const gjx_comm* c) { return c && c->t ? c->t->world : -1; }
int gjx_comm_lse_combine(gjx_comm* c, const uint64_t* records, int32_t n_batch, uint64_t* gathered, int32_t* out_e,
                         uint64_t* out_q, float* out_lse, gjx_stream s) {
  if (!c || !c->t) return GJX_ERR_INVALID;
  return gjx_sharded::lse_combine(*c->t, records, n_batch, gathered, out_e, out_q, out_lse, s, dev_copy);
}
int gjx_smc_sharded_run_lgssm(gjx_comm* c, const gjx_smc_config* cfg, const gjx_lgssm* model, const float* y_host,
                              const gjx_sharded_io* io, gjx_stream s) {
  if (!c || !c->t) return GJX_ERR_INVALID;
  return gjx_sharded::run_lgssm(*c->t, cfg, model, y_host, io, s);
}
int gjx_smc_sharded_run_hmm(gjx_comm* c, const gjx_smc_config* cfg, const gjx_hmm* model, const int32_t* y_host,
                            const uint32_t* trans_alias, const float* obs_logp, const gjx_sharded_io* io, gjx_stream s) {
  if (!c || !c->t) return GJX_ERR_INVALID;
  return gjx_sharded::run_hmm(*c->t, cfg, model, y_host, trans_alias, obs_logp, io, s);
}
int gjx_smc_sharded_run_plan(gjx_comm* c, const gjx_smc_config* cfg, gjx_smc_plan* plan, const float* obs_host,
                             const gjx_sharded_io* io, gjx_stream s) {
  if (!c || !c->t || !plan) return GJX_ERR_INVALID;
  return gjx_sharded::run_plan(*c->t, cfg, plan, plan->n_state, plan->n_obs, obs_host, io, s);
}

}  // extern "C"
