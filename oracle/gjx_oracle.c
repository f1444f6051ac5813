/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see gjx_oracle_math.h for the pinning statement).
 *
 * CPU implementation of include/gjx.h with HOST pointers; `gjx_stream` is ignored and every call
 * is synchronous.  Each function is a straight-line, sequential-semantics restatement of the
 * reference path (citations relative to /root/reference/src/genjax/_src):
 *   importance walk ........ generative_functions/static.py:340-399 (GenerateHandler),
 *                            distributions/distribution.py:117-147, 371-396
 *   ImportanceK batching ... inference/smc.py:298-315
 *   log-marginal / draw .... inference/smc.py:96-109
 *   Scan T-loop ............ generative_functions/combinators/scan.py:237-294
 * OpenMP only parallelises loops whose iterations are independent or whose reduction is an
 * exact integer sum / max, so results do not depend on the thread count.
 */
#define _POSIX_C_SOURCE 200809L /* clock_gettime, sched_yield (the peer transport's bounded wait) */
#include "../include/gjx.h"
#include "gjx_oracle_math.h"

#include <stdlib.h>
#include <time.h>
#include <sched.h>

#define O_TILE 1024u
#define O_CAT_FRAC 23

int gjx_version(int* major, int* minor) {
  if (major) *major = GJX_VERSION_MAJOR;
  if (minor) *minor = GJX_VERSION_MINOR;
  return GJX_OK;
}
const char* gjx_backend_name(void) { return "oracle-cpu"; }
int gjx_frac_bits(uint64_t n_total) { return o_frac_bits(n_total); }
uint64_t gjx_smc_tile(void) { return O_TILE; }
uint64_t gjx_num_tiles(uint64_t n) { return (n + O_TILE - 1) / O_TILE; }
#define O_ROW 256u
uint64_t gjx_num_max_partials(uint64_t n) { return (n + O_ROW - 1) / O_ROW; }
size_t gjx_workspace_bytes(int op, uint64_t n) { (void)op; (void)n; return 64; }

/* ---- keys ---------------------------------------------------------------------------------- */
static inline int key_words(int impl) { return GJX_KEY_WORDS(impl); }
static inline void parent_key(const gjx_keys* k, uint32_t out[4]) {
  out[0] = k->parent[0]; out[1] = k->parent[1];
  out[2] = (uint32_t)k->parent_lane; out[3] = (uint32_t)(k->parent_lane >> 32);
}
static inline void key_at(const gjx_keys* k, uint64_t i, uint32_t out[4]) {
  if (k->mode == 0) {
    const int w = key_words(k->impl);
    out[2] = 0u; out[3] = 0u;
    for (int c = 0; c < w; ++c) out[c] = k->keys[(uint64_t)w * i + c];
  } else {
    uint32_t parent[4];
    parent_key(k, parent);
    if (k->mode == 1) o_split_at(k->impl, parent, k->first + i, out);
    else o_key_copy(out, parent);
  }
}
static inline void key_store(int impl, uint32_t* out, uint64_t i, const uint32_t key[4]) {
  const int w = key_words(impl);
  for (int c = 0; c < w; ++c) out[(uint64_t)w * i + c] = key[c];
}
static inline o_stream stream_at(const gjx_keys* k, uint64_t i) {
  uint32_t key[4];
  key_at(k, i, key);
  return o_stream_make(k->impl, key, k->has_fold, k->fold);
}
static int keys_ok(const gjx_keys* k) {
  if (!k) return 0;
  if (k->impl != 0 && k->impl != 1) return 0;
  if (k->mode == 0) return k->keys != NULL;
  if (k->impl == 0 && k->parent_lane != 0) return 0;
  return k->mode == 1 || k->mode == 2;
}

int gjx_rng_keys(const gjx_keys* k, uint64_t n, uint32_t* out, gjx_stream s) {
  (void)s;
  if (!keys_ok(k) || (!out && n)) return GJX_ERR_INVALID;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    uint32_t key[4];
    key_at(k, (uint64_t)i, key);
    if (k->has_fold) { uint32_t f[4]; o_fold_in(k->impl, key, k->fold, f); o_key_copy(key, f); }
    key_store(k->impl, out, (uint64_t)i, key);
  }
  return GJX_OK;
}

int gjx_rng_split_each(const gjx_keys* k, uint64_t n, uint32_t m, uint32_t* out, gjx_stream s) {
  (void)s;
  if (!keys_ok(k) || (!out && n) || m == 0) return GJX_ERR_INVALID;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    uint32_t key[4], child[4];
    key_at(k, (uint64_t)i, key);
    if (k->has_fold) { uint32_t f[4]; o_fold_in(k->impl, key, k->fold, f); o_key_copy(key, f); }
    for (uint32_t j = 0; j < m; ++j) {
      o_split_at(k->impl, key, j, child);
      key_store(k->impl, out, (uint64_t)i * m + j, child);
    }
  }
  return GJX_OK;
}

int gjx_rng_bits(const gjx_keys* k, uint32_t sub, uint64_t n, uint32_t* out, gjx_stream s) {
  (void)s;
  if (!keys_ok(k) || (!out && n)) return GJX_ERR_INVALID;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    o_stream st = stream_at(k, (uint64_t)i);
    out[i] = o_bits32_at(&st, sub);
  }
  return GJX_OK;
}

/* ---- categorical helpers --------------------------------------------------------------------- */
static float row_max(const float* l, uint32_t K) {
  float m = l[0];
  for (uint32_t c = 1; c < K; ++c) m = l[c] > m ? l[c] : m;
  return m;
}
/* lse of a logits row: max + log(sum_k exp(l_k - max)), f32, sequential in k. */
static float row_lse(const float* l, uint32_t K) {
  float m = row_max(l, K);
  float acc = 0.0f;
  for (uint32_t c = 0; c < K; ++c) acc = acc + o_exp(l[c] - m);
  return m + o_log(acc);
}
static inline uint32_t cat_fix(float l, float m) {
  if (l == m) return 1u << O_CAT_FRAC;
  float d = l - m;
  if (!(d >= -80.0f)) return 0u;
  return (uint32_t)rintf(o_exp(d) * 8388608.0f);
}
/* inverse-CDF draw on the fixed-point CDF of one logits row. */
static int32_t cat_invcdf(const float* l, uint32_t K, uint32_t bits) {
  float m = row_max(l, K);
  uint64_t Q = 0;
  for (uint32_t c = 0; c < K; ++c) Q += cat_fix(l[c], m);
  uint64_t thr = ((uint64_t)bits * Q) >> 32;
  uint64_t C = 0;
  for (uint32_t c = 0; c < K; ++c) {
    C += cat_fix(l[c], m);
    if (C > thr) return (int32_t)c;
  }
  return (int32_t)(K - 1);
}
static inline float gumbel_from_bits(uint32_t bits) {
  const float tiny = 1.17549435e-38f;
  float u = o_uniform01(bits) + tiny; /* f*(1-tiny)+tiny with 1-tiny == 1 in f32 */
  u = u > tiny ? u : tiny;
  return -o_log(-o_log(u));
}
static int32_t cat_gumbel(const float* l, uint32_t K, const o_stream* st) {
  int32_t best = 0;
  float bv = -INFINITY;
  for (uint32_t c = 0; c < K; ++c) {
    float v = l[c] + gumbel_from_bits(o_bits32_at(st, c));
    if (v > bv || c == 0) { bv = v; best = (int32_t)c; }
  }
  return best;
}
static inline const float* cat_row(const float* logits, uint64_t n_rows, uint32_t K,
                                   const int32_t* row_index, uint64_t i) {
  uint64_t r = row_index ? (uint64_t)row_index[i] : (n_rows == 1 ? 0 : i);
  return logits + r * K;
}

/* ---- elementwise distributions --------------------------------------------------------------- */
#define OPND(a, i) ((a).ptr ? (a).ptr[i] : (a).scalar)

int gjx_sample_logpdf_normal(const gjx_keys* k, gjx_f32 loc, gjx_f32 scale, float* value_out,
                             float* score_out, uint64_t n, gjx_stream s) {
  (void)s;
  if (!keys_ok(k) || !value_out) return GJX_ERR_INVALID;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    o_stream st = stream_at(k, (uint64_t)i);
    float mu = OPND(loc, i), sg = OPND(scale, i);
    float eps = o_site_normal(&st);
    float t = sg * eps;
    float v = mu + t;
    value_out[i] = v;
    if (score_out) score_out[i] = o_logpdf_normal(v, mu, sg);
  }
  return GJX_OK;
}

int gjx_sample_logpdf_gamma(const gjx_keys* k, gjx_f32 concentration, gjx_f32 rate,
                            float* value_out, float* score_out, uint64_t n, gjx_stream s) {
  (void)s;
  if (!keys_ok(k) || !value_out) return GJX_ERR_INVALID;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    o_stream st = stream_at(k, (uint64_t)i);
    float a = OPND(concentration, i), b = OPND(rate, i);
    float v = o_std_gamma(&st, 0, a) / b;
    value_out[i] = v;
    if (score_out) score_out[i] = o_logpdf_gamma(v, a, b);
  }
  return GJX_OK;
}

int gjx_sample_logpdf_beta(const gjx_keys* k, gjx_f32 a_, gjx_f32 b_, float* value_out,
                           float* score_out, uint64_t n, gjx_stream s) {
  (void)s;
  if (!keys_ok(k) || !value_out) return GJX_ERR_INVALID;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    o_stream st = stream_at(k, (uint64_t)i);
    float a = OPND(a_, i), b = OPND(b_, i);
    float g1 = o_std_gamma(&st, 0, a);
    float g2 = o_std_gamma(&st, 1, b);
    float v = g1 / (g1 + g2);
    value_out[i] = v;
    if (score_out) score_out[i] = o_logpdf_beta(v, a, b);
  }
  return GJX_OK;
}

int gjx_sample_logpdf_bernoulli(const gjx_keys* k, gjx_f32 probs, uint8_t* value_out,
                                float* score_out, uint64_t n, gjx_stream s) {
  (void)s;
  if (!keys_ok(k) || !value_out) return GJX_ERR_INVALID;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    o_stream st = stream_at(k, (uint64_t)i);
    float p = OPND(probs, i);
    int e = o_uniform01(o_bits32_at(&st, 0)) < p;
    value_out[i] = (uint8_t)e;
    if (score_out) score_out[i] = o_logpdf_bernoulli(e, p);
  }
  return GJX_OK;
}

int gjx_sample_logpdf_categorical(const gjx_keys* k, const float* logits, uint64_t n_rows,
                                  uint32_t n_cat, const int32_t* row_index, int mode,
                                  int32_t* value_out, float* score_out, uint64_t n, gjx_stream s) {
  (void)s;
  if (!keys_ok(k) || !value_out || !logits || n_cat == 0 || (mode != 0 && mode != 1))
    return GJX_ERR_INVALID;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    o_stream st = stream_at(k, (uint64_t)i);
    const float* l = cat_row(logits, n_rows, n_cat, row_index, (uint64_t)i);
    int32_t v = mode == 0 ? cat_gumbel(l, n_cat, &st) : cat_invcdf(l, n_cat, o_bits32_at(&st, 0));
    value_out[i] = v;
    if (score_out) score_out[i] = l[v] - row_lse(l, n_cat);
  }
  return GJX_OK;
}

int gjx_logpdf_normal(gjx_f32 value, gjx_f32 loc, gjx_f32 scale, float* score_out, uint64_t n,
                      gjx_stream s) {
  (void)s;
  if (!score_out) return GJX_ERR_INVALID;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i)
    score_out[i] = o_logpdf_normal(OPND(value, i), OPND(loc, i), OPND(scale, i));
  return GJX_OK;
}
int gjx_logpdf_gamma(gjx_f32 value, gjx_f32 concentration, gjx_f32 rate, float* score_out,
                     uint64_t n, gjx_stream s) {
  (void)s;
  if (!score_out) return GJX_ERR_INVALID;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i)
    score_out[i] = o_logpdf_gamma(OPND(value, i), OPND(concentration, i), OPND(rate, i));
  return GJX_OK;
}
int gjx_logpdf_beta(gjx_f32 value, gjx_f32 a, gjx_f32 b, float* score_out, uint64_t n,
                    gjx_stream s) {
  (void)s;
  if (!score_out) return GJX_ERR_INVALID;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i)
    score_out[i] = o_logpdf_beta(OPND(value, i), OPND(a, i), OPND(b, i));
  return GJX_OK;
}
int gjx_logpdf_bernoulli(const uint8_t* value, int value_scalar, gjx_f32 probs, float* score_out,
                         uint64_t n, gjx_stream s) {
  (void)s;
  if (!score_out) return GJX_ERR_INVALID;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i)
    score_out[i] = o_logpdf_bernoulli(value ? value[i] != 0 : value_scalar != 0, OPND(probs, i));
  return GJX_OK;
}
int gjx_logpdf_categorical(const int32_t* value, int value_scalar, const float* logits,
                           uint64_t n_rows, uint32_t n_cat, const int32_t* row_index,
                           float* score_out, uint64_t n, gjx_stream s) {
  (void)s;
  if (!score_out || !logits || n_cat == 0) return GJX_ERR_INVALID;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    const float* l = cat_row(logits, n_rows, n_cat, row_index, (uint64_t)i);
    int32_t v = value ? value[i] : value_scalar;
    score_out[i] = (v < 0 || (uint32_t)v >= n_cat) ? -INFINITY : l[v] - row_lse(l, n_cat);
  }
  return GJX_OK;
}

/* ---- GJX_ARG_EXPR: postfix programs as distribution arguments (gjx.h) ----------------------------- */
/* Well-formed: at most GJX_MAX_EXPR_OPS entries, operands in range for the plan kind (n_state < 0: importance plan — INPUT /
 * PARAM; otherwise STATE (if allowed) / OBS), the stack never deeper than 8 and exactly one value left at the end. */
static int expr_ok(const gjx_arg* a, int s, int n_state, int n_obs, int allow_state) {
  const gjx_expr_op* ops = (const gjx_expr_op*)(const void*)a->table;
  if (!ops || a->ref < 1 || a->ref > GJX_MAX_EXPR_OPS) return 0;
  int depth = 0;
  for (int k = 0; k < a->ref; ++k) {
    switch (ops[k].op) {
      case GJX_EXPR_CONST: ++depth; break;
      case GJX_EXPR_SITE: if (ops[k].ref < 0 || ops[k].ref >= s) return 0; ++depth; break;
      case GJX_EXPR_INPUT: if (n_state >= 0 || ops[k].ref < 0 || ops[k].ref >= 16) return 0; ++depth; break;
      case GJX_EXPR_PARAM: if (n_state >= 0 || ops[k].ref < 0 || ops[k].ref >= GJX_MAX_PARAMS) return 0; ++depth; break;
      case GJX_EXPR_STATE: if (n_state < 0 || !allow_state || ops[k].ref < 0 || ops[k].ref >= n_state) return 0; ++depth; break;
      case GJX_EXPR_OBS: if (n_state < 0 || ops[k].ref < 0 || ops[k].ref >= n_obs) return 0; ++depth; break;
      case GJX_EXPR_ADD: case GJX_EXPR_SUB: case GJX_EXPR_MUL: case GJX_EXPR_DIV: case GJX_EXPR_MAX: case GJX_EXPR_MIN:
      case GJX_EXPR_LT: case GJX_EXPR_LE: case GJX_EXPR_EQ:
        if (depth < 2) return 0;
        --depth;
        break;
      case GJX_EXPR_SELECT:
        if (depth < 3) return 0;
        depth -= 2;
        break;
      case GJX_EXPR_NEG: case GJX_EXPR_EXP: case GJX_EXPR_LOG: case GJX_EXPR_SQRT: case GJX_EXPR_ABS: if (depth < 1) return 0; break;
      default: return 0;
    }
    if (depth > 8) return 0;
  }
  return depth == 1;
}
/* The plan's own copy of every program of a site table (the caller's arrays need not outlive plan creation). */
typedef struct { gjx_expr_op ops[GJX_MAX_SITES][2][GJX_MAX_EXPR_OPS]; } expr_store;
static void expr_adopt(gjx_site* sites, int n, expr_store* st) {
  for (int q = 0; q < n; ++q)
    for (int a = 0; a < 2; ++a)
      if (sites[q].arg[a].kind == GJX_ARG_EXPR) {
        memcpy(st->ops[q][a], (const void*)sites[q].arg[a].table, sizeof(gjx_expr_op) * (size_t)sites[q].arg[a].ref);
        sites[q].arg[a].table = (const float*)(const void*)st->ops[q][a];
      }
}

/* ---- fused static-model importance ------------------------------------------------------------ */
/* Nested calls (gjx.h gjx_scope): which key every site draws under and with which fold, derived once per plan. */
typedef struct {
  int n_scopes;                         /* calls; 0 = a flat body (the implicit numbering) */
  int site_scope[GJX_MAX_SITES];        /* 0 = the body's own key */
  uint32_t fold_t[GJX_MAX_SITES];       /* THREEFRY: the site's 1-based `@` counter within its scope */
  uint32_t fold_p[GJX_MAX_SITES];       /* PHILOX: its 0-based index among the scope's randomness-consuming sites / calls */
  int parent[GJX_MAX_SCOPES + 1], begin[GJX_MAX_SCOPES + 1], end[GJX_MAX_SCOPES + 1];
  uint32_t s_fold_t[GJX_MAX_SCOPES + 1], s_fold_p[GJX_MAX_SCOPES + 1]; /* the counter the CALL took in its caller */
} scope_info;

static int derive_scopes(const gjx_site* sites, int n_sites, const gjx_scope* sc, int n_sc, scope_info* out) {
  if (n_sc < 0 || n_sc > GJX_MAX_SCOPES || (n_sc && !sc)) return 0;
  struct { int id, end; uint32_t ct, dr; } fr[8];
  int depth = 0, next = 0;
  fr[depth].id = 0; fr[depth].end = n_sites; fr[depth].ct = 1; fr[depth].dr = 0; ++depth;
  out->n_scopes = n_sc;
  out->parent[0] = -1;
  for (int q = 0; q <= n_sites; ++q) {
    while (next < n_sc && sc[next].begin == q) { /* calls made at this position, in call order */
      const gjx_scope* k = &sc[next];
      if (k->parent < 0 || k->parent > next || k->end < k->begin || k->end > n_sites) return 0;
      while (depth > 0 && fr[depth - 1].id != k->parent) {
        if (fr[depth - 1].end > q) return 0; /* the caller is not the innermost open scope */
        --depth;
      }
      if (depth == 0 || k->end > fr[depth - 1].end || depth >= 5) return 0;
      out->parent[next + 1] = k->parent;
      out->begin[next + 1] = k->begin;
      out->end[next + 1] = k->end;
      out->s_fold_t[next + 1] = fr[depth - 1].ct++;
      out->s_fold_p[next + 1] = fr[depth - 1].dr++;
      fr[depth].id = next + 1; fr[depth].end = k->end; fr[depth].ct = 1; fr[depth].dr = 0; ++depth;
      ++next;
    }
    if (next < n_sc && sc[next].begin < q) return 0; /* not in call order */
    while (depth > 1 && fr[depth - 1].end <= q) --depth;
    if (q == n_sites) break;
    out->site_scope[q] = fr[depth - 1].id;
    out->fold_t[q] = fr[depth - 1].ct++;
    out->fold_p[q] = fr[depth - 1].dr;
    if (!sites[q].observed) fr[depth - 1].dr++;
  }
  return next == n_sc;
}

struct gjx_plan {
  int n_sites;
  gjx_site sites[GJX_MAX_SITES];
  int n_params;
  float params[GJX_MAX_PARAMS]; /* GJX_ARG_PARAM values (gjx_plan_set_params) */
  expr_store expr;
  scope_info scopes;
};

static int arg_ok(const gjx_arg* a, int s, int allow_site) {
  switch (a->kind) {
    case GJX_ARG_CONST: return 1;
    case GJX_ARG_SITE: return allow_site && a->ref >= 0 && a->ref < s;
    case GJX_ARG_INPUT: return a->ref >= 0;
    case GJX_ARG_TABLE: return allow_site && a->ref >= 0 && a->ref < s && a->table != NULL;
    case GJX_ARG_PARAM: return a->ref >= 0 && a->ref < GJX_MAX_PARAMS;
    case GJX_ARG_EXPR: return allow_site && expr_ok(a, s, -1, -1, 0);
    default: return 0;
  }
}

int gjx_plan_create(const gjx_site* sites, int n_sites, gjx_plan** out) {
  if (!sites || !out || n_sites <= 0 || n_sites > GJX_MAX_SITES) return GJX_ERR_INVALID;
  for (int s = 0; s < n_sites; ++s) {
    const gjx_site* st = &sites[s];
    if (st->dist < 0 || st->dist > GJX_DIST_CATEGORICAL) return GJX_ERR_INVALID;
    if (!arg_ok(&st->arg[0], s, 1)) return GJX_ERR_INVALID;
    if (st->dist != GJX_DIST_BERNOULLI && st->dist != GJX_DIST_CATEGORICAL &&
        !arg_ok(&st->arg[1], s, 1))
      return GJX_ERR_INVALID;
    if (st->observed && !(st->obs.kind == GJX_ARG_CONST || st->obs.kind == GJX_ARG_INPUT ||
                          (st->obs.kind == GJX_ARG_PARAM && st->obs.ref >= 0 && st->obs.ref < GJX_MAX_PARAMS)))
      return GJX_ERR_INVALID;
    if (st->dist == GJX_DIST_CATEGORICAL &&
        (!st->logits || st->n_cat <= 0 || st->n_rows <= 0 || (st->cat_mode != 0 && st->cat_mode != 1) ||
         st->arg[0].kind == GJX_ARG_EXPR))
      return GJX_ERR_INVALID;
  }
  gjx_plan* p = (gjx_plan*)calloc(1, sizeof(gjx_plan));
  if (!p) return GJX_ERR_LAUNCH;
  p->n_sites = n_sites;
  memcpy(p->sites, sites, sizeof(gjx_site) * (size_t)n_sites);
  expr_adopt(p->sites, n_sites, &p->expr);
  *out = p;
  return GJX_OK;
}
int gjx_plan_create_ex(const gjx_site* sites, int n_sites, uint32_t flags, gjx_plan** out) {
  if (flags & ~(uint32_t)GJX_PLAN_FAST_MATH) return GJX_ERR_INVALID;
  return gjx_plan_create(sites, n_sites, out); /* the oracle is the exact specification: FAST_MATH is not its concern */
}
int gjx_plan_create_scoped(const gjx_site* sites, int n_sites, const gjx_scope* scopes, int n_scopes, uint32_t flags,
                           gjx_plan** out) {
  gjx_plan* p = NULL;
  int rc = gjx_plan_create_ex(sites, n_sites, flags, &p);
  if (rc) return rc;
  if (!derive_scopes(p->sites, p->n_sites, scopes, n_scopes, &p->scopes)) {
    gjx_plan_destroy(p);
    return GJX_ERR_INVALID;
  }
  *out = p;
  return GJX_OK;
}
int gjx_plan_destroy(gjx_plan* p) { free(p); return GJX_OK; }
static int plan_max_param(const gjx_plan* p) {
  int mx = -1;
  for (int q = 0; q < p->n_sites; ++q) {
    const gjx_site* st = &p->sites[q];
    for (int a = 0; a < 2; ++a) {
      if (st->arg[a].kind == GJX_ARG_PARAM && st->arg[a].ref > mx) mx = st->arg[a].ref;
      if (st->arg[a].kind == GJX_ARG_EXPR) {
        const gjx_expr_op* ops = (const gjx_expr_op*)(const void*)st->arg[a].table;
        for (int k = 0; k < st->arg[a].ref; ++k)
          if (ops[k].op == GJX_EXPR_PARAM && ops[k].ref > mx) mx = ops[k].ref;
      }
    }
    if (st->observed && st->obs.kind == GJX_ARG_PARAM && st->obs.ref > mx) mx = st->obs.ref;
  }
  return mx;
}
int gjx_plan_set_params(gjx_plan* p, const float* params, int n_params) {
  if (!p || n_params < 0 || n_params > GJX_MAX_PARAMS || (n_params && !params) || n_params <= plan_max_param(p))
    return GJX_ERR_INVALID;
  memcpy(p->params, params, sizeof(float) * (size_t)n_params);
  p->n_params = n_params;
  return GJX_OK;
}
int gjx_plan_specialized_source(const gjx_plan* p, int impl, char* buf, size_t buf_len, size_t* needed) {
  (void)p; (void)impl; (void)buf; (void)buf_len; (void)needed;
  return GJX_ERR_UNSUPPORTED;
}
int gjx_plan_compile_check(const gjx_plan* p, int impl) { (void)p; (void)impl; return GJX_ERR_UNSUPPORTED; }
int gjx_jit_compile_source(const char* source) { (void)source; return GJX_ERR_UNSUPPORTED; }
int gjx_jit_stats(uint64_t* compiles, uint64_t* cached_modules, uint64_t* evictions) {
  if (compiles) *compiles = 0;
  if (cached_modules) *cached_modules = 0;
  if (evictions) *evictions = 0;
  return GJX_OK;
}

int gjx_jit_routes(uint64_t* child_compiles, uint64_t* inproc_compiles, uint64_t* child_failures, uint64_t* spawn_failures) {
  if (child_compiles) *child_compiles = 0;
  if (inproc_compiles) *inproc_compiles = 0;
  if (child_failures) *child_failures = 0;
  if (spawn_failures) *spawn_failures = 0;
  return GJX_OK;
}

int gjx_smc_run_graph_stats(uint64_t* captures, uint64_t* replays) {
  if (captures) *captures = 0;
  if (replays) *replays = 0;
  return GJX_OK;
}

typedef struct { float f; int32_t i; int is_int; } site_val;

static inline float sv_as_f32(const site_val* v) { return v->is_int ? (float)v->i : v->f; }
static inline int32_t sv_as_i32(const site_val* v) { return v->is_int ? v->i : (int32_t)rintf(v->f); }

/* Everything one particle's walk over a site table reads. */
typedef struct {
  int impl;
  uint32_t pkey[4];          /* importance: particle key; smc: slot key split(step_key)[slot] */
  int pair_normals;          /* importance: 1 */
  const float* const* in;    /* importance: input columns */
  uint64_t i;                /* importance: particle index into the input columns */
  const uint32_t* quad_key;  /* smc under PHILOX: the step key — one-word draws come from the slot's quad block */
  uint64_t slot;             /* smc: the output slot */
  const float* state;        /* smc: the ancestor's state columns (NULL at step 0) */
  const float* obs;          /* smc: this step's observation constants */
  const float* params;       /* importance: the plan's GJX_ARG_PARAM values */
  const scope_info* scopes;  /* importance plans with nested calls (NULL / n_scopes 0: a flat body) */
} walk_ctx;

static inline float eval_arg(const gjx_arg* a, const site_val* vals, const walk_ctx* c) {
  switch (a->kind) {
    case GJX_ARG_CONST: return a->offset;
    case GJX_ARG_SITE: { float t = a->scale * sv_as_f32(&vals[a->ref]); return t + a->offset; }
    case GJX_ARG_INPUT: { float t = a->scale * c->in[a->ref][c->i]; return t + a->offset; }
    case GJX_ARG_STATE: { float t = a->scale * c->state[a->ref]; return t + a->offset; }
    case GJX_ARG_OBS: { float t = a->scale * c->obs[a->ref]; return t + a->offset; }
    case GJX_ARG_PARAM: { float t = a->scale * c->params[a->ref]; return t + a->offset; }
    case GJX_ARG_EXPR: {
      const gjx_expr_op* ops = (const gjx_expr_op*)(const void*)a->table;
      float st[8];
      int d = 0;
      for (int k = 0; k < a->ref; ++k) {
        switch (ops[k].op) {
          case GJX_EXPR_CONST: st[d++] = ops[k].value; break;
          case GJX_EXPR_SITE: st[d++] = sv_as_f32(&vals[ops[k].ref]); break;
          case GJX_EXPR_INPUT: st[d++] = c->in[ops[k].ref][c->i]; break;
          case GJX_EXPR_PARAM: st[d++] = c->params[ops[k].ref]; break;
          case GJX_EXPR_STATE: st[d++] = c->state[ops[k].ref]; break;
          case GJX_EXPR_OBS: st[d++] = c->obs[ops[k].ref]; break;
          case GJX_EXPR_ADD: { const float r = st[d - 2] + st[d - 1]; st[--d - 1] = r; break; }
          case GJX_EXPR_SUB: { const float r = st[d - 2] - st[d - 1]; st[--d - 1] = r; break; }
          case GJX_EXPR_MUL: { const float r = st[d - 2] * st[d - 1]; st[--d - 1] = r; break; }
          case GJX_EXPR_DIV: { const float r = st[d - 2] / st[d - 1]; st[--d - 1] = r; break; }
          case GJX_EXPR_MAX: { const float r = o_e_max(st[d - 2], st[d - 1]); st[--d - 1] = r; break; }
          case GJX_EXPR_MIN: { const float r = o_e_min(st[d - 2], st[d - 1]); st[--d - 1] = r; break; }
          case GJX_EXPR_LT: { const float r = st[d - 2] < st[d - 1] ? 1.0f : 0.0f; st[--d - 1] = r; break; }
          case GJX_EXPR_LE: { const float r = st[d - 2] <= st[d - 1] ? 1.0f : 0.0f; st[--d - 1] = r; break; }
          case GJX_EXPR_EQ: { const float r = st[d - 2] == st[d - 1] ? 1.0f : 0.0f; st[--d - 1] = r; break; }
          case GJX_EXPR_SELECT: { const float r = st[d - 3] != 0.0f ? st[d - 2] : st[d - 1]; d -= 2; st[d - 1] = r; break; }
          case GJX_EXPR_EXP: st[d - 1] = o_e_exp(st[d - 1]); break;
          case GJX_EXPR_LOG: st[d - 1] = o_log(st[d - 1]); break;
          case GJX_EXPR_SQRT: st[d - 1] = sqrtf(st[d - 1]); break;
          case GJX_EXPR_ABS: st[d - 1] = fabsf(st[d - 1]); break;
          default: st[d - 1] = -st[d - 1]; break; /* GJX_EXPR_NEG */
        }
      }
      return st[0];
    }
    default: return a->table[sv_as_i32(&vals[a->ref])];
  }
}

/* The walk of static.py:340-399 for one particle: values into vals[], returns weight / score. */
static void site_walk(const gjx_site* sites, int n_sites, const walk_ctx* c, site_val* vals, float* w_out,
                      float* sc_out) {
  float w = 0.0f, sc = 0.0f;
  uint32_t draws = 0;
  /* nested calls: scope k draws under fold_in(key of its caller, the counter the call took) */
  const scope_info* si = (c->scopes && c->scopes->n_scopes > 0) ? c->scopes : NULL;
  uint32_t skey[GJX_MAX_SCOPES + 1][4];
  float sw[GJX_MAX_SCOPES + 1], ssc[GJX_MAX_SCOPES + 1];
  for (int k = 0; k <= GJX_MAX_SCOPES; ++k) sw[k] = ssc[k] = 0.0f;
  if (si) {
    memcpy(skey[0], c->pkey, sizeof skey[0]);
    for (int k = 1; k <= si->n_scopes; ++k)
      o_fold_in(c->impl, skey[si->parent[k]], c->impl == 0 ? si->s_fold_t[k] : si->s_fold_p[k], skey[k]);
  }
  for (int q = 0; q < n_sites; ++q) {
    const gjx_site* st = &sites[q];
    site_val v;
    v.f = 0.0f; v.i = 0;
    v.is_int = (st->dist == GJX_DIST_BERNOULLI || st->dist == GJX_DIST_CATEGORICAL);
    float a0 = 0.0f, a1 = 0.0f;
    const float* row = NULL;
    if (st->dist == GJX_DIST_CATEGORICAL) {
      int32_t r = 0;
      if (st->arg[0].kind == GJX_ARG_SITE) r = sv_as_i32(&vals[st->arg[0].ref]);
      else if (st->arg[0].kind == GJX_ARG_CONST) r = (int32_t)rintf(st->arg[0].offset);
      else r = (int32_t)rintf(eval_arg(&st->arg[0], vals, c));
      if (r < 0) r = 0;
      if (r >= st->n_rows) r = st->n_rows - 1;
      row = st->logits + (size_t)r * (size_t)st->n_cat;
    } else {
      a0 = eval_arg(&st->arg[0], vals, c);
      if (st->dist != GJX_DIST_BERNOULLI) a1 = eval_arg(&st->arg[1], vals, c);
    }
    float lp;
    if (st->observed) {
      float ov = st->obs.kind == GJX_ARG_CONST ? st->obs.offset
               : st->obs.kind == GJX_ARG_OBS ? c->obs[st->obs.ref]
               : st->obs.kind == GJX_ARG_PARAM ? eval_arg(&st->obs, vals, c) : c->in[st->obs.ref][c->i];
      if (v.is_int) v.i = (int32_t)rintf(ov); else v.f = ov;
    } else {
      /* THREEFRY: site counter from 1 (static.py:349-352); PHILOX: index among the sampled sites */
      const uint32_t f = si ? (c->impl == 0 ? si->fold_t[q] : si->fold_p[q]) : (c->impl == 0 ? (uint32_t)(q + 1) : draws);
      ++draws;
      o_stream strm = o_stream_make(c->impl, si ? skey[si->site_scope[q]] : c->pkey, 1, f);
      /* one-word draws: SMC slots under PHILOX take word (slot & 3) of their quad's block number f */
      /* (a callee's sites draw under their own lone key: the quad's blocks serve the body's own sites only) */
      const uint32_t* quad_key = (si && si->site_scope[q] != 0) ? NULL : c->quad_key;
      const uint32_t bits0 = quad_key ? o_smc_quad_word(quad_key, c->slot, f) : o_bits32_at(&strm, 0);
      switch (st->dist) {
        case GJX_DIST_NORMAL: { /* PHILOX pairs draws: importance over particle pairs, SMC inside the slot's quad */
          float eps = quad_key ? o_smc_quad_normal(quad_key, c->slot, f)
                    : c->pair_normals ? o_site_normal(&strm) : o_std_normal(bits0);
          float t = a1 * eps; v.f = a0 + t; break; }
        case GJX_DIST_GAMMA: v.f = o_std_gamma(&strm, 0, a0) / a1; break;
        case GJX_DIST_BETA: { float g1 = o_std_gamma(&strm, 0, a0), g2 = o_std_gamma(&strm, 1, a1); v.f = g1 / (g1 + g2); break; }
        case GJX_DIST_BERNOULLI: v.i = o_uniform01(bits0) < a0; break;
        default: v.i = st->cat_mode == 0 ? cat_gumbel(row, (uint32_t)st->n_cat, &strm)
                                         : cat_invcdf(row, (uint32_t)st->n_cat, bits0);
      }
    }
    switch (st->dist) {
      case GJX_DIST_NORMAL: lp = o_logpdf_normal(v.f, a0, a1); break;
      case GJX_DIST_GAMMA: lp = o_logpdf_gamma(v.f, a0, a1); break;
      case GJX_DIST_BETA: lp = o_logpdf_beta(v.f, a0, a1); break;
      case GJX_DIST_BERNOULLI: lp = o_logpdf_bernoulli(v.i != 0, a0); break;
      default: lp = (v.i < 0 || v.i >= st->n_cat) ? -INFINITY : row[v.i] - row_lse(row, (uint32_t)st->n_cat);
    }
    if (si && si->site_scope[q] != 0) { /* a callee's weight and score are ITS totals (static.py:374-380) */
      const int k = si->site_scope[q];
      ssc[k] = ssc[k] + lp;
      if (st->observed) sw[k] = sw[k] + lp;
    } else {
      sc = sc + lp;
      if (st->observed) w = w + lp;
    }
    vals[q] = v;
    if (si)
      for (int k = si->n_scopes; k >= 1; --k) /* the calls that return here, inner ones first */
        if (si->end[k] == q + 1 && si->begin[k] < q + 1) {
          const int pa = si->parent[k];
          if (pa == 0) { w = w + sw[k]; sc = sc + ssc[k]; }
          else { sw[pa] = sw[pa] + sw[k]; ssc[pa] = ssc[pa] + ssc[k]; }
        }
  }
  *w_out = w;
  *sc_out = sc;
}

int gjx_plan_prepare(gjx_plan* p, const gjx_keys* pk) { return (p && keys_ok(pk)) ? GJX_OK : GJX_ERR_INVALID; }
int gjx_map_f32(int op, const float* x, float c, float* out, uint64_t n, gjx_stream s) {
  (void)s;
  if (!x || !out || op < GJX_MAP_EXP || op > GJX_MAP_ABS) return GJX_ERR_INVALID;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    const float v = x[i];
    out[i] = op == GJX_MAP_EXP ? o_e_exp(v) : op == GJX_MAP_LOG ? o_log(v) : op == GJX_MAP_DIV ? v / c : op == GJX_MAP_RDIV ? c / v
             : op == GJX_MAP_SQRT ? sqrtf(v) : fabsf(v);
  }
  return GJX_OK;
}

int gjx_importance_run(const gjx_plan* p, const gjx_keys* pk, const float* const* input_cols,
                       int n_input_cols, void* const* value_cols, int n_value_cols, float* score,
                       float* logw, uint64_t n, float* max_partials, int32_t* row_e, uint64_t* row_s,
                       const gjx_lse_out* lse, gjx_stream s) {
  (void)s;
  if (!p || !keys_ok(pk) || pk->has_fold || (!logw && !row_e) || ((row_e == NULL) != (row_s == NULL)) ||
      (lse && (!row_e || !row_s || !lse->tickets)))
    return GJX_ERR_INVALID;
  if (!logw) { /* only the row sums are asked for: the weights live in a scratch column */
    float* tmp = (float*)malloc(sizeof(float) * (n ? n : 1));
    if (!tmp) return GJX_ERR_LAUNCH;
    int rc = gjx_importance_run(p, pk, input_cols, n_input_cols, value_cols, n_value_cols, score, tmp, n, max_partials, row_e,
                                row_s, lse, s);
    free(tmp);
    return rc;
  }
  for (int q = 0; q < p->n_sites; ++q) {
    const gjx_site* st = &p->sites[q];
    if (st->out_col >= n_value_cols) return GJX_ERR_INVALID;
    for (int a = 0; a < 2; ++a)
      if (st->arg[a].kind == GJX_ARG_INPUT && st->arg[a].ref >= n_input_cols) return GJX_ERR_INVALID;
    if (st->observed && st->obs.kind == GJX_ARG_INPUT && st->obs.ref >= n_input_cols)
      return GJX_ERR_INVALID;
  }
  if (plan_max_param(p) >= p->n_params) return GJX_ERR_INVALID; /* parameters referenced but never set */
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    walk_ctx c;
    memset(&c, 0, sizeof c);
    c.impl = pk->impl;
    c.params = p->params;
    c.scopes = &p->scopes;
    c.pair_normals = 1;
    key_at(pk, (uint64_t)i, c.pkey);
    c.in = input_cols;
    c.i = (uint64_t)i;
    site_val vals[GJX_MAX_SITES];
    float w, sc;
    site_walk(p->sites, p->n_sites, &c, vals, &w, &sc);
    for (int q = 0; q < p->n_sites; ++q) {
      const gjx_site* st = &p->sites[q];
      if (st->out_col >= 0) {
        if (vals[q].is_int) ((int32_t*)value_cols[st->out_col])[i] = vals[q].i;
        else ((float*)value_cols[st->out_col])[i] = vals[q].f;
      }
    }
    logw[i] = w;
    if (score) score[i] = sc;
  }
  if (max_partials) {
    for (uint64_t b = 0; b * O_ROW < n; ++b) {
      float m = -INFINITY;
      for (uint64_t i = b * O_ROW; i < n && i < (b + 1) * O_ROW; ++i) m = logw[i] > m ? logw[i] : m;
      max_partials[b] = m;
    }
  }
  if (row_e && row_s) {
    int rc = gjx_row_stats(logw, n, row_e, row_s, s);
    if (rc || !lse) return rc;
    float l = 0.0f;
    rc = gjx_lse_rows(row_e, row_s, gjx_num_max_partials(n), lse->e, lse->q, &l, lse->record, s);
    if (lse->lse) *lse->lse = l;
    if (lse->lse_shifted) *lse->lse_shifted = l - lse->shift;
    return rc;
  }
  return GJX_OK;
}

int gjx_importance_estimate(const gjx_estimate_io* io, uint32_t k0, uint32_t k1, uint64_t lane, float* out, float shift,
                            gjx_stream s) {
  if (!io || !io->plan || !io->row_e || !io->row_s || !io->lse.tickets || !out || (io->impl != 0 && io->impl != 1) ||
      (io->impl == 0 && lane != 0))
    return GJX_ERR_INVALID;
  for (int q = 0; q < io->plan->n_sites; ++q)
    if (io->plan->sites[q].out_col >= 0) return GJX_ERR_INVALID;
  /* key, sub = split(key); key, sub = split(sub); particle keys = split(sub, K) (inference/smc.py:86, 299-300) */
  gjx_keys k;
  memset(&k, 0, sizeof k);
  k.impl = io->impl; k.mode = 1;
  k.parent[0] = k0; k.parent[1] = k1; k.parent_lane = lane;
  uint32_t c[4];
  for (int rep = 0; rep < 2; ++rep) {
    key_at(&k, 1, c);
    k.parent[0] = c[0]; k.parent[1] = c[1];
    k.parent_lane = io->impl == 1 ? (((uint64_t)c[3] << 32) | c[2]) : 0;
  }
  gjx_lse_out lse = io->lse;
  lse.lse_shifted = out;
  lse.shift = shift;
  return gjx_importance_run(io->plan, &k, io->input_cols, io->n_input_cols, NULL, 0, NULL, NULL, io->n, NULL, io->row_e, io->row_s,
                            &lse, s);
}

int gjx_importance_run_batch(const gjx_plan* p, const gjx_keys* pk, int32_t n_pass, uint64_t pass_stride,
                             uint64_t row_stride, const float* const* input_cols, int n_input_cols,
                             void* const* value_cols, int n_value_cols, float* score, float* logw, uint64_t n,
                             float* max_partials, int32_t* row_e, uint64_t* row_s, gjx_stream s) {
  if (!pk || n_pass < 1 || n_pass > 32 || pass_stride < n || row_stride < gjx_num_max_partials(n) ||
      n_value_cols < 0 || n_value_cols > GJX_MAX_SITES)
    return GJX_ERR_INVALID;
  for (int32_t b = 0; b < n_pass; ++b) { /* the definition: n_pass separate passes */
    void* vc[GJX_MAX_SITES];
    for (int c = 0; c < n_value_cols; ++c) vc[c] = value_cols[c] ? (char*)value_cols[c] + 4 * (size_t)b * pass_stride : NULL;
    int rc = gjx_importance_run(p, pk + b, input_cols, n_input_cols, vc, n_value_cols,
                                score ? score + (size_t)b * pass_stride : NULL, logw + (size_t)b * pass_stride, n,
                                max_partials ? max_partials + (size_t)b * row_stride : NULL,
                                row_e ? row_e + (size_t)b * row_stride : NULL,
                                row_s ? row_s + (size_t)b * row_stride : NULL, NULL, s);
    if (rc) return rc;
  }
  return GJX_OK;
}

/* ---- row-anchored weights ----------------------------------------------------- */
int gjx_row_stats(const float* x, uint64_t n, int32_t* row_e, uint64_t* row_s, gjx_stream s) {
  (void)s;
  if (!x || !row_e || !row_s || n == 0) return GJX_ERR_INVALID;
  const uint64_t nr = gjx_num_max_partials(n);
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < (int64_t)nr; ++b) {
    uint64_t lo = (uint64_t)b * O_ROW, hi = lo + O_ROW < n ? lo + O_ROW : n;
    float m = -INFINITY;
    for (uint64_t i = lo; i < hi; ++i) m = x[i] > m ? x[i] : m;
    int32_t e = o_row_anchor(m);
    uint64_t acc = 0;
    for (uint64_t i = lo; i < hi; ++i) acc += o_rowfix(x[i], e);
    row_e[b] = e;
    row_s[b] = acc;
  }
  return GJX_OK;
}
/* Shared tail: Q = sum_d B_d >> d, optional record, f32 lse. */
static void lse_emit(int32_t e, const uint64_t* B, int32_t* out_e, uint64_t* out_q, float* out_lse,
                     uint64_t* out_record) {
  uint64_t Q = 0;
  for (int d = 0; d < GJX_LSE_RECORD_WORDS - 1; ++d) Q += B[d] >> d;
  if (out_record) {
    out_record[0] = (uint64_t)(int64_t)e;
    memcpy(out_record + 1, B, sizeof(uint64_t) * (GJX_LSE_RECORD_WORDS - 1));
  }
  if (out_e) *out_e = e;
  if (out_q) *out_q = Q;
  if (out_lse) {
    if (e == O_ROW_EMPTY || Q == 0) *out_lse = -INFINITY;
    else {
      float t1 = (float)e * 0.69314718055994531f;
      float t2 = o_log((float)Q * o_u2f((uint32_t)(127 - O_ROW_FRAC) << 23));
      *out_lse = t1 + t2;
    }
  }
}
int gjx_lse_rows(const int32_t* row_e, const uint64_t* row_s, uint64_t n_rows, int32_t* out_e,
                 uint64_t* out_q, float* out_lse, uint64_t* out_record, gjx_stream s) {
  (void)s;
  if (!row_e || !row_s || n_rows == 0) return GJX_ERR_INVALID;
  int32_t e = O_ROW_EMPTY;
  for (uint64_t b = 0; b < n_rows; ++b) e = row_e[b] > e ? row_e[b] : e;
  uint64_t B[GJX_LSE_RECORD_WORDS - 1] = {0};
  for (uint64_t b = 0; b < n_rows; ++b) {
    if (row_e[b] == O_ROW_EMPTY) continue;
    int64_t d = (int64_t)e - (int64_t)row_e[b];
    if (d < GJX_LSE_RECORD_WORDS - 1) B[d] += row_s[b];
  }
  lse_emit(e, B, out_e, out_q, out_lse, out_record);
  return GJX_OK;
}
int gjx_lse_rows_batch(const int32_t* row_e, const uint64_t* row_s, uint64_t n_rows, int32_t n_batch,
                       uint64_t batch_stride, int32_t* out_e, uint64_t* out_q, float* out_lse,
                       uint64_t* out_record, gjx_stream s) {
  if (!row_e || !row_s || n_rows == 0 || n_batch < 1 || batch_stride < n_rows) return GJX_ERR_INVALID;
  for (int32_t b = 0; b < n_batch; ++b) {
    int rc = gjx_lse_rows(row_e + (uint64_t)b * batch_stride, row_s + (uint64_t)b * batch_stride, n_rows,
                          out_e ? out_e + b : NULL, out_q ? out_q + b : NULL, out_lse ? out_lse + b : NULL,
                          out_record ? out_record + (uint64_t)b * GJX_LSE_RECORD_WORDS : NULL, s);
    if (rc) return rc;
  }
  return GJX_OK;
}
int gjx_lse_combine(const uint64_t* records, int32_t n_records, uint64_t record_stride, int32_t n_batch,
                    uint64_t batch_stride, int32_t* out_e, uint64_t* out_q, float* out_lse,
                    uint64_t* out_record, gjx_stream s) {
  (void)s;
  if (!records || n_records < 1 || n_batch < 1 || record_stride < GJX_LSE_RECORD_WORDS) return GJX_ERR_INVALID;
  for (int32_t p = 0; p < n_batch; ++p) {
    const uint64_t* base = records + (uint64_t)p * batch_stride;
    int32_t e = O_ROW_EMPTY;
    for (int32_t r = 0; r < n_records; ++r) {
      int32_t er = (int32_t)(int64_t)base[(uint64_t)r * record_stride];
      e = er > e ? er : e;
    }
    uint64_t B[GJX_LSE_RECORD_WORDS - 1] = {0};
    for (int32_t r = 0; r < n_records; ++r) {
      const uint64_t* rec = base + (uint64_t)r * record_stride;
      int32_t er = (int32_t)(int64_t)rec[0];
      if (er == O_ROW_EMPTY) continue;
      int64_t off = (int64_t)e - (int64_t)er;
      for (int64_t k = 0; k + off < GJX_LSE_RECORD_WORDS - 1; ++k) B[k + off] += rec[1 + k];
    }
    lse_emit(e, B, out_e ? out_e + p : NULL, out_q ? out_q + p : NULL, out_lse ? out_lse + p : NULL,
             out_record ? out_record + (uint64_t)p * GJX_LSE_RECORD_WORDS : NULL);
  }
  return GJX_OK;
}

/* ---- weights ------------------------------------------------------------------------------------ */
int gjx_max_f32(const float* x, uint64_t n, const float* max_partials_in, float* out_max, void* ws,
                size_t ws_bytes, gjx_stream s) {
  (void)ws; (void)ws_bytes; (void)s;
  if ((!x && !max_partials_in) || !out_max || n == 0) return GJX_ERR_INVALID;
  float m = -INFINITY;
  if (max_partials_in) {
    for (uint64_t b = 0; b < gjx_num_max_partials(n); ++b) m = max_partials_in[b] > m ? max_partials_in[b] : m;
  } else {
#pragma omp parallel for reduction(max : m) schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) m = x[i] > m ? x[i] : m;
  }
  *out_max = m;
  return GJX_OK;
}
int gjx_expsum_fix(const float* x, uint64_t n, const float* max_dev, int frac_bits, uint64_t* out_q,
                   void* ws, size_t ws_bytes, gjx_stream s) {
  (void)ws; (void)ws_bytes; (void)s;
  if (!x || !max_dev || !out_q || frac_bits < 1 || frac_bits > 40) return GJX_ERR_INVALID;
  float m = *max_dev;
  uint64_t Q = 0;
#pragma omp parallel for reduction(+ : Q) schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i) Q += o_fixw(x[i], m, frac_bits);
  *out_q = Q;
  return GJX_OK;
}
static float lse_from(float m, uint64_t q, int frac) {
  /* (float)q is a correctly rounded u64->f32 conversion; scaling by 2^-frac is exact. */
  float qf = (float)q * o_u2f((uint32_t)(127 - frac) << 23);
  return m + o_log(qf);
}
int gjx_lse_finish(const float* max_dev, const uint64_t* q_dev, int frac_bits, float* out_lse,
                   gjx_stream s) {
  (void)s;
  if (!max_dev || !q_dev || !out_lse) return GJX_ERR_INVALID;
  *out_lse = lse_from(*max_dev, *q_dev, frac_bits);
  return GJX_OK;
}
int gjx_logsumexp_f32(const float* x, uint64_t n, const float* max_partials_in, float* out_lse,
                      float* out_max, uint64_t* out_q, void* ws, size_t ws_bytes, gjx_stream s) {
  float m;
  uint64_t q;
  int frac = o_frac_bits(n);
  if (!x) return GJX_ERR_INVALID;
  int rc = gjx_max_f32(x, n, max_partials_in, &m, ws, ws_bytes, s);
  if (rc) return rc;
  rc = gjx_expsum_fix(x, n, &m, frac, &q, ws, ws_bytes, s);
  if (rc) return rc;
  if (out_lse) *out_lse = lse_from(m, q, frac);
  if (out_max) *out_max = m;
  if (out_q) *out_q = q;
  return GJX_OK;
}

static inline uint64_t mulhi64(uint64_t a, uint64_t b) {
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
}

int gjx_categorical_index(const gjx_keys* key, const float* logits, uint64_t n, int64_t* out_idx,
                          int mode, void* ws, size_t ws_bytes, gjx_stream s) {
  (void)ws; (void)ws_bytes; (void)s;
  if (!keys_ok(key) || !logits || !out_idx || n == 0 || (mode != 0 && mode != 1)) return GJX_ERR_INVALID;
  o_stream st = stream_at(key, 0);
  if (mode == 0) {
    int64_t best = 0;
    float bv = -INFINITY;
    for (uint64_t i = 0; i < n; ++i) {
      float v = logits[i] + gumbel_from_bits(o_bits32_at(&st, (uint32_t)i));
      if (v > bv || i == 0) { bv = v; best = (int64_t)i; }
    }
    *out_idx = best;
    return GJX_OK;
  }
  float m = -INFINITY;
  for (uint64_t i = 0; i < n; ++i) m = logits[i] > m ? logits[i] : m;
  int frac = o_frac_bits(n);
  uint64_t Q = 0;
  for (uint64_t i = 0; i < n; ++i) Q += o_fixw(logits[i], m, frac);
  uint64_t thr = mulhi64(o_bits64_at(&st, 0), Q);
  uint64_t C = 0;
  int64_t idx = (int64_t)n - 1;
  for (uint64_t i = 0; i < n; ++i) {
    C += o_fixw(logits[i], m, frac);
    if (C > thr) { idx = (int64_t)i; break; }
  }
  *out_idx = idx;
  return GJX_OK;
}

int gjx_categorical_index_batch(const gjx_keys* keys, int32_t n_batch, const float* logits, uint64_t n, uint64_t stride,
                                int64_t* out_idx, gjx_stream s) {
  if (!keys || !logits || !out_idx || n_batch < 1 || n_batch > 64 || n == 0 || n > O_TILE || stride < n) return GJX_ERR_INVALID;
  for (int b = 0; b < n_batch; ++b) { /* by definition: the single draws */
    const int rc = gjx_categorical_index(&keys[b], logits + (size_t)b * stride, n, out_idx + b, 0, NULL, 0, s);
    if (rc) return rc;
  }
  return GJX_OK;
}

/* ---- tile-anchored weights (gjx.h gjx_tile_rec) ---------------------------------- *
 * Sequential restatement: per tile of 1024 particles the maximum (the sequential `x > m ? x : m`, a NaN is skipped),
 * its power-of-two anchor, the fixed-point weights (one u32 per particle), their running sum after every 64th
 * particle, the ESS sums. */
#define O_ESS_SHIFT (O_ROW_FRAC - 16)
#define O_SUBS 16
#define O_SUBLEN (O_TILE / O_SUBS)
static void tile_emit(const float* lw, uint64_t cnt, uint32_t* qw, gjx_tile_rec* rec, gjx_tile_sub* sub, gjx_tile_ess* ess) {
  float m = -INFINITY;
  for (uint64_t i = 0; i < cnt; ++i) m = lw[i] > m ? lw[i] : m;
  const int32_t e = o_row_anchor(m);
  uint64_t run = 0, r1 = 0, r2 = 0;
  for (uint64_t i = 0; i < O_TILE; ++i) {
    if (i < cnt) {
      const uint64_t q = o_rowfix(lw[i], e);
      qw[i] = (uint32_t)q;
      run += q;
      const uint64_t r = q >> O_ESS_SHIFT;
      r1 += r;
      r2 += r * r;
    }
    if ((i + 1) % O_SUBLEN == 0) sub->sub[i / O_SUBLEN] = run;
  }
  rec->s = run;
  rec->e = e;
  rec->pad = 0;
  if (ess) { ess->r1 = r1; ess->r2 = r2; }
}
int gjx_tile_weights(const float* x, uint64_t n, uint32_t* qw, gjx_tile_rec* recs, gjx_tile_sub* subs, gjx_tile_ess* ess,
                     gjx_stream s) {
  (void)s;
  if (!x || !qw || !recs || !subs || n == 0) return GJX_ERR_INVALID;
  const uint64_t nt = gjx_num_tiles(n);
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < (int64_t)nt; ++b) {
    const uint64_t lo = (uint64_t)b * O_TILE, cnt = lo + O_TILE <= n ? O_TILE : n - lo;
    tile_emit(x + lo, cnt, qw + lo, &recs[b], &subs[b], ess ? &ess[b] : NULL);
  }
  return GJX_OK;
}
/* The merge of a population's records: anchor, every tile's shift, the exclusive prefix of the shifted masses. */
typedef struct {
  int32_t e;
  uint64_t Q, R1, R2;
  uint64_t* pre; /* [nt + 1] */
  uint8_t* d;    /* [nt]: 64 = the tile carries no mass */
} merged;
static inline uint64_t shr64(uint64_t v, int d) { return d >= 64 ? 0 : v >> d; }
static int merge_records(const gjx_tile_rec* recs, const gjx_tile_ess* ess, uint64_t nt, merged* m) {
  m->pre = (uint64_t*)malloc(sizeof(uint64_t) * (nt + 1));
  m->d = (uint8_t*)malloc(nt ? nt : 1);
  if (!m->pre || !m->d) { free(m->pre); free(m->d); return GJX_ERR_LAUNCH; }
  int32_t e = O_ROW_EMPTY;
  for (uint64_t b = 0; b < nt; ++b) e = recs[b].e > e ? recs[b].e : e;
  uint64_t run = 0, r1 = 0, r2 = 0;
  for (uint64_t b = 0; b < nt; ++b) {
    int d = 64;
    if (recs[b].e != O_ROW_EMPTY) {
      const int64_t dd = (int64_t)e - (int64_t)recs[b].e;
      d = dd > 63 ? 64 : (int)dd;
    }
    m->d[b] = (uint8_t)d;
    m->pre[b] = run;
    run += shr64(recs[b].s, d);
    if (ess) { r1 += shr64(ess[b].r1, d); r2 += shr64(ess[b].r2, 2 * d); }
  }
  m->pre[nt] = run;
  m->e = e; m->Q = run; m->R1 = r1; m->R2 = r2;
  return GJX_OK;
}
static void merged_free(merged* m) { free(m->pre); free(m->d); }
int gjx_tile_merge(const gjx_tile_rec* recs, uint64_t n_tiles, int32_t* out_e, uint64_t* out_q, gjx_stream s) {
  (void)s;
  if (!recs || n_tiles == 0) return GJX_ERR_INVALID;
  merged m;
  int rc = merge_records(recs, NULL, n_tiles, &m);
  if (rc) return rc;
  if (out_e) *out_e = m.e;
  if (out_q) *out_q = m.Q;
  merged_free(&m);
  return GJX_OK;
}

/* ---- the comb: the float64 operations every backend evaluates ------------------------------------ */
/* teeth (j + u0), j in [0, n_out), strictly below a position: ceil, clamped to [0, n_out] (a NaN counts as 0) */
static inline int32_t comb_clamp(double t, int32_t n_out) {
  const double c = ceil(t);
  if (!(c > 0.0)) return 0;
  if (c >= (double)n_out) return n_out;
  return (int32_t)c;
}
static inline double comb_base(uint64_t P, double scale, double u0) { return (double)P * scale - u0; }
static inline int32_t comb_tile(uint64_t P, double scale, double u0, int32_t n_out) { return comb_clamp(comb_base(P, scale, u0), n_out); }
static inline int32_t comb_in_tile(double c, double scale_t, double base, int32_t nhi, int32_t n_out) {
  const int32_t t = comb_clamp(fma(c, scale_t, base), n_out);
  return t < nhi ? t : nhi;
}
static inline double comb_tile_scale(double scale, int d) { return d >= 64 ? 0.0 : ldexp(scale, -d); }
static inline double u0_from_bits(uint64_t U) { return (double)(U >> 11) * 0x1.0p-53; }

/* ---- r04: the peer transport (gjx.h: gjx_smc_peers).  The source population is distributed: element i of a per-particle
 * array (tile k of a per-tile array) lives in the arena of rank (i / 1024) / tiles_per_rank (k / tiles_per_rank), at the address
 * it has in this rank's arena plus delta[owner] bytes.  Here the arenas are host memory of one process and the ranks are
 * threads: the wait is a bounded spin on the arrival words, the signal a release store. */
typedef struct {
  const gjx_smc_peers* p; /* NULL: everything is local */
  uint64_t tiles_per_rank;
} peer_view;
static peer_view peer_view_of(const gjx_smc_config* cfg) {
  peer_view v = {cfg->peers, 1};
  if (cfg->peers) v.tiles_per_rank = cfg->n_total / (uint64_t)cfg->peers->world / O_TILE;
  return v;
}
static inline const void* peer_at(const peer_view* v, const void* p, uint64_t particle) {
  if (!v->p) return p;
  return (const char*)p + v->p->delta[(particle / O_TILE) / v->tiles_per_rank];
}
static int peers_ok(const gjx_smc_config* c) {
  const gjx_smc_peers* p = c->peers;
  if (!p || p->world < 2 || p->world > GJX_MAX_PEERS || p->rank < 0 || p->rank >= p->world || !p->flags || !p->error ||
      p->delta[p->rank] != 0 || c->n_filters > 1)
    return 0;
  const uint64_t w = (uint64_t)p->world;
  return c->n_total % (w * O_TILE) == 0 && c->n_local == c->n_total / w && c->first_slot == (uint64_t)p->rank * c->n_local;
}
static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}
/* -> 1 when every arrival word is >= value (then an acquire fence), 0 after the timeout (*error set) */
static int peer_wait_host(const gjx_smc_peers* p, uint64_t value) {
  const double t0 = now_ms(), limit = p->timeout_ms ? (double)p->timeout_ms : 10000.0;
  /* (a wait that has already timed out on this rank is not waited for again: every later wait of the run fails at once) */
  if (__atomic_load_n(p->error, __ATOMIC_RELAXED) != 0u) return 0;
  for (;;) {
    int ready = 1;
    for (int q = 0; q < p->world; ++q) ready = ready && __atomic_load_n(&p->flags[q], __ATOMIC_RELAXED) >= value;
    if (ready) break;
    if (now_ms() - t0 > limit) {
      __atomic_store_n(p->error, 1u, __ATOMIC_RELAXED);
      return 0;
    }
    sched_yield();
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return 1;
}
int gjx_smc_peer_signal_fused(const gjx_smc_config* cfg) {
  /* (as the device library: populations beyond 1024 tiles — there the step's first launch waits for the peers anyway) */
  return cfg && cfg->peers && cfg->n_filters <= 1 && gjx_num_tiles(cfg->n_total) > 1024 ? 1 : 0;
}
int gjx_smc_peer_wait(const gjx_smc_peers* peers, uint64_t value, gjx_stream s) {
  (void)s;
  if (!peers || peers->world < 2 || peers->world > GJX_MAX_PEERS || !peers->flags || !peers->error) return GJX_ERR_INVALID;
  (void)peer_wait_host(peers, value); /* (a timeout is reported through *error, like the device's) */
  return GJX_OK;
}
int gjx_smc_peer_signal(const gjx_smc_peers* peers, const gjx_tile_rec* recs, const gjx_tile_ess* ess, uint64_t first_tile,
                        uint64_t n_tiles, uint64_t value, gjx_stream s) {
  (void)s;
  if (!peers || peers->world < 2 || peers->world > GJX_MAX_PEERS || peers->rank < 0 || peers->rank >= peers->world ||
      !peers->flags || !peers->error || (n_tiles > 0 && !recs))
    return GJX_ERR_INVALID;
  for (int o = 0; o < peers->world; ++o) {
    if (o != peers->rank && recs && n_tiles) {
      memcpy((char*)(uintptr_t)(recs + first_tile) + peers->delta[o], recs + first_tile, sizeof(gjx_tile_rec) * n_tiles);
      if (ess) memcpy((char*)(uintptr_t)(ess + first_tile) + peers->delta[o], ess + first_tile, sizeof(gjx_tile_ess) * n_tiles);
    }
  }
  for (int o = 0; o < peers->world; ++o) {
    uint64_t* word = (uint64_t*)((char*)peers->flags + peers->delta[o]) + peers->rank;
    __atomic_store_n(word, value, __ATOMIC_RELEASE);
  }
  return GJX_OK;
}
int gjx_smc_records_pack(const gjx_smc_config* cfg, int world, int unpack, gjx_tile_rec* recs, gjx_tile_ess* ess, void* stage,
                         gjx_stream s) {
  (void)s;
  if (!cfg || !recs || !ess || !stage || world < 1 || world > 64 || cfg->n_local == 0 || cfg->n_local % O_TILE ||
      cfg->n_total != cfg->n_local * (uint64_t)world || cfg->first_slot % cfg->n_local)
    return GJX_ERR_INVALID;
  const uint64_t tl = cfg->n_local / O_TILE;
  const int rank = (int)(cfg->first_slot / cfg->n_local);
  for (int r = 0; r < world; ++r) {
    if ((r == rank) == (unpack != 0)) continue;
    char* slot = (char*)stage + (size_t)r * tl * 32;
    if (unpack) {
      memcpy(recs + (size_t)r * tl, slot, tl * 16);
      memcpy(ess + (size_t)r * tl, slot + tl * 16, tl * 16);
    } else {
      memcpy(slot, recs + (size_t)r * tl, tl * 16);
      memcpy(slot + tl * 16, ess + (size_t)r * tl, tl * 16);
    }
  }
  return GJX_OK;
}

/* Ancestors of the slots [lo, hi) of an n_out-tooth comb over n particles with stored weights qw and merged records:
 * teeth below the start of tile b: nlo_b = comb_tile(pre[b]); below particle i of tile b (c = running sum of q inside
 * the tile): n_i = min(comb_clamp(fma(c, scale 2^-d_b, base_b)), nlo_{b+1}), a tile's last particle ending at nlo_{b+1},
 * the population's last at n_out; slot j takes the first particle with n_i > j.  Tiles whose slots (known from the
 * merged records alone) all lie outside [lo, hi) are skipped without reading their weights: a rank of a sharded filter
 * holds only the source ranges it needs. */
static void systematic_ancestors(const uint32_t* qw, uint64_t n, const merged* m, uint64_t nt, uint64_t n_out_u,
                                 double u0, int64_t lo, int64_t hi, int32_t* anc /* [hi - lo] */, const peer_view* pv) {
  const int32_t n_out = (int32_t)n_out_u;
  if (m->Q == 0) { /* no mass at all: the population is kept — slot j takes particle floor(j n / n_out) */
    const double ratio = (double)n / (double)n_out_u;
    for (int64_t j = lo; j < hi; ++j) {
      const uint64_t g = (uint64_t)floor((double)j * ratio);
      anc[j - lo] = (int32_t)(g < n ? g : n - 1);
    }
    return;
  }
  const double scale = (double)n_out_u / (double)m->Q;
#pragma omp parallel for schedule(dynamic, 8)
  for (int64_t bb = 0; bb < (int64_t)nt; ++bb) {
    const uint64_t b = (uint64_t)bb;
    const int32_t t_lo = comb_tile(m->pre[b], scale, u0, n_out);
    const int32_t t_hi = b + 1 == nt ? n_out : comb_tile(m->pre[b + 1], scale, u0, n_out);
    if (t_hi <= lo || t_lo >= hi) continue;
    const double scale_t = comb_tile_scale(scale, m->d[b]), base = comb_base(m->pre[b], scale, u0);
    int64_t prev = t_lo;
    const uint64_t i0 = b * O_TILE, i1 = (b + 1) * O_TILE < n ? (b + 1) * O_TILE : n;
    double c = 0.0;
    const uint32_t* qt = pv ? (const uint32_t*)peer_at(pv, qw, i0) : qw; /* (the tile's weights, where they live) */
    for (uint64_t i = i0; i < i1; ++i) {
      c += (double)qt[i];
      const int64_t ni = (i + 1 == i1) ? t_hi : comb_in_tile(c, scale_t, base, t_hi, n_out);
      const int64_t a = prev > lo ? prev : lo, e = ni < hi ? ni : hi;
      for (int64_t j = a; j < e; ++j) anc[j - lo] = (int32_t)i;
      if (ni > prev) prev = ni;
    }
  }
}

int gjx_resample_systematic(const gjx_keys* key, const float* logw, uint64_t n, uint64_t n_out,
                            int32_t* ancestors, int32_t* out_e, uint64_t* out_q, void* ws,
                            size_t ws_bytes, gjx_stream s) {
  (void)ws; (void)ws_bytes; (void)s;
  if (!keys_ok(key) || !logw || !ancestors || n == 0 || n_out == 0 || n > 0x7fffffffull || n_out > 0x7fffffffull) return GJX_ERR_INVALID;
  const uint64_t nt = gjx_num_tiles(n);
  uint32_t* qw = (uint32_t*)malloc(sizeof(uint32_t) * n);
  gjx_tile_rec* recs = (gjx_tile_rec*)malloc(sizeof(gjx_tile_rec) * nt);
  gjx_tile_sub* subs = (gjx_tile_sub*)malloc(sizeof(gjx_tile_sub) * nt);
  merged m;
  int rc = (qw && recs && subs) ? gjx_tile_weights(logw, n, qw, recs, subs, NULL, NULL) : GJX_ERR_LAUNCH;
  if (!rc) rc = merge_records(recs, NULL, nt, &m);
  if (!rc) {
    o_stream st = stream_at(key, 0);
    const double u0 = u0_from_bits(o_bits64_at(&st, 0));
    systematic_ancestors(qw, n, &m, nt, n_out, u0, 0, (int64_t)n_out, ancestors, NULL);
    if (out_e) *out_e = m.e;
    if (out_q) *out_q = m.Q;
    merged_free(&m);
  }
  free(qw); free(recs); free(subs);
  return rc;
}

int gjx_resample_multinomial(const gjx_keys* key, const float* logw, uint64_t n, uint64_t n_out,
                             int32_t* ancestors, float* out_max, uint64_t* out_q, void* ws,
                             size_t ws_bytes, gjx_stream s) {
  (void)ws; (void)ws_bytes; (void)s;
  if (!keys_ok(key) || !logw || !ancestors || n == 0 || n_out == 0) return GJX_ERR_INVALID;
  float m = -INFINITY;
  for (uint64_t i = 0; i < n; ++i) m = logw[i] > m ? logw[i] : m;
  int frac = o_frac_bits(n);
  uint64_t* cdf = (uint64_t*)malloc(sizeof(uint64_t) * n);
  if (!cdf) return GJX_ERR_LAUNCH;
  uint64_t C = 0;
  for (uint64_t i = 0; i < n; ++i) { C += o_fixw(logw[i], m, frac); cdf[i] = C; }
  o_stream st = stream_at(key, 0);
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < (int64_t)n_out; ++j) {
    uint64_t thr = mulhi64(o_bits64_at(&st, (uint32_t)j), C);
    uint64_t lo = 0, hi = n - 1; /* first i with cdf[i] > thr */
    while (lo < hi) {
      uint64_t mid = (lo + hi) >> 1;
      if (cdf[mid] > thr) hi = mid; else lo = mid + 1;
    }
    ancestors[j] = (int32_t)lo;
  }
  free(cdf);
  if (out_max) *out_max = m;
  if (out_q) *out_q = C;
  return GJX_OK;
}

int gjx_gather_cols(const int32_t* ancestors, uint64_t n_out, const void* const* src_cols,
                    void* const* dst_cols, int n_cols, gjx_stream s) {
  (void)s;
  if (!ancestors || !src_cols || !dst_cols || n_cols < 0) return GJX_ERR_INVALID;
  for (int c = 0; c < n_cols; ++c) {
    const uint32_t* src = (const uint32_t*)src_cols[c];
    uint32_t* dst = (uint32_t*)dst_cols[c];
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < (int64_t)n_out; ++j) dst[j] = src[ancestors[j]];
  }
  return GJX_OK;
}

/* ---- fused bootstrap SMC ------------------------------------------------------------------------ */
static int cfg_ok(const gjx_smc_config* c) {
  return c && (c->impl == 0 || c->impl == 1) && c->n_total > 0 && c->n_local > 0 &&
         c->first_slot + c->n_local <= c->n_total && c->n_steps > 0 && c->step_keys &&
         c->resample_keys && (c->first_slot % O_TILE) == 0 && !(c->ess_threshold < 0.0f);
}
/* ESS-adaptive resampling (gjx.h: gjx_smc_config.ess_threshold).  The decision is a function of exact integer sums
 * and three double operations, so every backend and every sharding takes the same one. */
static int cfg_adaptive(const gjx_smc_config* c) { return c->ess_threshold > 0.0f && c->ess_threshold < 1.0f; }
static int ess_says_resample(uint64_t r1, uint64_t r2, double thr) {
  if (!(thr > 0.0) || r2 == 0) return 1;
  const double a = (double)r1 * (double)r1;
  const double b = thr * (double)r2;
  return a < b;
}
static int pop_ok(const gjx_smc_pop* p, int n_state, int adaptive) {
  if (!p || !p->qw || !p->recs || !p->subs) return 0;
  for (int k = 0; k < n_state; ++k)
    if (!p->state[k]) return 0;
  return !adaptive || (p->logw && p->ess);
}

int gjx_smc_finish(const gjx_smc_config* cfg, const gjx_tile_rec* recs, int32_t* e_out, uint64_t* q_out, gjx_stream s) {
  if (!cfg_ok(cfg) || !recs || cfg->n_filters > 1) return GJX_ERR_INVALID;
  return gjx_tile_merge(recs, gjx_num_tiles(cfg->n_total), e_out, q_out, s);
}

/* Source-tile ranges of `world` equal blocks of output slots: tile b can own slots in [ceil(P_b) - 1,
   ceil(P_{b+1})) for some comb offset u0 in [0, 1) (teeth_below above; P = prefix * N / Q in double), the last
   tile up to N. */
int gjx_smc_source_ranges(const gjx_smc_config* cfg, const gjx_tile_rec* recs, const gjx_tile_ess* ess, int world,
                          int64_t ticket, int64_t* out_ranges, gjx_stream s) {
  (void)s;
  if (!cfg_ok(cfg) || !recs || !out_ranges || world < 1 || world > 64 || cfg->n_total % (uint64_t)world)
    return GJX_ERR_INVALID;
  if (cfg_adaptive(cfg) && !ess) return GJX_ERR_INVALID;
  const uint64_t N = cfg->n_total, nt = gjx_num_tiles(N), nl = N / (uint64_t)world;
  merged m;
  int rc = merge_records(recs, cfg_adaptive(cfg) ? ess : NULL, nt, &m);
  if (rc) return rc;
  if (cfg_adaptive(cfg) && !ess_says_resample(m.R1, m.R2, (double)cfg->ess_threshold * (double)N)) {
    /* the next step keeps its particles: every block's sources are its own tiles */
    for (int j = 0; j < world; ++j) {
      out_ranges[2 * j] = (int64_t)((uint64_t)j * (nl / O_TILE));
      out_ranges[2 * j + 1] = (int64_t)((uint64_t)(j + 1) * (nl / O_TILE));
    }
    out_ranges[2 * world] = ticket;
    merged_free(&m);
    return GJX_OK;
  }
  const uint64_t Q = m.Q;
  const double scale = (double)N / (double)Q, nd = (double)N;
  for (int j = 0; j < world; ++j) {
    const double lo = (double)((uint64_t)j * nl), hi = (double)((uint64_t)(j + 1) * nl);
    int64_t first = 0, end = 0;
    for (uint64_t b = 0; b < nt; ++b) {
      double lower = ceil((double)m.pre[b] * scale);
      if (!(lower < nd)) lower = nd;
      lower = lower - 1.0 > 0.0 ? lower - 1.0 : 0.0;
      double upper = ceil((double)m.pre[b + 1] * scale);
      if (b + 1 == nt || !(upper < nd)) upper = nd;
      first += upper <= lo;
      end += lower < hi;
    }
    /* no mass at all: the population is kept, every block's sources are its own tiles */
    out_ranges[2 * j] = Q == 0 ? (int64_t)((uint64_t)j * (nl / O_TILE)) : first;
    out_ranges[2 * j + 1] = Q == 0 ? (int64_t)((uint64_t)(j + 1) * (nl / O_TILE)) : end;
  }
  out_ranges[2 * world] = ticket;
  merged_free(&m);
  return GJX_OK;
}

/* Front half of every step (t >= 1): the merge of the previous population's records (its anchor and total mass go to
 * prev_e_out / prev_q_out), the step's flag, and the ancestors of the rank's slots — by systematic resampling, or the
 * identity when an adaptive filter keeps its particles.
 * Returns 1 if the step resamples (the new log-weights start from 0), 0 if it accumulates, < 0 on error. */
static int smc_step_front(const gjx_smc_config* cfg, int t, const gjx_smc_pop* prev, int32_t* prev_e_out,
                          uint64_t* prev_q_out, int32_t* anc) {
  const uint64_t N = cfg->n_total, nt = gjx_num_tiles(N);
  const int ad = cfg_adaptive(cfg);
  merged m;
  int rc = merge_records(prev->recs, ad ? prev->ess : NULL, nt, &m);
  if (rc) return rc;
  if (prev_e_out) *prev_e_out = m.e;
  if (prev_q_out) *prev_q_out = m.Q;
  const int res = !ad || ess_says_resample(m.R1, m.R2, (double)cfg->ess_threshold * (double)N);
  if (cfg->resampled_out && !(cfg->n_filters > 1)) cfg->resampled_out[t] = res;
  if (res) {
    const uint32_t rkey[4] = {cfg->resample_keys[2 * t], cfg->resample_keys[2 * t + 1], 0u, 0u};
    o_stream st = o_stream_make(cfg->impl, rkey, 0, 0);
    const double u0 = u0_from_bits(o_bits64_at(&st, 0));
    const peer_view pv = peer_view_of(cfg);
    systematic_ancestors(prev->qw, N, &m, nt, N, u0, (int64_t)cfg->first_slot, (int64_t)(cfg->first_slot + cfg->n_local), anc, &pv);
  } else {
    for (uint64_t j = 0; j < cfg->n_local; ++j) anc[j] = (int32_t)(cfg->first_slot + j);
  }
  merged_free(&m);
  return res;
}
/* Back half of every step: the fixed-point weights and records of the rank's new log-weights lw[n_local]. */
static void smc_step_back(const gjx_smc_config* cfg, const gjx_smc_pop* out, const float* lw) {
  const uint64_t nl = cfg->n_local, tile0 = cfg->first_slot / O_TILE, ntl = gjx_num_tiles(nl);
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < (int64_t)ntl; ++b) {
    const uint64_t lo = (uint64_t)b * O_TILE, cnt = lo + O_TILE <= nl ? O_TILE : nl - lo;
    tile_emit(lw + lo, cnt, out->qw + lo, &out->recs[tile0 + (uint64_t)b], &out->subs[tile0 + (uint64_t)b],
              cfg_adaptive(cfg) ? &out->ess[tile0 + (uint64_t)b] : NULL);
  }
  if (out->logw) memcpy(out->logw, lw, sizeof(float) * nl);
}

/* One step of a filter.  propagate(ctx, j, anc or -1, &w): writes slot j's new state, returns its log-weight increment. */
typedef float (*propagate_fn)(void* ctx, uint64_t j_local, int64_t ancestor);
static int smc_step_generic(const gjx_smc_config* cfg, int t, int n_state, const gjx_smc_pop* prev, const gjx_smc_pop* out,
                            int32_t* prev_e_out, uint64_t* prev_q_out, int32_t* ancestors_out, propagate_fn fn, void* ctx) {
  const int ad = cfg_adaptive(cfg);
  if (!pop_ok(out, n_state, ad) || (t > 0 && (!pop_ok(prev, n_state, ad) || prev->recs == out->recs))) return GJX_ERR_INVALID;
  if (cfg->peers) { /* the peer transport: nothing of the source population is read before every peer has arrived */
    if (!peers_ok(cfg)) return GJX_ERR_INVALID;
    if (cfg->peers->signal_value != 0) { /* a deferred signal of the previous step goes out first (gjx.h) */
      const int rc = gjx_smc_peer_signal(cfg->peers, cfg->peers->signal_recs, cfg->peers->signal_ess, cfg->peers->signal_first_tile,
                                         cfg->peers->signal_n_tiles, cfg->peers->signal_value, NULL);
      if (rc) return rc;
    }
    if (t > 0 && !peer_wait_host(cfg->peers, cfg->peers->wait_value)) return GJX_ERR_LAUNCH;
  }
  const uint64_t nl = cfg->n_local;
  int32_t* anc = NULL;
  int carry = 0;
  if (t > 0) {
    anc = ancestors_out ? ancestors_out : (int32_t*)malloc(sizeof(int32_t) * nl);
    if (!anc) return GJX_ERR_LAUNCH;
    const int res = smc_step_front(cfg, t, prev, prev_e_out, prev_q_out, anc);
    if (res < 0) {
      if (anc != ancestors_out) free(anc);
      return res;
    }
    carry = !res;
  } else if (ancestors_out) {
    for (uint64_t j = 0; j < nl; ++j) ancestors_out[j] = (int32_t)(cfg->first_slot + j);
  }
  float* lw = (float*)malloc(sizeof(float) * nl);
  if (!lw) {
    if (anc && anc != ancestors_out) free(anc);
    return GJX_ERR_LAUNCH;
  }
#pragma omp parallel for schedule(static)
  for (int64_t j = 0; j < (int64_t)nl; ++j) {
    float w = fn(ctx, (uint64_t)j, t > 0 ? (int64_t)anc[j] : -1);
    if (carry) w = w + prev->logw[anc[j]]; /* no resampling at this step: the log-weight accumulates */
    lw[j] = w;
  }
  smc_step_back(cfg, out, lw);
  free(lw);
  if (anc && anc != ancestors_out) free(anc);
  return GJX_OK;
}

typedef struct {
  const gjx_smc_config* cfg;
  const gjx_lgssm* mdl;
  int t;
  float y;
  const float* prev_x;
  float* x_out;
  uint32_t skey[4];
} lgssm_ctx;
static float lgssm_propagate(void* vc, uint64_t j, int64_t a) {
  lgssm_ctx* c = (lgssm_ctx*)vc;
  const float eps = o_smc_slot_normal(c->cfg->impl, c->skey, c->cfg->first_slot + j);
  float x;
  if (a < 0) {
    const float tt = c->mdl->x0_scale * eps;
    x = c->mdl->x0_loc + tt;
  } else {
    const peer_view pv = peer_view_of(c->cfg);
    const float mean = c->mdl->a * *(const float*)peer_at(&pv, c->prev_x + a, (uint64_t)a);
    const float tt = c->mdl->q * eps;
    x = mean + tt;
  }
  c->x_out[j] = x;
  return o_logpdf_normal(c->y, x, c->mdl->r);
}
int gjx_smc_lgssm_step(const gjx_smc_config* cfg, const gjx_lgssm* mdl, int t, float y_t, const gjx_smc_pop* prev,
                       const gjx_smc_pop* out, int32_t* prev_e_out, uint64_t* prev_q_out, int32_t* ancestors_out,
                       gjx_stream s) {
  (void)s;
  if (!cfg_ok(cfg) || !mdl || t < 0 || t >= cfg->n_steps || !out || (t > 0 && !prev) || cfg->n_filters > 1) return GJX_ERR_INVALID;
  lgssm_ctx c = {cfg, mdl, t, y_t, t > 0 ? (const float*)prev->state[0] : NULL, (float*)out->state[0],
                 {cfg->step_keys[2 * t], cfg->step_keys[2 * t + 1], 0u, 0u}};
  return smc_step_generic(cfg, t, 1, prev, out, prev_e_out, prev_q_out, ancestors_out, lgssm_propagate, &c);
}

/* HMM transition tables as ALIAS tables: one 4-byte table word per draw.  Row r of the table
 * holds K packed entries (threshold24 << 8) | alias.  Construction, all in integers: p_c = cat_fix(l_c, max l),
 * Q = sum p_c, scaled_c = p_c * K; columns with scaled < Q queue up as "small", the others as "large", both in
 * increasing column order; repeatedly the front small column s is paired with the front large column g
 * (entry s: accept mass scaled_s, alias g; g gives up Q - scaled_s and moves to the BACK of the small queue
 * once it falls below Q); columns left over accept always.  threshold24 = floor(accept * 2^24 / Q). */
uint64_t gjx_hmm_alias_words(int32_t n_states) {
  return n_states > 0 ? (uint64_t)n_states * (uint64_t)n_states : 0;
}
static void hmm_alias_row(const float* l, uint32_t K, uint32_t* row) {
  uint64_t scaled[256];
  uint16_t small[512], large[256]; /* queues: a column enters `small` at most twice (once from `large`) */
  uint32_t hs = 0, ts = 0, hl = 0, tl = 0;
  const float m = row_max(l, K);
  uint64_t Q = 0;
  for (uint32_t c = 0; c < K; ++c) { scaled[c] = (uint64_t)cat_fix(l[c], m); Q += scaled[c]; }
  for (uint32_t c = 0; c < K; ++c) {
    scaled[c] *= K;
    if (scaled[c] < Q) small[ts++] = (uint16_t)c; else large[tl++] = (uint16_t)c;
  }
  for (uint32_t c = 0; c < K; ++c) row[c] = (0xffffffu << 8) | c; /* accept always */
  while (hs < ts && hl < tl) {
    const uint32_t sc = small[hs++], g = large[hl];
    row[sc] = ((uint32_t)((scaled[sc] << 24) / Q) << 8) | g;
    scaled[g] -= Q - scaled[sc];
    if (scaled[g] < Q) { ++hl; small[ts++] = (uint16_t)g; }
  }
}
/* The state drawn from row `row` with 32 random bits: column = floor(bits K / 2^32), the next 24 bits of the
 * product decide between the column and its alias. */
static inline uint32_t hmm_alias_draw(const uint32_t* row, uint32_t K, uint32_t bits) {
  const uint64_t t = (uint64_t)bits * (uint64_t)K;
  const uint32_t col = (uint32_t)(t >> 32), f24 = (uint32_t)t >> 8;
  const uint32_t e = row[col];
  return f24 < (e >> 8) ? col : (e & 255u);
}
int gjx_hmm_prepare(const gjx_hmm* mdl, uint32_t* trans_alias, float* obs_logp, gjx_stream s) {
  (void)s;
  if (!mdl || !trans_alias || !obs_logp || mdl->n_states <= 0 || mdl->n_states > 256 ||
      !mdl->trans_logits || !mdl->obs_logits)
    return GJX_ERR_INVALID;
  const uint32_t K = (uint32_t)mdl->n_states;
  for (uint32_t r = 0; r < K; ++r) {
    hmm_alias_row(mdl->trans_logits + (size_t)r * K, K, trans_alias + (size_t)r * K);
    const float* o = mdl->obs_logits + (size_t)r * K;
    float lse = row_lse(o, K);
    for (uint32_t c = 0; c < K; ++c) obs_logp[(size_t)r * K + c] = o[c] - lse;
  }
  return GJX_OK;
}

typedef struct {
  const gjx_smc_config* cfg;
  const gjx_hmm* mdl;
  int32_t y;
  const int32_t* prev_z;
  int32_t* z_out;
  const uint32_t* trans_alias;
  const float* obs_logp;
  uint32_t skey[4];
} hmm_ctx;
static float hmm_propagate(void* vc, uint64_t j, int64_t a) {
  hmm_ctx* c = (hmm_ctx*)vc;
  const uint32_t K = (uint32_t)c->mdl->n_states;
  const uint32_t bits = o_smc_slot_bits(c->cfg->impl, c->skey, c->cfg->first_slot + j);
  const peer_view pv = peer_view_of(c->cfg);
  const int32_t zp = a < 0 ? c->mdl->init_state : *(const int32_t*)peer_at(&pv, c->prev_z + a, (uint64_t)a);
  const uint32_t z = hmm_alias_draw(c->trans_alias + (size_t)zp * K, K, bits);
  c->z_out[j] = (int32_t)z;
  return c->obs_logp[(size_t)z * K + (uint32_t)c->y];
}
int gjx_smc_hmm_step(const gjx_smc_config* cfg, const gjx_hmm* mdl, int t, int32_t y_t, const gjx_smc_pop* prev,
                     const gjx_smc_pop* out, int32_t* prev_e_out, uint64_t* prev_q_out, const uint32_t* trans_alias,
                     const float* obs_logp, int32_t* ancestors_out, gjx_stream s) {
  (void)s;
  if (!cfg_ok(cfg) || !mdl || t < 0 || t >= cfg->n_steps || !out || (t > 0 && !prev) || !trans_alias || !obs_logp ||
      y_t < 0 || y_t >= mdl->n_states || cfg->n_filters > 1)
    return GJX_ERR_INVALID;
  hmm_ctx c = {cfg, mdl, y_t, t > 0 ? (const int32_t*)prev->state[0] : NULL, (int32_t*)out->state[0], trans_alias, obs_logp,
               {cfg->step_keys[2 * t], cfg->step_keys[2 * t + 1], 0u, 0u}};
  return smc_step_generic(cfg, t, 1, prev, out, prev_e_out, prev_q_out, ancestors_out, hmm_propagate, &c);
}

/* Whole single-device runs: the straightforward T-loop over the steps, on two populations of the run's own. */
typedef int (*step_fn)(void* ctx, const gjx_smc_config* cfg, int t, const gjx_smc_pop* prev, const gjx_smc_pop* out,
                       int32_t* pe, uint64_t* pq, int32_t* anc);
static int pop_alloc(gjx_smc_pop* p, uint64_t N, int n_state, int adaptive) {
  memset(p, 0, sizeof *p);
  const uint64_t nt = gjx_num_tiles(N);
  int ok = 1;
  for (int k = 0; k < n_state; ++k) ok = ok && (p->state[k] = malloc(4 * N)) != NULL;
  ok = ok && (p->qw = (uint32_t*)malloc(4 * N)) != NULL;
  ok = ok && (p->logw = (float*)malloc(4 * N)) != NULL;
  ok = ok && (p->recs = (gjx_tile_rec*)malloc(sizeof(gjx_tile_rec) * nt)) != NULL;
  ok = ok && (p->subs = (gjx_tile_sub*)malloc(sizeof(gjx_tile_sub) * nt)) != NULL;
  if (adaptive) ok = ok && (p->ess = (gjx_tile_ess*)calloc(nt, sizeof(gjx_tile_ess))) != NULL;
  return ok ? GJX_OK : GJX_ERR_LAUNCH;
}
static void pop_free(gjx_smc_pop* p) {
  for (int k = 0; k < GJX_SMC_MAX_STATE; ++k) free(p->state[k]);
  free(p->qw); free(p->logw); free(p->recs); free(p->subs); free(p->ess);
}
static int smc_run_one(const gjx_smc_config* cfg, int n_state, step_fn step, void* ctx, int32_t* out_e, uint64_t* out_q,
                       void* const* state_out, float* logw_out, int32_t* ancestors_out) {
  if (!cfg_ok(cfg) || cfg->first_slot != 0 || cfg->n_local != cfg->n_total || !out_e || !out_q || !state_out || !logw_out)
    return GJX_ERR_INVALID;
  const uint64_t N = cfg->n_total;
  const int ad = cfg_adaptive(cfg), T = cfg->n_steps;
  if (ad && !cfg->resampled_out) return GJX_ERR_INVALID;
  if (cfg->resampled_out) memset(cfg->resampled_out, 0, sizeof(int32_t) * (size_t)T);
  gjx_smc_pop pop[2];
  int rc = pop_alloc(&pop[0], N, n_state, ad);
  if (!rc) rc = pop_alloc(&pop[1], N, n_state, ad);
  for (int t = 0; t < T && rc == GJX_OK; ++t) {
    const int cur = t & 1, prv = cur ^ 1;
    rc = step(ctx, cfg, t, &pop[prv], &pop[cur], t ? &out_e[t - 1] : NULL, t ? &out_q[t - 1] : NULL,
              ancestors_out ? ancestors_out + (size_t)t * N : NULL);
  }
  const int last = (T - 1) & 1;
  if (rc == GJX_OK) rc = gjx_smc_finish(cfg, pop[last].recs, &out_e[T - 1], &out_q[T - 1], NULL);
  if (rc == GJX_OK) {
    for (int k = 0; k < n_state; ++k) memcpy(state_out[k], pop[last].state[k], 4 * N);
    memcpy(logw_out, pop[last].logw, 4 * N);
  }
  pop_free(&pop[0]);
  pop_free(&pop[1]);
  return rc;
}
/* Several filters (gjx_smc_config.n_filters): by definition, each filter's own single run, with its keys
 * [f, T, 2] and its slice of every output. */
static int smc_run_filters(const gjx_smc_config* cfg, int n_state, step_fn step, void* ctx, int32_t* out_e, uint64_t* out_q,
                           void* const* state_out, float* logw_out, int32_t* ancestors_out) {
  if (!cfg || !state_out) return GJX_ERR_INVALID;
  if (cfg->n_filters <= 1) return smc_run_one(cfg, n_state, step, ctx, out_e, out_q, state_out, logw_out, ancestors_out);
  if (!cfg_ok(cfg) || !out_e || !out_q || !logw_out) return GJX_ERR_INVALID;
  const int F = cfg->n_filters, T = cfg->n_steps;
  const uint64_t N = cfg->n_total, stride = cfg->filter_stride;
  if (F > 16 || stride < N) return GJX_ERR_INVALID;
  int32_t* anc1 = ancestors_out ? (int32_t*)malloc(sizeof(int32_t) * (size_t)T * N) : NULL;
  int rc = GJX_OK;
  for (int f = 0; f < F && !rc; ++f) {
    gjx_smc_config c = *cfg;
    c.n_filters = 0;
    c.step_keys = cfg->step_keys + 2 * (size_t)f * T;
    c.resample_keys = cfg->resample_keys + 2 * (size_t)f * T;
    c.resampled_out = cfg->resampled_out ? cfg->resampled_out + (size_t)f * T : NULL;
    void* cols[GJX_SMC_MAX_STATE];
    for (int k = 0; k < n_state; ++k) cols[k] = (char*)state_out[k] + 4 * (size_t)f * stride;
    rc = smc_run_one(&c, n_state, step, ctx, out_e + (size_t)f * T, out_q + (size_t)f * T, cols,
                     logw_out + (size_t)f * stride, anc1);
    if (!rc && anc1)  /* [T, N] -> [T, F, stride] */
      for (int t = 0; t < T; ++t)
        memcpy(ancestors_out + ((size_t)t * F + f) * stride, anc1 + (size_t)t * N, sizeof(int32_t) * N);
  }
  free(anc1);
  return rc;
}
typedef struct { const gjx_lgssm* mdl; const float* y; } lgssm_run;
static int lgssm_run_step(void* vc, const gjx_smc_config* cfg, int t, const gjx_smc_pop* prev, const gjx_smc_pop* out,
                          int32_t* pe, uint64_t* pq, int32_t* anc) {
  lgssm_run* r = (lgssm_run*)vc;
  return gjx_smc_lgssm_step(cfg, r->mdl, t, r->y[t], prev, out, pe, pq, anc, NULL);
}
int gjx_smc_run_lgssm(const gjx_smc_config* cfg, const gjx_lgssm* model, const float* y,
                      int32_t* out_e, uint64_t* out_q, float* state_out, float* logw_out,
                      int32_t* ancestors_out, void* ws, size_t ws_bytes, gjx_stream s) {
  (void)ws; (void)ws_bytes; (void)s;
  if (!model || !y || !state_out) return GJX_ERR_INVALID;
  lgssm_run r = {model, y};
  void* st[1] = {state_out};
  return smc_run_filters(cfg, 1, lgssm_run_step, &r, out_e, out_q, st, logw_out, ancestors_out);
}
typedef struct { const gjx_hmm* mdl; const int32_t* y; uint32_t* tcdf; float* ologp; } hmm_run;
static int hmm_run_step(void* vc, const gjx_smc_config* cfg, int t, const gjx_smc_pop* prev, const gjx_smc_pop* out,
                        int32_t* pe, uint64_t* pq, int32_t* anc) {
  hmm_run* r = (hmm_run*)vc;
  return gjx_smc_hmm_step(cfg, r->mdl, t, r->y[t], prev, out, pe, pq, r->tcdf, r->ologp, anc, NULL);
}
int gjx_smc_run_hmm(const gjx_smc_config* cfg, const gjx_hmm* model, const int32_t* y,
                    int32_t* out_e, uint64_t* out_q, int32_t* state_out, float* logw_out,
                    int32_t* ancestors_out, void* ws, size_t ws_bytes, gjx_stream s) {
  (void)ws; (void)ws_bytes; (void)s;
  if (!model || !y || !state_out || model->n_states <= 0 || model->n_states > 256) return GJX_ERR_INVALID;
  const size_t kk = (size_t)model->n_states * (size_t)model->n_states;
  hmm_run r = {model, y, (uint32_t*)malloc(4 * (size_t)gjx_hmm_alias_words(model->n_states)), (float*)malloc(4 * kk)};
  int rc = (r.tcdf && r.ologp) ? gjx_hmm_prepare(model, r.tcdf, r.ologp, NULL) : GJX_ERR_LAUNCH;
  void* st[1] = {state_out};
  if (!rc) rc = smc_run_filters(cfg, 1, hmm_run_step, &r, out_e, out_q, st, logw_out, ancestors_out);
  free(r.tcdf); free(r.ologp);
  return rc;
}

/* ---- oracle-only probes (not part of include/gjx.h): raw ciphers and math-spec functions, so the
 * tests can pin them against Random123 known answers and scipy float64. ------------------------ */
void gjo_threefry2x32(const uint32_t key[2], const uint32_t ctr[2], uint32_t out[2]) {
  o_threefry2x32(key[0], key[1], ctr[0], ctr[1], &out[0], &out[1]);
}
void gjo_philox4x32(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]) {
  o_philox4x32(key[0], key[1], ctr, out);
}
/* fn: 0 log, 1 exp, 2 erfinv, 3 lgamma, 4 std_normal(bits as float bit pattern), 5 uniform01 */
void gjo_math(int fn, const float* x, float* y, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) {
    switch (fn) {
      case 0: y[i] = o_log(x[i]); break;
      case 1: y[i] = o_exp(x[i]); break;
      case 2: y[i] = o_erfinv(x[i]); break;
      case 3: y[i] = o_lgamma(x[i]); break;
      case 4: y[i] = o_std_normal(o_f2u(x[i])); break;
      default: y[i] = o_uniform01(o_f2u(x[i])); break;
    }
  }
}

/* the Box-Muller transform of two draw words (tests: accuracy against float64) */
void gjo_bm_pair(const uint32_t* w_radius, const uint32_t* w_angle, float* z_cos, float* z_sin, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) o_bm_pair(w_radius[i], w_angle[i], &z_cos[i], &z_sin[i]);
}

/* ---- bootstrap SMC for a user model (init + step site tables) ------------------------------------ */
/* programs of the state arguments (init_state / next_state): the plan's own copies */
typedef struct { gjx_expr_op ops[GJX_SMC_MAX_STATE][GJX_MAX_EXPR_OPS]; } state_expr_store;
static void state_expr_adopt(gjx_arg* args, int n, state_expr_store* st) {
  for (int k = 0; k < n; ++k)
    if (args[k].kind == GJX_ARG_EXPR) {
      memcpy(st->ops[k], (const void*)args[k].table, sizeof(gjx_expr_op) * (size_t)args[k].ref);
      args[k].table = (const float*)(const void*)st->ops[k];
    }
}
struct gjx_smc_plan {
  gjx_smc_model m;
  gjx_site init_sites[GJX_MAX_SITES];
  gjx_site step_sites[GJX_MAX_SITES];
  expr_store init_expr, step_expr;
  state_expr_store init_state_expr, next_state_expr;
  scope_info init_scopes, step_scopes;
};

static int smc_arg_ok(const gjx_arg* a, int s, int n_state, int n_obs, int allow_state) {
  switch (a->kind) {
    case GJX_ARG_CONST: return 1;
    case GJX_ARG_SITE: return a->ref >= 0 && a->ref < s;
    case GJX_ARG_TABLE: return a->ref >= 0 && a->ref < s && a->table != NULL;
    case GJX_ARG_STATE: return allow_state && a->ref >= 0 && a->ref < n_state;
    case GJX_ARG_OBS: return a->ref >= 0 && a->ref < n_obs;
    case GJX_ARG_EXPR: return expr_ok(a, s, n_state, n_obs, allow_state);
    default: return 0;
  }
}
static int smc_sites_ok(const gjx_site* sites, int n, int n_state, int n_obs, int allow_state) {
  if (!sites || n <= 0 || n > GJX_MAX_SITES) return 0;
  for (int s = 0; s < n; ++s) {
    const gjx_site* st = &sites[s];
    if (st->dist < 0 || st->dist > GJX_DIST_CATEGORICAL) return 0;
    if (!smc_arg_ok(&st->arg[0], s, n_state, n_obs, allow_state)) return 0;
    if (st->dist != GJX_DIST_BERNOULLI && st->dist != GJX_DIST_CATEGORICAL &&
        !smc_arg_ok(&st->arg[1], s, n_state, n_obs, allow_state))
      return 0;
    if (st->observed && !(st->obs.kind == GJX_ARG_CONST || (st->obs.kind == GJX_ARG_OBS && st->obs.ref >= 0 && st->obs.ref < n_obs)))
      return 0;
    if (st->dist == GJX_DIST_CATEGORICAL &&
        (!st->logits || st->n_cat <= 0 || st->n_rows <= 0 || (st->cat_mode != 0 && st->cat_mode != 1) ||
         st->arg[0].kind == GJX_ARG_EXPR))
      return 0;
  }
  return 1;
}

int gjx_smc_plan_create(const gjx_smc_model* m, gjx_smc_plan** out) {
  if (!m || !out || m->n_state < 1 || m->n_state > GJX_SMC_MAX_STATE || m->n_obs < 0 || m->n_obs > GJX_SMC_MAX_OBS)
    return GJX_ERR_INVALID;
  if (!smc_sites_ok(m->init_sites, m->n_init_sites, m->n_state, m->n_obs, 0)) return GJX_ERR_INVALID;
  if (!smc_sites_ok(m->step_sites, m->n_step_sites, m->n_state, m->n_obs, 1)) return GJX_ERR_INVALID;
  for (int k = 0; k < m->n_state; ++k) {
    if (!smc_arg_ok(&m->init_state[k], m->n_init_sites, m->n_state, m->n_obs, 0) || m->init_state[k].kind == GJX_ARG_TABLE)
      return GJX_ERR_INVALID;
    if (!smc_arg_ok(&m->next_state[k], m->n_step_sites, m->n_state, m->n_obs, 1) || m->next_state[k].kind == GJX_ARG_TABLE)
      return GJX_ERR_INVALID;
  }
  gjx_smc_plan* p = (gjx_smc_plan*)malloc(sizeof(gjx_smc_plan));
  if (!p) return GJX_ERR_LAUNCH;
  p->m = *m;
  memcpy(p->init_sites, m->init_sites, sizeof(gjx_site) * (size_t)m->n_init_sites);
  memcpy(p->step_sites, m->step_sites, sizeof(gjx_site) * (size_t)m->n_step_sites);
  expr_adopt(p->init_sites, m->n_init_sites, &p->init_expr);
  expr_adopt(p->step_sites, m->n_step_sites, &p->step_expr);
  state_expr_adopt(p->m.init_state, m->n_state, &p->init_state_expr);
  state_expr_adopt(p->m.next_state, m->n_state, &p->next_state_expr);
  p->m.init_sites = p->init_sites;
  p->m.step_sites = p->step_sites;
  p->init_scopes.n_scopes = 0;
  p->step_scopes.n_scopes = 0;
  *out = p;
  return GJX_OK;
}
int gjx_smc_plan_destroy(gjx_smc_plan* p) { free(p); return GJX_OK; }
int gjx_smc_plan_create_scoped(const gjx_smc_model* m, const gjx_scope* init_scopes, int n_init_scopes,
                               const gjx_scope* step_scopes, int n_step_scopes, gjx_smc_plan** out) {
  gjx_smc_plan* p = NULL;
  int rc = gjx_smc_plan_create(m, &p);
  if (rc) return rc;
  if (!derive_scopes(p->init_sites, p->m.n_init_sites, init_scopes, n_init_scopes, &p->init_scopes) ||
      !derive_scopes(p->step_sites, p->m.n_step_sites, step_scopes, n_step_scopes, &p->step_scopes)) {
    gjx_smc_plan_destroy(p);
    return GJX_ERR_INVALID;
  }
  *out = p;
  return GJX_OK;
}
int gjo_smc_plan_dims(const gjx_smc_plan* p, int* n_state, int* n_obs) { /* for gjx_oracle_comm.cpp (the plan is opaque there) */
  if (!p) return GJX_ERR_INVALID;
  *n_state = p->m.n_state; *n_obs = p->m.n_obs;
  return GJX_OK;
}
int gjx_smc_plan_compile_check(const gjx_smc_plan* p, int impl) { (void)p; (void)impl; return GJX_ERR_UNSUPPORTED; }

/* ---- importance over a Scan model (scan.py:237-294): per particle, T steps with the chained key ---- */
struct gjx_scan_plan {
  gjx_scan_model m;
  gjx_site step_sites[GJX_MAX_SITES];
  expr_store step_expr;
  state_expr_store next_state_expr;
  scope_info scopes;
};
int gjx_scan_plan_create(const gjx_scan_model* m, uint32_t flags, gjx_scan_plan** out) {
  if (!m || !out || (flags & ~(uint32_t)GJX_PLAN_FAST_MATH) || m->n_state < 1 || m->n_state > GJX_SMC_MAX_STATE ||
      m->n_obs < 0 || m->n_obs > GJX_SMC_MAX_OBS)
    return GJX_ERR_INVALID;
  if (!smc_sites_ok(m->step_sites, m->n_step_sites, m->n_state, m->n_obs, 1)) return GJX_ERR_INVALID;
  for (int k = 0; k < m->n_state; ++k)
    if (!smc_arg_ok(&m->next_state[k], m->n_step_sites, m->n_state, m->n_obs, 1) || m->next_state[k].kind == GJX_ARG_TABLE)
      return GJX_ERR_INVALID;
  gjx_scan_plan* p = (gjx_scan_plan*)malloc(sizeof(gjx_scan_plan));
  if (!p) return GJX_ERR_LAUNCH;
  p->m = *m;
  memcpy(p->step_sites, m->step_sites, sizeof(gjx_site) * (size_t)m->n_step_sites);
  expr_adopt(p->step_sites, m->n_step_sites, &p->step_expr);
  state_expr_adopt(p->m.next_state, m->n_state, &p->next_state_expr);
  p->m.step_sites = p->step_sites;
  p->scopes.n_scopes = 0;
  *out = p;
  return GJX_OK;
}
int gjx_scan_plan_destroy(gjx_scan_plan* p) { free(p); return GJX_OK; }
int gjx_scan_plan_create_scoped(const gjx_scan_model* m, const gjx_scope* scopes, int n_scopes, uint32_t flags,
                                gjx_scan_plan** out) {
  gjx_scan_plan* p = NULL;
  int rc = gjx_scan_plan_create(m, flags, &p);
  if (rc) return rc;
  if (!derive_scopes(p->step_sites, p->m.n_step_sites, scopes, n_scopes, &p->scopes)) {
    gjx_scan_plan_destroy(p);
    return GJX_ERR_INVALID;
  }
  *out = p;
  return GJX_OK;
}
int gjx_scan_plan_compile_check(const gjx_scan_plan* p, int impl) { (void)p; (void)impl; return GJX_ERR_UNSUPPORTED; }
int gjx_scan_run(gjx_scan_plan* p, const gjx_scan_io* io, gjx_stream s) {
  if (!p || !io || !keys_ok(io->particle_keys) || io->particle_keys->has_fold || !io->logw || io->n_steps < 1 ||
      io->col_stride < io->n || (p->m.n_obs > 0 && !io->obs) || !io->carry0 || io->n_value_cols < 0 ||
      io->n_value_cols > GJX_MAX_SITES || ((io->row_e == NULL) != (io->row_s == NULL)) ||
      (io->lse && (!io->row_e || !io->lse->tickets)))
    return GJX_ERR_INVALID;
  /* PHILOX step keys put t + 1 above bit 40 of the lane */
  if (io->particle_keys->impl == 1 &&
      (io->n_steps >= (1 << 24) - 1 || (io->particle_keys->mode == 1 && io->particle_keys->first + io->n >= (1ull << 40))))
    return GJX_ERR_INVALID;
  const gjx_scan_model* m = &p->m;
  for (int q = 0; q < m->n_step_sites; ++q) {
    const gjx_site* st = &m->step_sites[q];
    if (st->out_col >= io->n_value_cols || (st->out_col >= 0 && !io->value_cols[st->out_col])) return GJX_ERR_INVALID;
  }
  const int D = m->n_state, impl = io->particle_keys->impl;
  const uint64_t n = io->n;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    walk_ctx c;
    memset(&c, 0, sizeof c);
    c.impl = impl;
    c.pair_normals = 1;
    c.i = (uint64_t)i;
    c.scopes = &p->scopes;
    uint32_t key[4];
    key_at(io->particle_keys, (uint64_t)i, key);
    float st[GJX_SMC_MAX_STATE], wt = 0.0f, sct = 0.0f;
    for (int k = 0; k < D; ++k)
      st[k] = (io->carry0_cols && io->carry0_cols[k]) ? io->carry0_cols[k][i] : io->carry0[k];
    for (int t = 0; t < io->n_steps; ++t) {
      if (impl == 0) {
        uint32_t kt[4];
        o_fold_in(impl, key, (uint32_t)t, kt); /* THREEFRY, chained: the folded key is carried (scan.py:267-268, 276) */
        memcpy(key, kt, sizeof key);
        memcpy(c.pkey, key, sizeof key);
      } else { /* PHILOX: the particle's key on lane L + (t + 1) 2^40 — no cipher block for a key (gjx.h gjx_scan_run) */
        memcpy(c.pkey, key, sizeof key);
        c.pkey[3] = key[3] + (((uint32_t)t + 1u) << 8);
      }
      c.state = st;
      c.obs = m->n_obs ? io->obs + (size_t)t * (size_t)m->n_obs : NULL;
      site_val vals[GJX_MAX_SITES];
      float w, sc;
      site_walk(m->step_sites, m->n_step_sites, &c, vals, &w, &sc);
      for (int q = 0; q < m->n_step_sites; ++q) {
        const gjx_site* sq = &m->step_sites[q];
        if (sq->out_col < 0) continue;
        const size_t at = (size_t)t * io->col_stride + (size_t)i;
        if (vals[q].is_int) ((int32_t*)io->value_cols[sq->out_col])[at] = vals[q].i;
        else ((float*)io->value_cols[sq->out_col])[at] = vals[q].f;
      }
      float nx[GJX_SMC_MAX_STATE];
      for (int k = 0; k < D; ++k) nx[k] = eval_arg(&m->next_state[k], vals, &c);
      for (int k = 0; k < D; ++k) st[k] = nx[k];
      wt = wt + w;   /* scan.py:293 */
      sct = sct + sc; /* scan.py:290 */
    }
    for (int k = 0; k < D; ++k)
      if (io->carry_out && io->carry_out[k]) io->carry_out[k][i] = st[k];
    io->logw[i] = wt;
    if (io->score) io->score[i] = sct;
  }
  if (io->max_partials) {
    for (uint64_t b = 0; b * O_ROW < n; ++b) {
      float mx = -INFINITY;
      for (uint64_t i = b * O_ROW; i < n && i < (b + 1) * O_ROW; ++i) mx = io->logw[i] > mx ? io->logw[i] : mx;
      io->max_partials[b] = mx;
    }
  }
  if (io->row_e && io->row_s) {
    int rc = gjx_row_stats(io->logw, n, io->row_e, io->row_s, s);
    if (rc || !io->lse) return rc;
    float l = 0.0f;
    rc = gjx_lse_rows(io->row_e, io->row_s, gjx_num_max_partials(n), io->lse->e, io->lse->q, &l, io->lse->record, s);
    if (io->lse->lse) *io->lse->lse = l;
    if (io->lse->lse_shifted) *io->lse->lse_shifted = l - io->lse->shift;
    return rc;
  }
  return GJX_OK;
}

/* One step of a plan-driven filter for the slots [first_slot, first_slot + n_local) (the sharded driver's piece). */
typedef struct {
  const gjx_smc_config* cfg;
  const gjx_smc_model* m;
  int t;
  const float* obs;
  const gjx_smc_pop* prev;
  const gjx_smc_pop* out;
  uint32_t skey[4];
  const scope_info* init_scopes; /* nested calls inside init / step (NULL: flat bodies) */
  const scope_info* step_scopes;
} plan_ctx;
static float plan_propagate(void* vc, uint64_t j, int64_t a) {
  plan_ctx* p = (plan_ctx*)vc;
  const gjx_smc_model* m = p->m;
  const int D = m->n_state;
  const uint64_t slot = p->cfg->first_slot + j;
  walk_ctx c;
  memset(&c, 0, sizeof c);
  c.impl = p->cfg->impl;
  o_split_at(p->cfg->impl, p->skey, slot, c.pkey);
  c.quad_key = p->cfg->impl == 1 ? p->skey : NULL;
  c.slot = slot;
  float prev[GJX_SMC_MAX_STATE];
  if (a >= 0) {
    const peer_view pv = peer_view_of(p->cfg);
    for (int k = 0; k < D; ++k) prev[k] = *(const float*)peer_at(&pv, (const float*)p->prev->state[k] + a, (uint64_t)a);
  }
  c.state = a >= 0 ? prev : NULL;
  c.obs = p->obs;
  c.scopes = p->t == 0 ? p->init_scopes : p->step_scopes;
  const gjx_site* sites = p->t == 0 ? m->init_sites : m->step_sites;
  const int ns = p->t == 0 ? m->n_init_sites : m->n_step_sites;
  const gjx_arg* nxt = p->t == 0 ? m->init_state : m->next_state;
  site_val vals[GJX_MAX_SITES];
  float w, sc;
  site_walk(sites, ns, &c, vals, &w, &sc);
  for (int k = 0; k < D; ++k) ((float*)p->out->state[k])[j] = eval_arg(&nxt[k], vals, &c);
  return w;
}
int gjx_smc_plan_step(const gjx_smc_config* cfg, gjx_smc_plan* plan, int t, const float* obs_t, const gjx_smc_pop* prev,
                      const gjx_smc_pop* out, int32_t* prev_e_out, uint64_t* prev_q_out, int32_t* ancestors_out,
                      gjx_stream s) {
  (void)s;
  if (!cfg_ok(cfg) || !plan || t < 0 || t >= cfg->n_steps || !out || (t > 0 && !prev) || (plan->m.n_obs > 0 && !obs_t) ||
      cfg->n_filters > 1)
    return GJX_ERR_INVALID;
  plan_ctx c = {cfg, &plan->m, t, obs_t, prev, out, {cfg->step_keys[2 * t], cfg->step_keys[2 * t + 1], 0u, 0u},
                &plan->init_scopes, &plan->step_scopes};
  return smc_step_generic(cfg, t, plan->m.n_state, prev, out, prev_e_out, prev_q_out, ancestors_out, plan_propagate, &c);
}

typedef struct { gjx_smc_plan* plan; const float* obs; } plan_run;
static int plan_run_step(void* vc, const gjx_smc_config* cfg, int t, const gjx_smc_pop* prev, const gjx_smc_pop* out,
                         int32_t* pe, uint64_t* pq, int32_t* anc) {
  plan_run* r = (plan_run*)vc;
  const int no = r->plan->m.n_obs;
  return gjx_smc_plan_step(cfg, r->plan, t, no ? r->obs + (size_t)t * (size_t)no : NULL, prev, out, pe, pq, anc, NULL);
}
/* Several filters: each filter's own single run, with its keys [f, T, 2] and its slice of every output. */
int gjx_smc_run_plan(const gjx_smc_config* cfg, gjx_smc_plan* plan, const float* obs_host, int32_t* out_e,
                     uint64_t* out_q, float* const* state_out, float* logw_out, int32_t* ancestors_out,
                     void* ws, size_t ws_bytes, gjx_stream s) {
  (void)ws; (void)ws_bytes; (void)s;
  if (!cfg || !plan || !state_out || (plan->m.n_obs > 0 && !obs_host)) return GJX_ERR_INVALID;
  plan_run r = {plan, obs_host};
  void* st[GJX_SMC_MAX_STATE];
  for (int k = 0; k < plan->m.n_state; ++k) {
    if (!state_out[k]) return GJX_ERR_INVALID;
    st[k] = state_out[k];
  }
  return smc_run_filters(cfg, plan->m.n_state, plan_run_step, &r, out_e, out_q, st, logw_out, ancestors_out);
}
