/*
 * gjx.h — C-ABI of the MI355X-native vectorised-trace inference backend.
 *
 * The reference (genjax-dev/genjax-chi) is pure Python on JAX and has NO FFI / plugin boundary
 * (SURVEY.md F1, §8b): there is no existing native interface to bind.  This header therefore
 * defines the boundary a maintainer WOULD bind (ctypes stub in INTEGRATION.md); every entry point
 * names the reference call site(s) it replaces (paths relative to the reference's src/genjax/_src).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch types.  All entry points return int:
 *    GJX_OK (0) or a negative gjx_status.  No C++ exception crosses the boundary.
 *  - Pointers documented "dev" are device (HBM) pointers for libgjx_hip.so and host pointers for
 *    the CPU oracle build (oracle/libgjx_oracle.so, test infrastructure only).  Pointers
 *    documented "host" are always host memory.  All memory is BORROWED: the library never frees
 *    or retains a caller pointer past the stream-ordered call (plans copy what they keep).
 *  - Every compute entry point takes a gjx_stream (hipStream_t) and is asynchronous w.r.t. the
 *    host; no entry point synchronises, allocates or frees device memory (graph-capturable).
 *    Scratch comes from a caller-provided workspace (gjx_workspace_bytes()).
 *  - Particle axis is the fastest-varying axis of every column (struct-of-arrays trace buffer).
 *  - All arithmetic follows the bit-exact f32/u64 specification in DESIGN.md §3 ("math spec");
 *    HIP and oracle results are bit-identical on identical counters.
 */
#ifndef GJX_H
#define GJX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 0.8: the arithmetic spec of PHILOX Normal sites changed (table-driven Box-Muller): same words, same
 * pairing, normals that differ from 0.7's in the last bits.  Library and oracle of one version agree bit for bit. */
#define GJX_VERSION_MAJOR 0
#define GJX_VERSION_MINOR 10

typedef void* gjx_stream; /* hipStream_t; ignored by the oracle build */

typedef enum {
  GJX_OK = 0,
  GJX_ERR_INVALID = -1,      /* bad argument (null pointer, size 0 where not allowed, bad enum) */
  GJX_ERR_UNSUPPORTED = -2,  /* valid request this build cannot run */
  GJX_ERR_WORKSPACE = -3,    /* workspace too small */
  GJX_ERR_LAUNCH = -4,       /* HIP launch / runtime failure */
  GJX_ERR_NO_DEVICE = -5,    /* no usable gfx950 device */
  GJX_ERR_JIT = -6           /* run-time specialisation (hiprtc compile / module load) of a plan failed; the cause is
                                logged to stderr.  Never a silent fallback: GJX_PLAN_JIT_FALLBACK=1 opts into the
                                table interpreter instead */
} gjx_status;

/* ---- PRNG ------------------------------------------------------------------------------- */

/* Cipher / derivation scheme.  THREEFRY reproduces jax.random's key tree (SURVEY App. A:
 * threefry2x32, partitionable split/fold_in): keys are 2 words.  PHILOX is the native counter
 * scheme (Philox4x32-10): a key is 4 words (k0, k1, lane_lo, lane_hi) — a 64-bit
 * cipher key plus a 64-bit LANE that occupies counter words 0,1 of every block.  The children of a
 * lane-0 key are (same cipher key, lane i+1): a population's keys cost no cipher block and share
 * one cipher key; the children of a laned key are hashed to fresh lane-0 keys. */
typedef enum { GJX_RNG_THREEFRY = 0, GJX_RNG_PHILOX = 1 } gjx_rng_impl;
#define GJX_KEY_WORDS(impl) ((impl) == GJX_RNG_PHILOX ? 4 : 2)

/* A batch of n per-particle keys, either materialised or lazily derived in-register.
 *   mode 0: keys[i] = row i of keys            (dev u32[n, GJX_KEY_WORDS(impl)])
 *   mode 1: keys[i] = split(parent, *)[first + i]                 (nothing read from HBM)
 *   mode 2: keys[i] = parent for every i                          (one literal key, by value)
 * parent = (parent[0], parent[1]) with lane parent_lane (PHILOX; must be 0 for THREEFRY).
 * If has_fold, the draws of a leaf site come from the key's stream number `fold`: THREEFRY uses
 * fold_in(keys[i], fold) with fold = the per-`@`-site counter of generative_functions/
 * static.py:349-352 (from 1, constrained sites included); PHILOX carries the fold in the block
 * counter, with fold = the 0-based index of the site among the sites that consume randomness; the
 * single-word draws of a lane-0 key go four to a block of its own, those of a laned key two to a block
 * shared with its pair partner (particles 2i, 2i+1 of the batch: DESIGN.md §3.2).
 * Replaces: jax.random.split / fold_in call sites inference/smc.py:299-300, static.py:261,350. */
typedef struct {
  int32_t impl;        /* gjx_rng_impl */
  int32_t mode;        /* 0 explicit, 1 lazy split, 2 literal */
  const uint32_t* keys;/* dev, mode 0 */
  uint32_t parent[2];  /* mode 1 / 2 */
  uint64_t first;      /* mode 1: global index of element 0 (shard offset) */
  int32_t has_fold;
  uint32_t fold;
  uint64_t parent_lane;/* mode 1 / 2, PHILOX: lane of the parent key (0 = none) */
} gjx_keys;

/* f32 operand: per-particle column (dev, stride 1) or broadcast scalar when ptr == NULL. */
typedef struct {
  const float* ptr;
  float scalar;
} gjx_f32;

int gjx_version(int* major, int* minor);
const char* gjx_backend_name(void); /* "hip-gfx950" or "oracle-cpu" */

/* out[i] = the i-th key described by k (after the optional fold: fold_in(key, fold), a fresh
 * lane-0 key).  dev u32[n, GJX_KEY_WORDS(impl)].
 * Replaces jax.random.split (smc.py:300,386; vmap.py:186,201) and fold_in (static.py:350,
 * scan.py:213,268) when keys must be materialised. */
int gjx_rng_keys(const gjx_keys* k, uint64_t n, uint32_t* out, gjx_stream s);

/* out[i*m + j] = split(key_i, m)[j] for the n keys described by k (after the optional fold):
 * the nested key batch of Vmap.generate / simulate (combinators/vmap.py:186,201).
 * dev u32[n, m, GJX_KEY_WORDS(impl)]. */
int gjx_rng_split_each(const gjx_keys* k, uint64_t n, uint32_t m, uint32_t* out, gjx_stream s);

/* out[i] = 32 random bits of key i, sub-stream `sub` (DESIGN.md §3.2 bits32_at).  dev u32[n]. */
int gjx_rng_bits(const gjx_keys* k, uint32_t sub, uint64_t n, uint32_t* out, gjx_stream s);

/* ---- distributions: fused sample + log-density ------------------------------------------- *
 * Replaces ExactDensity.random_weighted / estimate_logpdf
 * (generative_functions/distributions/distribution.py:371-396) over the TFP wrappers
 * (distributions/tensorflow_probability/__init__.py:35-64; instances normal 259, gamma 164,
 * beta 82, flip 155, bernoulli 72, categorical 102-104), batched over the particle axis as
 * ImportanceK's vmap does (inference/smc.py:302-310).
 * value_out / score_out are dev [n]; score_out may be NULL.
 * Normal: THREEFRY draws sqrt(2) erfinv(u) (jax).  PHILOX with a fold (a model site) draws the
 * Box-Muller pair of particles (2i, 2i+1) of the key batch: each element derives the
 * pair's block from the key lane, so this call, gjx_importance_run and a scalar run agree bit for bit. */
int gjx_sample_logpdf_normal(const gjx_keys* k, gjx_f32 loc, gjx_f32 scale, float* value_out,
                             float* score_out, uint64_t n, gjx_stream s);
int gjx_sample_logpdf_gamma(const gjx_keys* k, gjx_f32 concentration, gjx_f32 rate,
                            float* value_out, float* score_out, uint64_t n, gjx_stream s);
int gjx_sample_logpdf_beta(const gjx_keys* k, gjx_f32 a, gjx_f32 b, float* value_out,
                           float* score_out, uint64_t n, gjx_stream s);
/* probs in [0,1]; value is uint8 (0/1) — tfd.Bernoulli(probs=p, dtype=bool). */
int gjx_sample_logpdf_bernoulli(const gjx_keys* k, gjx_f32 probs, uint8_t* value_out,
                                float* score_out, uint64_t n, gjx_stream s);
/* logits: dev f32 [n_rows, n_cat] row-major; row of particle i = row_index ? row_index[i] : (n_rows==1 ? 0 : i).
 * mode 0 = Gumbel-max (jax.random.categorical semantics, n_cat uniforms per draw),
 * mode 1 = inverse-CDF on the fixed-point CDF (one uniform per draw). value int32. */
int gjx_sample_logpdf_categorical(const gjx_keys* k, const float* logits, uint64_t n_rows,
                                  uint32_t n_cat, const int32_t* row_index, int mode,
                                  int32_t* value_out, float* score_out, uint64_t n, gjx_stream s);

/* Elementwise f32 functions of the spec over a column: what `exp` / `log` / a division by a number between the sites
 * of a model body (jnp.exp, jnp.log, x / c in the reference's bodies, e.g. tests/inference/test_smc.py:63) compute on
 * the per-site path — the same bits GJX_EXPR_EXP / _LOG / _DIV compute inside a fused plan, so a body gives the same
 * trace whichever route it takes.  out[i] = exp(x[i]) | log(x[i]) | x[i] / c | c / x[i] | sqrt(x[i]) | |x[i]|; x, out dev f32[n]
 * (may alias).  (sigmoid(x) = 1 / (1 + exp(-x)) is these steps, on both routes: the host API expands it.) */
enum { GJX_MAP_EXP = 0, GJX_MAP_LOG = 1, GJX_MAP_DIV = 2, GJX_MAP_RDIV = 3, GJX_MAP_SQRT = 4, GJX_MAP_ABS = 5 };
int gjx_map_f32(int op, const float* x, float c, float* out, uint64_t n, gjx_stream s);

/* log-density of given values (constrained sites: distribution.py:144-147, 383-396). */
int gjx_logpdf_normal(gjx_f32 value, gjx_f32 loc, gjx_f32 scale, float* score_out, uint64_t n,
                      gjx_stream s);
int gjx_logpdf_gamma(gjx_f32 value, gjx_f32 concentration, gjx_f32 rate, float* score_out,
                     uint64_t n, gjx_stream s);
int gjx_logpdf_beta(gjx_f32 value, gjx_f32 a, gjx_f32 b, float* score_out, uint64_t n,
                    gjx_stream s);
/* value: dev uint8[n] or NULL => value_scalar for every particle. */
int gjx_logpdf_bernoulli(const uint8_t* value, int value_scalar, gjx_f32 probs, float* score_out,
                         uint64_t n, gjx_stream s);
int gjx_logpdf_categorical(const int32_t* value, int value_scalar, const float* logits,
                           uint64_t n_rows, uint32_t n_cat, const int32_t* row_index,
                           float* score_out, uint64_t n, gjx_stream s);

/* ---- fused static-model importance (the `@gen` body as one kernel) ------------------------ *
 * Replaces vmap(target.importance) over StaticGenerativeFunction.generate
 * (inference/smc.py:308-310 -> inference/sp.py:83-87 -> generative_functions/static.py:340-399):
 * per particle, walk the `@` sites in program order; site s uses key fold_in(particle_key, s+1);
 * latent sites sample and add their log-density to score; observed sites add their log-density
 * to score AND weight (static.py:377; StaticTrace.get_score static.py:102-105). */
typedef enum {
  GJX_DIST_NORMAL = 0,     /* args: loc, scale */
  GJX_DIST_GAMMA = 1,      /* args: concentration, rate */
  GJX_DIST_BETA = 2,       /* args: concentration1, concentration0 */
  GJX_DIST_BERNOULLI = 3,  /* args: probs (flip) */
  GJX_DIST_CATEGORICAL = 4 /* arg0 selects the logits row; table = logits [n_rows,n_cat] */
} gjx_dist;

typedef enum {
  GJX_ARG_CONST = 0, /* value = offset */
  GJX_ARG_SITE = 1,  /* value = scale * (value of site `ref`, as f32) + offset */
  GJX_ARG_INPUT = 2, /* value = scale * input_cols[ref][i] + offset */
  GJX_ARG_TABLE = 3, /* value = table[(int) value of site `ref`] (dev f32 table) */
  GJX_ARG_STATE = 4, /* SMC plans: value = scale * state[ref] of the particle's ANCESTOR + offset */
  GJX_ARG_OBS = 5,   /* SMC plans: value = scale * obs[t][ref] + offset (this step's observation constants) */
  GJX_ARG_PARAM = 6, /* importance plans: value = scale * params[ref] + offset — a LAUNCH-UNIFORM parameter
                        (gjx_plan_set_params): observations and model arguments that change from dataset to dataset
                        without changing the model's structure.  libgjx_hip.so passes them as kernel arguments
                        (scalar registers), so one specialised kernel serves every dataset: no recompilation. */
  GJX_ARG_EXPR = 7   /* value = a small POSTFIX PROGRAM over earlier sites, input columns, parameters / state / observation
                        constants and literals (+ - * / and negation): `table` points at `ref` gjx_expr_op entries in HOST memory (copied at plan
                        creation).  What a model body writes between its `@` sites — `normal(w * x + b, s)`
                        (static.py:340-380 runs that arithmetic as traced jnp ops) — evaluated per particle in f32, one
                        rounding per operation, in program order (no fusion).  Distribution arguments of sites and the
                        state arguments of SMC / scan plans (`x + 0.1 * v`: a deterministic update of a carried component);
                        not observed values, not the row of a categorical site.  libgjx_hip.so runs such
                        plans as specialised kernels only (GJX_ERR_UNSUPPORTED if specialisation is turned off). */
} gjx_arg_kind;
#define GJX_MAX_PARAMS 64

typedef enum {
  GJX_EXPR_CONST = 0, /* push `value` */
  GJX_EXPR_SITE = 1,  /* push the value of site `ref` (an integer-valued site converts to f32) */
  GJX_EXPR_INPUT = 2, /* push input_cols[ref][i] */
  GJX_EXPR_PARAM = 3, /* push params[ref]            (importance plans) */
  GJX_EXPR_STATE = 4, /* push state[ref] of the ancestor (scan / SMC plans) */
  GJX_EXPR_OBS = 5,   /* push obs[t][ref]            (scan / SMC plans) */
  GJX_EXPR_ADD = 6,   /* pop b, pop a, push a + b */
  GJX_EXPR_SUB = 7,   /* ... a - b */
  GJX_EXPR_MUL = 8,   /* ... a * b */
  GJX_EXPR_NEG = 9,   /* pop a, push -a */
  GJX_EXPR_DIV = 10,  /* pop b, pop a, push a / b (IEEE, correctly rounded) */
  GJX_EXPR_EXP = 11,  /* pop a, push exp(a): the spec's f32 exp (gjx_map_f32 GJX_MAP_EXP computes the same bits) */
  GJX_EXPR_LOG = 12,  /* pop a, push log(a): the spec's f32 log (0 -> -inf, negative -> NaN) */
  GJX_EXPR_SQRT = 13, /* pop a, push sqrt(a) (IEEE, correctly rounded; negative -> NaN) */
  GJX_EXPR_ABS = 14,  /* pop a, push |a| */
  GJX_EXPR_MAX = 15,  /* pop b, pop a, push max(a, b): a NaN if either is one (torch.maximum / jnp.maximum) */
  GJX_EXPR_MIN = 16,  /* ... min(a, b) */
  GJX_EXPR_LT = 17,   /* pop b, pop a, push a < b ? 1 : 0 (IEEE: false with a NaN) */
  GJX_EXPR_LE = 18,   /* ... a <= b */
  GJX_EXPR_EQ = 19,   /* ... a == b */
  GJX_EXPR_SELECT = 20 /* pop f, pop t, pop c, push c != 0 ? t : f (jnp.where(c, t, f)) */
} gjx_expr_opcode;
typedef struct {
  int32_t op;   /* gjx_expr_opcode */
  int32_t ref;
  float value;
} gjx_expr_op;
#define GJX_MAX_EXPR_OPS 32 /* per argument (r03: 16 -> 32: sigmoid / softplus / where expand to several entries); the evaluation stack is at most 8 deep */

typedef struct {
  int32_t kind;
  int32_t ref;
  float scale;
  float offset;
  const float* table; /* GJX_ARG_TABLE: dev f32 table; GJX_ARG_EXPR: (const gjx_expr_op*) host program of `ref` entries */
} gjx_arg;

typedef struct {
  int32_t dist;       /* gjx_dist */
  int32_t observed;   /* 0 latent (sampled), 1 observed (constrained) */
  int32_t out_col;    /* index into value_cols, or -1: do not materialise this site's value */
  int32_t n_cat;      /* categorical: number of categories */
  int32_t n_rows;     /* categorical: rows in `logits` */
  int32_t cat_mode;   /* categorical: 0 Gumbel-max, 1 inverse-CDF */
  gjx_arg arg[2];
  gjx_arg obs;        /* observed value: CONST, INPUT or PARAM (int-valued dists: rounded to nearest) */
  const float* logits;/* dev f32 [n_rows,n_cat], categorical only (borrowed until plan destroy) */
} gjx_site;

typedef struct gjx_plan gjx_plan;

#define GJX_MAX_SITES 64

int gjx_plan_create(const gjx_site* sites /*host*/, int n_sites, gjx_plan** out);
/* Plan options (flags of gjx_plan_create_ex; gjx_plan_create passes 0).
 *  GJX_PLAN_FAST_MATH: the north star bounds log-weights by 1e-5 relative on the path WITHOUT resampling, so an
 *    importance plan may opt into the hardware transcendentals (v_log / v_exp / v_sqrt / v_sin / v_cos) for the
 *    CONTINUOUS parts of the walk: the Box-Muller transform of PHILOX Normal sites, the transcendental terms of
 *    log-densities and the row-anchored weight sums.  Everything that decides something (gamma rejection tests,
 *    categorical CDFs, Bernoulli thresholds) stays on the exact functions, so the same particles are drawn from the
 *    same counters and every value / log-weight agrees with the exact plan to <= 1e-5 relative (tested); results
 *    are no longer bit-identical to the oracle.  Honoured by the specialised (hiprtc) kernels of libgjx_hip.so; the
 *    table interpreter and the oracle ignore it (they ARE the exact specification). */
#define GJX_PLAN_FAST_MATH 1u
int gjx_plan_create_ex(const gjx_site* sites /*host*/, int n_sites, uint32_t flags, gjx_plan** out);
/* Nested `@gen` calls inside a plan (static.py:175-193, 349-352, 374-375; generative_function.py:1568-1583): the body
 * `callee(args) @ "addr"` takes ONE counter of its caller like any other `@` site, runs under
 * sub_key = fold_in(key, counter), and numbers its own sites from 1 (THREEFRY; PHILOX: from draw 0 of the lone key
 * sub_key).  The site table stays FLAT, in program order; scope k (1-based; 0 is the plan's own body) says which
 * contiguous range [begin, end) of it — deeper calls included — one call produced and which scope made the call.  Scopes
 * are listed in call order (begin non-decreasing, a caller before its callees); a call that visits no site has
 * begin == end (it still takes its counter, at that position).  Depth <= 4, at most GJX_MAX_SCOPES calls.
 * Plans with scopes run as specialised kernels only (like GJX_ARG_EXPR programs); the oracle walks them as written. */
typedef struct {
  int32_t parent; /* the calling scope: 0 = the plan's body, k = scopes[k - 1] */
  int32_t begin, end;
} gjx_scope;
#define GJX_MAX_SCOPES 16
int gjx_plan_create_scoped(const gjx_site* sites /*host*/, int n_sites, const gjx_scope* scopes /*host*/, int n_scopes,
                           uint32_t flags, gjx_plan** out);
int gjx_plan_destroy(gjx_plan* p);
/* The values of the plan's GJX_ARG_PARAM references for the launches that follow (host f32[n_params], copied; n_params
 * <= GJX_MAX_PARAMS and > every referenced index).  Per-site constants that depend on them (1/scale, the log
 * normaliser, lgamma terms) are re-derived on the host by the spec functions, exactly as plan creation does for
 * constants.  Set-then-run of one plan must be ordered by the caller (a plan is not a concurrent object). */
int gjx_plan_set_params(gjx_plan* p, const float* params /*host*/, int n_params);
/* Plan specialisation (libgjx_hip.so): on first use per RNG scheme a plan is lowered to a
 * straight-line gfx950 kernel — the same device functions in the same order, constants folded —
 * and compiled with hiprtc; GJX_PLAN_JIT=0 keeps the table-interpreter kernel.  Diagnostics:
 * the generated HIP source (buf nullable; *needed = bytes incl. NUL) and an offline compile check
 * (needs no GPU).  The oracle build returns GJX_ERR_UNSUPPORTED for both. */
int gjx_plan_specialized_source(const gjx_plan* p, int impl, char* buf, size_t buf_len, size_t* needed);
/* Build (hiprtc) and load now the kernel gjx_importance_run would build on its first launch with
 * this key form, so that no launch pays the ~0.2 s compilation.  Optional; the oracle build returns
 * GJX_OK and does nothing. */
int gjx_plan_prepare(gjx_plan* p, const gjx_keys* particle_keys);
int gjx_plan_compile_check(const gjx_plan* p, int impl);
/* Counters of the run-time specialisation cache (libgjx_hip.so; the oracle reports zeros): hiprtc compilations so far,
 * code objects currently loaded, code objects unloaded.  Modules are keyed by their generated source — the model's
 * STRUCTURE (values that arrive as GJX_ARG_PARAM are kernel arguments, not source) — reference-counted by the plans
 * using them, and unreferenced ones are kept for reuse up to GJX_JIT_CACHE_MAX (default 64) entries, then unloaded
 * least-recently-used first.  Each output nullable. */
int gjx_jit_stats(uint64_t* compiles, uint64_t* cached_modules, uint64_t* evictions);
/* Which ROUTE compiled the generated kernels of this process (r04; each output nullable): child_compiles = code objects
 * produced by the helper process gjx_jitc (the default and only route unless the caller opts out), inproc_compiles = by
 * hiprtc inside the calling process (GJX_JIT_INPROC=1, or GJX_JIT_INPROC_FALLBACK=1 after a helper that could not be
 * started), child_failures = helpers that rejected a source or died on it (each a GJX_ERR_JIT for the caller),
 * spawn_failures = helpers that could not be started.  Without the opt-in a spawn failure is GJX_ERR_JIT: the compiler
 * never moves into the caller's address space silently.  (The oracle reports zeros.) */
int gjx_jit_routes(uint64_t* child_compiles, uint64_t* inproc_compiles, uint64_t* child_failures, uint64_t* spawn_failures);
/* r04 (0.10): whole one-filter runs (gjx_smc_run_lgssm / gjx_smc_run_hmm) are REPLAYED as one hipGraph from the second run of a
 * shape on (same sizes, generator, threshold and buffers; keys, observations and the model's scalar parameters are free: they
 * travel in a device block).  captures = graphs instantiated, replays = runs that were one graph launch; each nullable.  The
 * results are the plain run's bit for bit.  GJX_SMC_GRAPH=0 keeps every run a stream of launches.  (The oracle reports zeros.) */
int gjx_smc_run_graph_stats(uint64_t* captures, uint64_t* replays);
/* Compile a kernel source the way generated plan kernels are compiled (gfx950, the device header available as
 * "gjx_device.hpp"), needs no GPU: GJX_OK, or GJX_ERR_JIT if the compiler rejects the source OR DIES on it.  The
 * compiler runs in a child process (csrc/gjx_jitc.cpp): an AMDGPU-backend crash on generated source — it happened once —
 * is an error code and a log line for the caller, never an abort.  (GJX_ERR_UNSUPPORTED in the oracle build.) */
int gjx_jit_compile_source(const char* source);
/* particle_keys: the per-particle keys BEFORE the per-site fold (has_fold must be 0).
 * input_cols / value_cols: host arrays of dev pointers (each column dev [n], 4-byte elements:
 * f32, or int32 for Bernoulli/Categorical values; at most 16 input columns).  score, logw: dev
 * f32[n] (score nullable; logw nullable when row_e / row_s are given: an estimate that needs only logsumexp(lw) — a plan
 * whose sites all have out_col = -1, score and logw null — writes no per-particle column at all).  max_partials: nullable dev f32[gjx_num_max_partials(n)]; when given, the
 * kernel also stores the maxima of logw per 256-particle row so the following log-sum-exp skips
 * its max pass.  row_e / row_s: nullable (both or neither) row-anchored partial sums, see
 * gjx_lse_rows: with them the log-marginal needs one further tiny kernel and no pass over logw.
 * lse: nullable; needs row_e/row_s.  The pass's log-sum-exp (the outputs of gjx_lse_rows over these
 * rows, each nullable) is produced by the SAME launch: the workgroup that finishes last folds the row
 * pairs (replaces logsumexp(lw) at inference/smc.py:97 with zero extra launches).  tickets: dev
 * u32[GJX_LSE_TICKET_WORDS] owned by the caller, zero before the first use; every launch leaves it zero,
 * so launches sharing it must be stream-ordered.  (r03: the struct grew by lse_shifted / shift: zero-initialise it.) */
#define GJX_LSE_TICKET_WORDS (17 * 64) /* 17 counters, each on its own 256-byte line */
typedef struct {
  int32_t* e;       /* dev int32[1] */
  uint64_t* q;      /* dev u64[1] */
  float* lse;       /* dev f32[1] */
  uint64_t* record; /* dev u64[GJX_LSE_RECORD_WORDS] */
  uint32_t* tickets;
  float* lse_shifted; /* nullable dev f32[1]: lse - shift (ONE f32 subtraction), e.g. shift = log K gives the log-marginal
                         estimate logsumexp(lw) - log K (inference/smc.py:96-97) out of the same launch */
  float shift;
} gjx_lse_out;
int gjx_importance_run(const gjx_plan* p, const gjx_keys* particle_keys,
                       const float* const* input_cols, int n_input_cols, void* const* value_cols,
                       int n_value_cols, float* score, float* logw, uint64_t n,
                       float* max_partials, int32_t* row_e, uint64_t* row_s, const gjx_lse_out* lse,
                       gjx_stream s);

/* ONE library call for `ImportanceK(target, K).log_marginal_likelihood_estimate(key)` (inference/smc.py:83-97, 296-318): the
 * reference's key derivation — key, sub = split(key) in the estimate, key, sub = split(sub) in run_smc, particle keys =
 * split(sub, K) (lazy: nothing is materialised) — then the walk of an ESTIMATE-ONLY plan (every site's out_col = -1: no
 * value column, no score, no log-weight column), the fold of its row sums by the workgroup that finishes last, and
 * out[0] = logsumexp(lw) - shift (shift = log K).  The caller's scalar key arrives by value (k0, k1, lane; lane 0 for
 * THREEFRY).  row_e / row_s / lse: the scratch of gjx_importance_run's fused form (lse.lse_shifted / lse.shift are ignored:
 * `out` and `shift` take their place).  One launch, no allocation, no synchronisation. */
typedef struct {
  const gjx_plan* plan;
  uint64_t n;
  const float* const* input_cols;
  int32_t n_input_cols;
  int32_t impl;
  int32_t* row_e;
  uint64_t* row_s;
  gjx_lse_out lse;
} gjx_estimate_io;
int gjx_importance_estimate(const gjx_estimate_io* io, uint32_t k0, uint32_t k1, uint64_t lane, float* out /*dev f32[1]*/,
                            float shift, gjx_stream s);

/* n_pass (<= 32) independent passes of the same plan — the algorithm vmapped over keys.  Pass b draws from
 * particle_keys[b] and writes b * pass_stride elements further in every output column (score, logw, value
 * columns; pass_stride >= n, even for the fast path) and b * row_stride entries further in max_partials /
 * row_e / row_s (row_stride >= gjx_num_max_partials(n)); input columns are shared.  Lazy key batches that
 * differ only in their lane-0 parent run as ONE launch: a 1e6-particle pass is under two rounds of the
 * machine, several passes per launch keep it full (17.5 instead of 21.5 us per pass on MI355X).  Results
 * are those of n_pass separate gjx_importance_run calls, bit for bit. */
int gjx_importance_run_batch(const gjx_plan* p, const gjx_keys* particle_keys /*host [n_pass]*/, int32_t n_pass,
                             uint64_t pass_stride, uint64_t row_stride, const float* const* input_cols,
                             int n_input_cols, void* const* value_cols, int n_value_cols, float* score,
                             float* logw, uint64_t n, float* max_partials, int32_t* row_e, uint64_t* row_s,
                             gjx_stream s);

/* ---- weights: log-sum-exp, single draw, resampling ---------------------------------------- */

typedef enum {
  GJX_OP_LOGSUMEXP = 0,
  GJX_OP_CATEGORICAL_INDEX = 1,
  GJX_OP_RESAMPLE = 2,
  GJX_OP_SMC = 3
} gjx_op;
size_t gjx_workspace_bytes(int op, uint64_t n);

/* Number of fractional bits of the fixed-point weight q = rint(exp(lw - max) * 2^frac) used by
 * every weight sum for a population of n_total particles. */
int gjx_frac_bits(uint64_t n_total);

/* Tiles: every kernel processes particles in tiles of gjx_smc_tile() (1024); per-tile partial
 * arrays have gjx_num_tiles(n) entries. */
uint64_t gjx_num_tiles(uint64_t n);
uint64_t gjx_num_max_partials(uint64_t n); /* rows: entries of max_partials / row_e / row_s */

/* Row-anchored log-sum-exp.  Each 256-particle row b is summarised by
 * (row_e[b], row_s[b]): anchor exponent e_b = ceil(max_b * log2 e) and S_b = sum of
 * rint(exp(x_i - e_b ln 2) * 2^30) as exact u64 — computable by the kernel that PRODUCES the
 * log-weights, before any global maximum exists (gjx_importance_run fills them when row_e/row_s
 * are given).  Rows combine exactly: e = max e_b; bucket B_d = sum of S_b over the rows with
 * e - e_b == d, d in [0, 64) (exact u64; rows further below carry no mass); Q = sum_d (B_d >> d);
 * lse = e ln 2 + log(Q 2^-30).  gjx_row_stats is the one-pass producer for arbitrary x.
 *
 * out_record (nullable, dev u64[GJX_LSE_RECORD_WORDS]): word 0 = e (int64), word 1+d = B_d — the
 * exchangeable summary of a population shard.  Bucket sums are exact integers, so records of shards
 * that split the population at row boundaries merge (gjx_lse_combine) into the SAME (e, Q, lse) as one
 * gjx_lse_rows over all rows: a sharded ImportanceK pass needs ONE all-gather of 520 bytes per rank and
 * no pass over logw (replaces logsumexp(lw) at inference/smc.py:97 for a sharded population). */
#define GJX_LSE_RECORD_WORDS 65
int gjx_row_stats(const float* x, uint64_t n, int32_t* row_e, uint64_t* row_s, gjx_stream s);
int gjx_lse_rows(const int32_t* row_e, const uint64_t* row_s, uint64_t n_rows, int32_t* out_e /*nullable*/,
                 uint64_t* out_q /*nullable*/, float* out_lse /*nullable*/, uint64_t* out_record /*nullable*/,
                 gjx_stream s);
/* The same fold for n_batch independent passes in ONE launch (one workgroup each): pass p's rows
 * start at row_e / row_s + p*batch_stride (batch_stride >= n_rows); outputs are arrays of n_batch
 * entries (out_record dev u64[n_batch, 65]).  A one-workgroup launch costs ~5 us of latency, more than a
 * quarter of the importance kernel at 1e6 particles: folding B passes together divides that by B. */
int gjx_lse_rows_batch(const int32_t* row_e, const uint64_t* row_s, uint64_t n_rows, int32_t n_batch,
                       uint64_t batch_stride, int32_t* out_e, uint64_t* out_q, float* out_lse,
                       uint64_t* out_record, gjx_stream s);
/* Merge records: for each of n_batch independent passes p (records of pass p start at
 * records + p*batch_stride words) combine n_records records lying record_stride words apart
 * (>= GJX_LSE_RECORD_WORDS; the layout of an all-gather of [n_batch, 65] blocks is record_stride =
 * n_batch*65, batch_stride = 65).  Outputs are arrays of n_batch entries (each nullable);
 * out_record dev u64[n_batch, 65] receives the merged records. */
int gjx_lse_combine(const uint64_t* records, int32_t n_records, uint64_t record_stride, int32_t n_batch,
                    uint64_t batch_stride, int32_t* out_e, uint64_t* out_q, float* out_lse,
                    uint64_t* out_record, gjx_stream s);

/* out_max[0] = max_i x[i] (dev f32). Pass 1 of logsumexp; multi-GPU callers all-reduce(max) it.
 * max_partials_in: nullable row maxima already produced by gjx_importance_run
 * (gjx_num_max_partials(n) entries; then x may be NULL and only the final reduction runs). */
int gjx_max_f32(const float* x, uint64_t n, const float* max_partials_in, float* out_max, void* ws,
                size_t ws_bytes, gjx_stream s);
/* out_q[0] = sum_i rint(exp(x[i] - max[0]) * 2^frac_bits) as exact u64 (order-independent);
 * multi-GPU callers all-reduce(sum) it. */
int gjx_expsum_fix(const float* x, uint64_t n, const float* max_dev, int frac_bits,
                   uint64_t* out_q, void* ws, size_t ws_bytes, gjx_stream s);
/* out_lse[0] = max + log(q * 2^-frac_bits) in f32 (math spec log). */
int gjx_lse_finish(const float* max_dev, const uint64_t* q_dev, int frac_bits, float* out_lse,
                   gjx_stream s);
/* Single-device convenience = the three calls above with frac = gjx_frac_bits(n).  out_lse /
 * out_max / out_q nullable; max_partials_in as for gjx_max_f32.
 * Replaces jax.scipy.special.logsumexp at inference/smc.py:97,107,464. */
int gjx_logsumexp_f32(const float* x, uint64_t n, const float* max_partials_in, float* out_lse,
                      float* out_max, uint64_t* out_q, void* ws, size_t ws_bytes, gjx_stream s);

/* One index ~ Categorical(softmax(logits)); key: scalar key (mode 0, n=1 keys) on host side of
 * gjx_keys.  mode 0 Gumbel-max (n uniforms; jax.random.categorical), mode 1 inverse-CDF.
 * Replaces ParticleCollection.sample_particle's draw (inference/smc.py:102-109). */
int gjx_categorical_index(const gjx_keys* key, const float* logits, uint64_t n, int64_t* out_idx,
                          int mode, void* ws, size_t ws_bytes, gjx_stream s);

/* r04: n_batch (<= 64) independent Gumbel-max draws in ONE launch — the draw of `sample_particle` (inference/smc.py:102-109)
 * under `vmap(alg.random_weighted)` over keys (README.md:111-113: 50 trials, one particle each).  Draw b takes its index among
 * the n logits at logits + b * stride under the scalar key keys[b] (host array; one generator); n <= gjx_smc_tile().  out_idx
 * dev int64[n_batch]; entry b equals gjx_categorical_index(&keys[b], logits + b * stride, n, mode 0) bit for bit. */
int gjx_categorical_index_batch(const gjx_keys* keys /*host [n_batch]*/, int32_t n_batch, const float* logits, uint64_t n,
                                uint64_t stride, int64_t* out_idx, gjx_stream s);

/* Tile-anchored weights: what a resampling reads.  Tile t = particles [1024 t, 1024 t + 1024):
 * e_t = ceil(max_t x * log2 e) (the row anchor of 3.5b on a tile); q_i = rint(exp(x_i - e_t ln 2) * 2^30), ONE u32 PER
 * PARTICLE (what a step stores instead of the log-weight); the tile's record: S_t = sum q_i and e_t (gjx_tile_rec), the
 * running sum of q after every 64th particle (gjx_tile_sub: sub[b] = sum over the tile's particles [0, 64 (b + 1));
 * sub[15] = S_t), the ESS sums r1 = sum (q_i >> 14), r2 = sum (q_i >> 14)^2 (gjx_tile_ess).  Everything is known to the workgroup that produces the tile's
 * log-weights — no grid-wide maximum — so the kernel that propagates a population emits it, and a bootstrap step is ONE
 * launch.  Records merge exactly: e = max e_t, d_t = e - e_t, M_t = S_t >> d_t (0 from d_t = 64), P_t = sum_{t' < t}
 * M_t', Q = sum_t M_t; lse = e ln 2 + log(Q 2^-30). */
typedef struct {
  uint64_t s; /* S_t */
  int32_t e;  /* e_t; GJX_TILE_EMPTY: the tile carries no mass */
  int32_t pad;
} gjx_tile_rec;
typedef struct {
  uint64_t sub[16]; /* running sum of q after every 64th particle of the tile */
} gjx_tile_sub;
typedef struct {
  uint64_t r1, r2; /* ESS sums */
} gjx_tile_ess;
#define GJX_TILE_EMPTY (-(1 << 30))
#define GJX_TILE_FRAC 30
/* The record of a tile lives in three DENSE arrays indexed by tile: recs (16 bytes: what every workgroup merges), subs
 * (128 bytes: what the few lanes that scan the tile read), ess (16 bytes: adaptive filters only).
 * qw dev u32[n], recs / subs / ess (nullable) dev [gjx_num_tiles(n)] of arbitrary log-weights x (one pass). */
int gjx_tile_weights(const float* x, uint64_t n, uint32_t* qw, gjx_tile_rec* recs, gjx_tile_sub* subs, gjx_tile_ess* ess,
                     gjx_stream s);
/* out_e[0], out_q[0] = the merged anchor and total mass of `recs` (each nullable). */
int gjx_tile_merge(const gjx_tile_rec* recs, uint64_t n_tiles, int32_t* out_e, uint64_t* out_q, gjx_stream s);

/* ancestors[j], j < n_out: systematic (one 64-bit uniform, monotone ancestors) or multinomial
 * (n_out iid draws) resampling from softmax(logw).
 * Systematic: tile-anchored weights (above): u0 = top 53 bits of the key's 64-bit draw, scale = n_out / f64(Q); teeth
 * below the START of tile t: nlo_t = clamp(ceil(f64(P_t) scale - u0), 0, n_out); below particle i of tile t, with c_i the
 * running sum of q inside the tile (exact in float64): n_i = min(clamp(ceil(fma(c_i, scale 2^-d_t, f64(P_t) scale - u0)),
 * 0, n_out), nlo_{t+1}), a tile's last particle ending at nlo_{t+1} and the population's last at n_out;
 * ancestors[j] = min{i : n_i > j}.  No mass at all (Q = 0): ancestors[j] = floor(j n / n_out).
 * out_e / out_q (nullable): merged anchor and total mass of logw.
 * Multinomial: max-anchored weights (§3.5); out_max / out_q: the (max, fixed-point sum) pair of logw.
 * NOT in the reference library (docs idiom
 * docs/cookbook/inactive/inference/importance_sampling.ipynb cell 16). */
int gjx_resample_systematic(const gjx_keys* key, const float* logw, uint64_t n, uint64_t n_out,
                            int32_t* ancestors, int32_t* out_e, uint64_t* out_q, void* ws,
                            size_t ws_bytes, gjx_stream s);
int gjx_resample_multinomial(const gjx_keys* key, const float* logw, uint64_t n, uint64_t n_out,
                             int32_t* ancestors, float* out_max, uint64_t* out_q, void* ws,
                             size_t ws_bytes, gjx_stream s);

/* dst_cols[c][j] = src_cols[c][ancestors[j]] for 4-byte columns.  src_cols/dst_cols: host arrays
 * of dev pointers.  Replaces ParticleCollection.get_particle's tree_map(v[idx])
 * (inference/smc.py:90-91) for a vector of indices. */
int gjx_gather_cols(const int32_t* ancestors, uint64_t n_out, const void* const* src_cols,
                    void* const* dst_cols, int n_cols, gjx_stream s);

/* ---- fused bootstrap SMC for the benchmark state-space models ------------------------------ *
 * One call enqueues the whole T-step filter — ONE kernel per step — on the stream, no host sync.
 * Particle slot j (global index) of step t draws ONE 32-bit word.  THREEFRY: the first
 * single-word draw of the slot key split(step_keys[t], *)[j] (site counter 1), normals by erfinv.  PHILOX: step
 * keys are lane-0 keys and slots 4g .. 4g+3 share the block PH(ctr = (g_lo, g_hi, 0, 'Q'), key = step key), slot j
 * taking word j & 3; the LGSSM's normals are Box-Muller pairs over the quad's words ((w0,w1) -> slots 4g, 4g+1;
 * (w2,w3) -> 4g+2, 4g+3)
 * (the kernel `@gen` body has one latent site; Scan.generate scan.py:237-294 is the reference's
 * T-loop, resampling itself is not in the reference: SURVEY F3/E3).
 * Multi-device: each rank owns slots [first_slot, first_slot+n_local) of n_total and passes
 * n_local < n_total; the exchange hooks below are used by the host between kernels. */
typedef struct {
  float x0_loc, x0_scale; /* x_0 ~ N(x0_loc, x0_scale) */
  float a;                /* x_t ~ N(a * x_{t-1}, q) */
  float q;
  float r;                /* y_t ~ N(x_t, r) */
} gjx_lgssm;

typedef struct {
  int32_t n_states;
  int32_t init_state;        /* z_{-1} (discrete_hmm.py:101 uses a fixed initial state) */
  const float* trans_logits; /* dev f32 [K,K]: row = previous state */
  const float* obs_logits;   /* dev f32 [K,K]: row = state, column = observation */
} gjx_hmm;

/* r04: the PEER transport of a sharded filter (below: "multi-GPU").  A population sharded over `world` ranks lives in
 * `world` ARENAS of identical layout, one per rank: the address of any element of any array on rank o is the address of the
 * same element on this rank plus delta[o] bytes (both as mapped in THIS process: the rank's own device memory, a peer's
 * through hipIpcOpenMemHandle / peer access, or another block of the same allocation for virtual ranks).  A step with
 * cfg->peers set reads the source population WHERE IT LIVES — tile k's weights, sub-prefixes and state columns from the
 * arena of its owner, rank k / (tiles / world) — and the tile records from its own arena, where every rank has deposited
 * them (gjx_smc_peer_signal): no collective, no exchange, nothing decided on the host between two steps.
 * Ordering: flags is this rank's row of arrival words (dev u64[GJX_MAX_PEERS] inside the arena, zero when allocated and
 * monotone afterwards; flags[q] is written by rank q only).  A step's kernel reads nothing of the source population before
 * every flags[q], q < world, is >= wait_value — a BOUNDED poll by one wave per workgroup (timeout_ms; 0 = 10 s), followed
 * by a system-scope acquire; a wait that times out sets *error (dev u32[1] in the arena) and the kernel returns without
 * writing anything, so a lost peer is an error the caller reads after the run, never a hang. */
#define GJX_MAX_PEERS 8
typedef struct gjx_smc_peers {
  int32_t world;                /* 2 .. GJX_MAX_PEERS */
  int32_t rank;
  int64_t delta[GJX_MAX_PEERS]; /* delta[rank] = 0 */
  uint64_t* flags;              /* this rank's arrival words, inside the arena */
  uint32_t* error;              /* inside the arena */
  uint64_t wait_value;          /* what the step waits for (the sharded drivers set it per step) */
  uint32_t timeout_ms;
  int32_t pad;
  /* A DEFERRED signal (signal_value != 0; the sharded drivers set it when gjx_smc_peer_signal_fused() says so): the step's
   * first launch — the group-record launch of a population beyond 1024 tiles, which waits for the peers anyway — first
   * deposits these records of the PREVIOUS step and raises signal_value, exactly as gjx_smc_peer_signal would have: the
   * signal launch of step t and the waiting launch of step t + 1 are ONE launch. */
  const gjx_tile_rec* signal_recs;
  const gjx_tile_ess* signal_ess;
  uint64_t signal_first_tile, signal_n_tiles, signal_value;
} gjx_smc_peers;

/* Layout-independent description of one SMC run. */
typedef struct {
  int32_t impl;            /* gjx_rng_impl */
  uint64_t n_total;        /* global particle count (< 2^31) */
  uint64_t first_slot;     /* this rank's first slot (multiple of the tile size) */
  uint64_t n_local;        /* this rank's slot count */
  int32_t n_steps;         /* T */
  const uint32_t* step_keys;     /* host u32[T,2]: per-step propagate keys (lane 0: fold_in results) */
  const uint32_t* resample_keys; /* host u32[T,2]: per-step resampling keys (entry 0 unused) */
  /* Several independent filters of n_total particles each, stepping in the same launches (the filter vmapped
   * over keys; whole-run calls gjx_smc_run_lgssm / gjx_smc_run_hmm on one device only; <= 16).  0 or 1: one
   * filter.  With F = n_filters > 1: step_keys / resample_keys are host u32[F,T,2]; filter f's particles lie
   * f * filter_stride further in state_out / logw_out (dev [F, filter_stride], filter_stride =
   * gjx_num_tiles(n_total) * gjx_smc_tile()), its results in out_e / out_q dev [F, T], its ancestors in
   * ancestors_out dev int32[T, F, filter_stride] (indices within the filter); the workspace is F times
   * gjx_workspace_bytes(GJX_OP_SMC, n).  A 1e6-particle step is ~1000 workgroups, under one round of an
   * MI355X: a few filters per launch fill it.  Each filter equals its own single run bit for bit. */
  int32_t n_filters;
  uint64_t filter_stride;
  /* ESS-adaptive resampling (not in the reference, which has no resampling step at all).
   * 0 (unset) or >= 1: systematic resampling before every step t >= 1 (the BASELINE configs).  In (0, 1): the
   * population of step t-1 is resampled only if its effective sample size is below ess_threshold * n_total;
   * otherwise every particle keeps its own ancestor (ancestors[t][j] = j), and its log-weight ACCUMULATES:
   * logw_t[j] = logw_{t-1}[j] + increment.  ESS is evaluated on exact integers so that every backend, tiling and
   * number of ranks takes the same decision: per tile r_i = q_i >> 14 (the top 16 bits of the tile-anchored weight),
   * R1_t = sum r_i, R2_t = sum r_i^2 (gjx_tile_ess); merged R1 = sum_t R1_t >> d_t, R2 = sum_t R2_t >> 2 d_t;
   * resample iff (double)R1 * (double)R1 < (ess_threshold * n_total) * (double)R2.
   * log Z = sum over the steps t that END an epoch (a resampling follows, or t = T-1) of
   * (e_t ln 2 + log(q_t 2^-30) - log N): the per-step (out_e, out_q) pairs are those of the accumulated weights.
   * resampled_out: nullable dev int32[T] ([F, T] for F filters): entry t = 1 if step t began with a resampling
   * (entry 0 is 0).  Required (non-NULL) when ess_threshold is in (0, 1). */
  float ess_threshold;
  int32_t* resampled_out;
  /* r04 (nullable; host): the source population of the steps t >= 1 is distributed over the arenas of peers (above).  One
   * filter, n_local = n_total / world, first_slot = rank * n_local, n_total a multiple of world * gjx_smc_tile(). */
  const gjx_smc_peers* peers;
} gjx_smc_config;

/* A population between two steps: what a step reads of the previous one and writes for the next. */
#define GJX_SMC_MAX_STATE 4
#define GJX_SMC_MAX_OBS 8
typedef struct {
  void* state[GJX_SMC_MAX_STATE]; /* 4-byte columns (f32 x / int32 z) */
  uint32_t* qw;                   /* tile-anchored fixed-point weights (gjx_tile_rec above) */
  float* logw;                    /* log-weights: nullable unless the filter is ESS-adaptive */
  gjx_tile_rec* recs;             /* GLOBAL dev [gjx_num_tiles(n_total)]: the tiles' records */
  gjx_tile_sub* subs;             /* GLOBAL dev [gjx_num_tiles(n_total)]: their sub-prefixes */
  gjx_tile_ess* ess;              /* GLOBAL dev [gjx_num_tiles(n_total)]: their ESS sums (adaptive filters only) */
  uint64_t* prefix;               /* scratch dev u64[gjx_num_tiles(n_total) + 4]: required only when the population READ by a
                                     step has more than 1024 tiles (n_total > 2^20) — the step then merges its records
                                     into this array with one extra small launch instead of in every workgroup */
} gjx_smc_pop;

/* Single-device whole run (first_slot = 0, n_local = n_total).  y: HOST array [T] (f32 for lgssm,
 * int32 for hmm) — observations are baked into the launches.
 * Outputs: out_e dev int32[T], out_q dev u64[T] (per-step merged anchor and total mass of the
 * incremental log-weights => log Z = sum_t (e_t ln 2 + log(q_t 2^-30) - log N)); state_out dev [n] final-step
 * particles (f32 x / int32 z) and logw_out dev f32[n] their weights (before the final resampling);
 * ancestors_out dev int32[T,n] or NULL (row 0 is the identity). */
int gjx_smc_run_lgssm(const gjx_smc_config* cfg, const gjx_lgssm* model, const float* y_host,
                      int32_t* out_e, uint64_t* out_q, float* state_out, float* logw_out,
                      int32_t* ancestors_out, void* ws, size_t ws_bytes, gjx_stream s);
int gjx_smc_run_hmm(const gjx_smc_config* cfg, const gjx_hmm* model, const int32_t* y_host,
                    int32_t* out_e, uint64_t* out_q, int32_t* state_out, float* logw_out,
                    int32_t* ancestors_out, void* ws, size_t ws_bytes, gjx_stream s);

/* One step as ONE launch (the whole-run calls are loops over these; the multi-device driver runs the
 * exchange between them).  step (t): for every slot j in [first_slot, first_slot+n_local): systematic-resampling
 * ancestor from the GLOBAL previous population `prev` (state / qw / logw dev [n_total], recs / subs / ess GLOBAL),
 * propagate, weight; writes `out`: state / qw / logw (nullable) dev [n_local] — LOCAL arrays, slot j at j - first_slot —
 * and the records of the rank's own tiles into the GLOBAL arrays out->recs / subs / ess (at the global tile index; they
 * must not alias prev's: ranks all-gather them).  prev_e_out / prev_q_out (nullable dev [1]):
 * the merged anchor and total mass of prev's weights.  ancestors_out nullable dev int32[n_local].  t == 0 ignores
 * prev.  Only the source tiles that own one of the rank's slots are read (gjx_smc_source_ranges).
 *  finish: e_out[0], q_out[0] = merged anchor / total mass of the last step's records. */
int gjx_smc_lgssm_step(const gjx_smc_config* cfg, const gjx_lgssm* model, int t, float y_t,
                       const gjx_smc_pop* prev, const gjx_smc_pop* out, int32_t* prev_e_out, uint64_t* prev_q_out,
                       int32_t* ancestors_out, gjx_stream s);
int gjx_smc_hmm_step(const gjx_smc_config* cfg, const gjx_hmm* model, int t, int32_t y_t,
                     const gjx_smc_pop* prev, const gjx_smc_pop* out, int32_t* prev_e_out, uint64_t* prev_q_out,
                     const uint32_t* trans_alias /* dev, from gjx_hmm_prepare */,
                     const float* obs_logp /* dev f32[K,K] from gjx_hmm_prepare */,
                     int32_t* ancestors_out, gjx_stream s);
int gjx_smc_finish(const gjx_smc_config* cfg, const gjx_tile_rec* recs, int32_t* e_out, uint64_t* q_out,
                   gjx_stream s);
/* A rank with n_local < n_total reads, in a step, ONLY the source tiles that own one of its slots (which tiles
 * those are follows from the records and the comb offset alone), so between steps it needs just that part
 * of the other ranks' particles.  source_ranges: for each of `world` equal contiguous blocks of output slots
 * (block j = slots [j n_total/world, (j+1) n_total/world)), out_ranges[2j], [2j+1] = the half-open range of
 * source tiles that can own a slot of the block at the next resampling — the exact range or one tile more at
 * either end (the comb offset is bounded, not derived, so the ranges of a step are known before its key is
 * used).  Ancestors are monotone in the slot, hence one contiguous range per block.  ess: the population's ESS sums
 * (adaptive filters: a step that keeps its particles needs no exchange: identity ranges).  out_ranges:
 * int64[2 world + 1], device memory or device-visible pinned host memory; out_ranges[2 world] = ticket is
 * stored LAST with a system-scope release, so a host that owns a pinned buffer can poll for its ticket and
 * read the ranges without synchronising the stream.  world <= 64 and n_total a multiple of world. */
int gjx_smc_source_ranges(const gjx_smc_config* cfg, const gjx_tile_rec* recs, const gjx_tile_ess* ess, int world,
                          int64_t ticket, int64_t* out_ranges, gjx_stream s);
/* ---- bootstrap SMC for a user model: init sites + step sites as plans ------------------------ *
 * The general form of the two fixed models above: x_0 comes from `init_sites`, every later step
 * walks `step_sites` for each output slot with GJX_ARG_STATE arguments reading the resampled
 * ancestor's state columns; observed sites contribute their log-density to the step's log-weight.
 * Slot key = split(step_key)[slot].  THREEFRY: sites draw from it exactly as in gjx_importance_run
 * (fold_in(., 1-based table position)).  PHILOX: single-word draw number f (0-based index among the
 * sampled sites) of slots 4g .. 4g+3 is ONE block, PH(ctr = (g_lo, g_hi, f, 'Q'), key = step key), slot j
 * taking word j & 3, and Normal sites pair Box-Muller inside the quad ((w0,w1) -> slots 4g, 4g+1; (w2,w3)
 * -> 4g+2, 4g+3) — the fixed models above are the case of one sampled site; multi-word samplers (gamma,
 * beta, Gumbel-max categorical) keep the slot key's streams.  libgjx_hip.so lowers the step to a hiprtc-compiled policy
 * inside the fused resample kernel.  State columns are f32 (integer-valued sites are converted).
 * r03: with the compiler switched off (GJX_PLAN_JIT=0), or failed and GJX_PLAN_JIT_FALLBACK=1, a filter runs through a
 * table-walking policy inside the same kernel (the site tables read at run time: the same bits, several times slower);
 * a model that holds programs (GJX_ARG_EXPR) has no such route (GJX_ERR_UNSUPPORTED / GJX_ERR_JIT). */
typedef struct {
  const gjx_site* init_sites;
  int32_t n_init_sites;
  const gjx_site* step_sites;
  int32_t n_step_sites;
  gjx_arg init_state[GJX_SMC_MAX_STATE]; /* state k after step 0, over the init sites (CONST/SITE/OBS) */
  gjx_arg next_state[GJX_SMC_MAX_STATE]; /* state k after a step (CONST/SITE/STATE/OBS) */
  int32_t n_state;
  int32_t n_obs;
} gjx_smc_model;
typedef struct gjx_smc_plan gjx_smc_plan;
int gjx_smc_plan_create(const gjx_smc_model* m /*host*/, gjx_smc_plan** out);
/* The same with nested `@gen` calls inside `init` / `step` (scopes over the two site tables, as gjx_plan_create_scoped): a
 * callee of slot j runs under fold_in(slot key, the counter its call took) — a lone key, so its draws are its own blocks
 * (the quad blocks serve the body's own sites only).  Compiled kernels only: the table-walking policy refuses such filters. */
int gjx_smc_plan_create_scoped(const gjx_smc_model* m /*host*/, const gjx_scope* init_scopes /*host*/, int n_init_scopes,
                               const gjx_scope* step_scopes /*host*/, int n_step_scopes, gjx_smc_plan** out);
int gjx_smc_plan_destroy(gjx_smc_plan* p);
int gjx_smc_plan_compile_check(const gjx_smc_plan* p, int impl); /* offline hiprtc compile, needs no GPU */
/* obs_host: host f32[T, n_obs].  state_out: host array of n_state dev f32[n] pointers (final-step
 * particles); other outputs as gjx_smc_run_lgssm, gjx_smc_config.n_filters included (F filters of the same
 * model and observations with their own keys step in the same launches: every state column dev f32[F, stride]).
 * Single device (first_slot 0, n_local n_total). */
/* One step of a plan-driven filter as ONE launch (the per-step piece for a multi-device driver, like
 * gjx_smc_lgssm_step; finish / source ranges are the model-independent calls above): obs_t host f32[n_obs]; the
 * populations' state columns are the plan's n_state f32 columns. */
int gjx_smc_plan_step(const gjx_smc_config* cfg, gjx_smc_plan* plan, int t, const float* obs_t,
                      const gjx_smc_pop* prev, const gjx_smc_pop* out, int32_t* prev_e_out, uint64_t* prev_q_out,
                      int32_t* ancestors_out, gjx_stream s);
int gjx_smc_run_plan(const gjx_smc_config* cfg, gjx_smc_plan* plan, const float* obs_host,
                     int32_t* out_e, uint64_t* out_q, float* const* state_out, float* logw_out,
                     int32_t* ancestors_out, void* ws, size_t ws_bytes, gjx_stream s);

/* ---- importance over a `Scan` (state-space) model: the whole T-step walk in ONE launch --------- *
 * Replaces Scan.generate (generative_functions/combinators/scan.py:237-294) under ImportanceK (no resampling): particle
 * i runs `kernel(carry, x_t) -> (carry', y_t)` for t = 0 .. T-1; step t walks `step_sites` with key_t exactly as
 * gjx_importance_run walks a plan with that particle key.  THREEFRY: the CHAINED key key_t = fold_in(key_{t-1}, t),
 * key_{-1} = the particle key (scan.py:267-268, 276: the folded key is carried).  PHILOX (r03): no cipher block for a
 * key — key_t = (the particle's cipher key, its lane + (t + 1) 2^40); lanes < 2^40, T < 2^24 - 1 (GJX_ERR_INVALID
 * otherwise): a step's keys are laned keys like a population's, so particle pairs share blocks and Box-Muller.  GJX_ARG_STATE reads the carry, GJX_ARG_OBS
 * this step's row of `obs` (observed values and scanned inputs x_t alike), `next_state` gives the new carry.
 * weight = ((0 + w_0) + w_1) + ..., score likewise — the f32 sums of scan.py:290, 293 in time order.  The carry and
 * the key chain live in registers; every sampled value is stored TIME-MAJOR, value_cols[c][t * col_stride + i] (each
 * step's store is one coalesced row), the layout the reference's vmapped ScanTrace presents transposed as [N, T]. */
typedef struct {
  const gjx_site* step_sites;
  int32_t n_step_sites;
  gjx_arg next_state[GJX_SMC_MAX_STATE]; /* carry k after a step (CONST/SITE/STATE/OBS) */
  int32_t n_state;
  int32_t n_obs;
} gjx_scan_model;
typedef struct gjx_scan_plan gjx_scan_plan;
int gjx_scan_plan_create(const gjx_scan_model* m /*host*/, uint32_t flags /* GJX_PLAN_* */, gjx_scan_plan** out);
/* The same with nested `@gen` calls inside the step kernel (scopes over `step_sites`, as gjx_plan_create_scoped): a callee
 * of step t runs under fold_in(key_t, the counter its call took). */
int gjx_scan_plan_create_scoped(const gjx_scan_model* m /*host*/, const gjx_scope* scopes /*host*/, int n_scopes, uint32_t flags,
                                gjx_scan_plan** out);
int gjx_scan_plan_destroy(gjx_scan_plan* p);
int gjx_scan_plan_compile_check(const gjx_scan_plan* p, int impl); /* offline hiprtc compile, needs no GPU */
typedef struct {
  const gjx_keys* particle_keys;   /* before the chain; has_fold must be 0 */
  uint64_t n;                      /* particles */
  int32_t n_steps;                 /* T >= 1 */
  const float* obs;                /* dev f32[T, n_obs] (nullable iff n_obs == 0) */
  const float* carry0;             /* host f32[n_state]: the initial carry where carry0_cols[k] is null */
  const float* const* carry0_cols; /* nullable; host array of n_state nullable dev f32[n] (per-particle initial carry) */
  void* const* value_cols;         /* host array of dev [T, col_stride] 4-byte columns (f32 / int32), as gjx_importance_run */
  int32_t n_value_cols;
  uint64_t col_stride;             /* >= n */
  float* const* carry_out;         /* nullable; host array of n_state nullable dev f32[n]: the final carry */
  float* score;                    /* nullable dev f32[n] */
  float* logw;                     /* dev f32[n] */
  float* max_partials;             /* nullable, as gjx_importance_run */
  int32_t* row_e;                  /* nullable (with row_s), as gjx_importance_run */
  uint64_t* row_s;
  const gjx_lse_out* lse;          /* nullable; needs row_e / row_s */
} gjx_scan_io;
int gjx_scan_run(gjx_scan_plan* p, const gjx_scan_io* io, gjx_stream s);

/* ---- multi-GPU: a native communicator, the log-Z combine and the whole sharded filter ---------- *
 * One process per GPU.  gjx_comm wraps the transport the sharded calls below use:
 *   RCCL  (libgjx_hip.so): gjx_comm_unique_id on rank 0, broadcast the 128 bytes out of band, gjx_comm_init_rccl on
 *         every rank (the RCCL library is loaded on first use: processes that never shard do not need it);
 *   local (both builds): `world` VIRTUAL ranks — threads of ONE process sharing one device and one stream — through a
 *         gjx_comm_group; collectives are host barriers + device copies on the shared stream.  Test transport: it runs
 *         the sharded protocol on the HIP kernels of a one-GPU box and on the CPU oracle.
 * The collectives are not exposed one by one: what a caller of this path needs are the two operations built on them. */
typedef struct gjx_comm gjx_comm;
typedef struct gjx_comm_group gjx_comm_group;
#define GJX_COMM_ID_BYTES 128
int gjx_comm_unique_id(void* id_out /*host, GJX_COMM_ID_BYTES*/);
int gjx_comm_init_rccl(const void* id /*host*/, int rank, int world, gjx_comm** out);
int gjx_comm_group_create(int world, gjx_comm_group** out);
int gjx_comm_group_destroy(gjx_comm_group* g);
int gjx_comm_init_local(gjx_comm_group* g, int rank, gjx_comm** out);
/*   callbacks (both builds): the collectives are the CALLER's — the all-gather and the grouped send/recv of whatever
 *         process group the host program already has (torch.distributed over gloo or RCCL, MPI, ...), so the native driver
 *         runs across REAL processes on any transport; this is how gjx_smc_sharded_run_* is tested with world_size-2/4
 *         gloo process groups on CPU.  allgather: in place, rank r's block is full + r * bytes_per_rank.  exchange: element
 *         ranges [a, b) of PARTICLES (multiples of the tile) keep their global position on both sides; column c holds one
 *         element of elem_bytes[c] bytes per units[c] particles (1: a per-particle column, the tile size: the per-tile
 *         sub-prefixes, which travel with the shuffle since r04), so its slice is the bytes [(a / units[c]) elem_bytes[c],
 *         (b / units[c]) elem_bytes[c]) of cols[c]; both sides list the same ranges.  Callbacks return 0 or a negative gjx
 *         status. */
typedef struct {
  int32_t peer;
  uint64_t a, b;
} gjx_seg;
typedef int (*gjx_allgather_fn)(void* user, void* full, uint64_t bytes_per_rank, gjx_stream s);
typedef int (*gjx_exchange_fn)(void* user, void* const* cols, const uint64_t* elem_bytes, const uint64_t* units, int32_t n_cols, const gjx_seg* sends,
                               int32_t n_sends, const gjx_seg* recvs, int32_t n_recvs, gjx_stream s);
typedef int (*gjx_stream_sync_fn)(void* user, gjx_stream s);
int gjx_comm_init_callbacks(int rank, int world, gjx_allgather_fn allgather, gjx_exchange_fn exchange,
                            gjx_stream_sync_fn stream_sync, void* user, gjx_comm** out);
/*   peers (both builds; r04): no collective at all.  Every rank's populations live in an arena of identical layout that its
 *         peers can address (gjx_smc_peers above); per step a rank issues its ONE step launch (which waits, bounded, for
 *         its peers' arrival words and reads remote source windows where they live) and ONE small launch that deposits the
 *         records of its tiles in every peer's arena and then raises its arrival word there (gjx_smc_peer_signal): no
 *         RCCL call, no range kernel, no host poll — the host enqueues all T steps without waiting for anything.
 *         `group` (nullable): the ranks are VIRTUAL ranks of one process sharing one stream (tests; gjx_comm_group_create):
 *         launches of one stream execute in enqueue order, so the driver holds a host barrier between "every rank has
 *         enqueued its signal" and "any rank enqueues the launch that waits for it".  wait_launch != 0: a one-workgroup
 *         wait launch in front of every step launch — for ranks that SHARE a device as separate processes (tests), so that
 *         a step's workgroups never occupy the device while the signal they wait for still needs it. */
int gjx_comm_init_peers(const gjx_smc_peers* peers /*host; copied*/, gjx_comm_group* group, int wait_launch, gjx_comm** out);
/* The two launches the peer transport is made of (the sharded drivers call them; exposed for drivers written elsewhere).
 *  signal: deposit the records (and ESS sums when `ess` is given) of this rank's tiles [first_tile, first_tile + n_tiles) —
 *          read from recs / ess in this rank's arena — at the same place in every peer's arena, make them and everything
 *          earlier launches of the stream wrote visible at system scope, then store `value` into word `rank` of every
 *          rank's flags.  recs may be NULL (n_tiles 0): only the arrival word is raised.
 *  wait:   a one-workgroup launch that returns when every flags[q] >= value (or sets *error after the timeout). */
int gjx_smc_peer_signal(const gjx_smc_peers* peers, const gjx_tile_rec* recs, const gjx_tile_ess* ess, uint64_t first_tile,
                        uint64_t n_tiles, uint64_t value, gjx_stream s);
int gjx_smc_peer_wait(const gjx_smc_peers* peers, uint64_t value, gjx_stream s);
/* 1 if a step t >= 1 of this configuration absorbs a deferred signal (gjx_smc_peers.signal_*) into its first launch. */
int gjx_smc_peer_signal_fused(const gjx_smc_config* cfg);
int gjx_comm_destroy(gjx_comm* c);
int gjx_comm_rank(const gjx_comm* c);
int gjx_comm_world(const gjx_comm* c);
/* Global log-marginal of n_batch importance passes sharded over the ranks (replaces logsumexp(lw) - log K of
 * inference/smc.py:97 across devices): every rank passes the 65-word records of ITS shard (gjx_lse_rows out_record,
 * dev u64[n_batch, GJX_LSE_RECORD_WORDS]); one all-gather into `gathered` (dev u64[world, n_batch, 65], caller-owned)
 * and gjx_lse_combine give every rank the same (e, q, lse) per pass — the bits of ONE fold over all rows. */
int gjx_comm_lse_combine(gjx_comm* c, const uint64_t* records, int32_t n_batch, uint64_t* gathered, int32_t* out_e,
                         uint64_t* out_q, float* out_lse, gjx_stream s);
/* The whole bootstrap filter sharded over the communicator's ranks (BASELINE configs[3]): per step ONE launch (own
 * slots) -> ONE all-gather of the tile records (r04: recs and — adaptive filters — ess packed into one message, 16-32 bytes
 * per 1024 particles; the sub-prefixes travel with the shuffle, for the requested range only; no all-reduce) -> ancestor
 * shuffle, driven from C: no interpreter
 * between the launches.  cfg: first_slot / n_local = this rank's block (n_total a multiple of world * gjx_smc_tile()),
 * one filter.  All arrays are GLOBAL-size device buffers the caller owns (a rank's own block is always current; remote
 * ranges are filled by the shuffle): pop[2] (double-buffered: state columns dev 4-byte [n_total], qw dev u32[n_total],
 * logw dev f32[n_total] (adaptive filters; else only the final population's, nullable), recs / subs / ess dev [tiles]),
 * out_e dev int32[T], out_q dev u64[T], ancestors nullable dev int32[T, n_local],
 *   ranges int64[2 world + 1] device-visible PINNED host memory (plain host memory in the oracle build).
 * shuffle 0 = by source ranges (each rank receives exactly the contiguous range its slots draw from, in place, by
 * grouped send/recv; the host polls the range kernel's ticket in `ranges`), 1 = all-gather of the population.
 * *received (nullable): particles this rank received over the run (written on every exit path).  Results equal the
 * single-device filter bit for bit for every world size. */
typedef struct {
  gjx_smc_pop pop[2];
  int32_t* out_e;
  uint64_t* out_q;
  int32_t* ancestors;
  int64_t* ranges;
  int32_t shuffle;
  uint64_t* received;
  void* stage;       /* r04: dev [world, tiles_local, 32 bytes] — ESS-adaptive filters on a collective transport (world > 1): a
                        rank's records and ESS sums are packed into ONE message per step; NULL otherwise / for the peer transport */
} gjx_sharded_io;
/* pack (unpack = 0): this rank's block of recs / ess -> its slot of `stage`; unpack (1): every OTHER rank's slot of `stage` ->
 * recs / ess.  world equal blocks of cfg->n_local / gjx_smc_tile() tiles; the rank is cfg->first_slot / cfg->n_local. */
int gjx_smc_records_pack(const gjx_smc_config* cfg, int world, int unpack, gjx_tile_rec* recs, gjx_tile_ess* ess, void* stage,
                         gjx_stream s);
int gjx_smc_sharded_run_lgssm(gjx_comm* c, const gjx_smc_config* cfg, const gjx_lgssm* model, const float* y_host,
                              const gjx_sharded_io* io, gjx_stream s);
int gjx_smc_sharded_run_hmm(gjx_comm* c, const gjx_smc_config* cfg, const gjx_hmm* model, const int32_t* y_host,
                            const uint32_t* trans_alias, const float* obs_logp, const gjx_sharded_io* io, gjx_stream s);
int gjx_smc_sharded_run_plan(gjx_comm* c, const gjx_smc_config* cfg, gjx_smc_plan* plan, const float* obs_host,
                             const gjx_sharded_io* io, gjx_stream s);

/* HMM tables.  trans_alias: dev u32[gjx_hmm_alias_words(K)] = K rows of K packed alias-table entries
 * (threshold24 << 8) | alias built from the row's fixed-point softmax weights: the next state
 * from 32 random bits is column = floor(bits K / 2^32) if the next 24 bits of bits*K are below the column's
 * threshold, else the column's alias — one 4-byte table load per draw (the HMM step is bound by the rate of
 * scattered table loads).  obs_logp dev f32[K,K] = log_softmax rows of obs_logits.  K <= 256. */
uint64_t gjx_hmm_alias_words(int32_t n_states);
int gjx_hmm_prepare(const gjx_hmm* model, uint32_t* trans_alias, float* obs_logp, gjx_stream s);
uint64_t gjx_smc_tile(void); /* particles per tile */

#ifdef __cplusplus
}
#endif
#endif /* GJX_H */
