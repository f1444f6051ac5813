// gjx_plan_jit.hpp — plan specialisation: one straight-line gfx950 kernel per site table.
//
// The interpreter kernel (k_importance) pays ~200 scalar instructions and ~28 scalar loads per
// site to decode the table (rocprofv3 PMC, profiles/r01_b_*).  A static model's table is known when
// the plan is created, so we emit the walk of static.py:340-399 as straight-line HIP — the same
// device functions from gjx_device.hpp in the same order, with every constant folded into a literal
// — and compile it for gfx950 with hiprtc.  Both kernels implement the one arithmetic spec and are
// bit-identical (tests/test_gpu_parity_abi.py runs each plan both ways).
//
// This is included by gjx_hip.hip inside its anonymous namespace users; it needs CSite / gjx_plan.
#pragma once
#include <hip/hiprtc.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <sstream>
#include <string>
#include <unordered_map>
#include <utility>

namespace gjx_jit {

// The device header, embedded at build time (see __graft_entry__.build()).
static const char kDeviceHeader[] =
#include "gjx_device_embed.inc"
    ;

inline std::string flit(float f) {
  char b[48];
  std::snprintf(b, sizeof b, "u2f(0x%08xu)", gjx::f2u(f));
  return b;
}
inline std::string plit(const void* p) {
  char b[64];
  std::snprintf(b, sizeof b, "((const float*)0x%llxull)", (unsigned long long)(uintptr_t)p);
  return b;
}

// Emits the walk of one site table as straight-line HIP, in exactly the interpreter's operation
// order.  mode 0 = importance (particle index `i`, particle key `pkey`, input columns, score and
// output columns); mode 1 = SMC step / init (slot `j`, `a.step_key`, ancestor state `st_k`,
// observation constants `a.obs[]`, weight only).
template <class CSiteT, class CArgT>
struct SiteEmitter {
  std::ostringstream& o;
  int impl, mode;
  const CSiteT* sites;
  int n_sites;
  const char* ind;

  static bool is_int(const CSiteT& s) { return s.dist >= GJX_DIST_BERNOULLI; }
  std::string val_f32(int site) const {
    return is_int(sites[site]) ? "(float)vi" + std::to_string(site) : "vf" + std::to_string(site);
  }
  std::string val_i32(int site) const {
    return is_int(sites[site]) ? "vi" + std::to_string(site)
                               : "(int32_t)__builtin_rintf(vf" + std::to_string(site) + ")";
  }
  std::string arg(const CArgT& a) const {
    switch (a.kind) {
      case GJX_ARG_CONST: return flit(a.offset);
      case GJX_ARG_SITE: return "((" + flit(a.scale) + " * " + val_f32(a.ref_site) + ") + " + flit(a.offset) + ")";
      case GJX_ARG_INPUT:
        return "((" + flit(a.scale) + " * cols.in[" + std::to_string(a.ref) + "][i]) + " + flit(a.offset) + ")";
      case GJX_ARG_STATE: return "((" + flit(a.scale) + " * st_" + std::to_string(a.ref) + ") + " + flit(a.offset) + ")";
      case GJX_ARG_OBS: return "((" + flit(a.scale) + " * a.obs[" + std::to_string(a.ref) + "]) + " + flit(a.offset) + ")";
      default: return plit(a.table) + "[" + val_i32(a.ref_site) + "]";
    }
  }
  // does any site draw (and so need the per-particle key)?
  bool needs_pk() const {
    for (int q = 0; q < n_sites; ++q)
      if (!sites[q].observed) return true;
    return false;
  }
  // fold of site q: THREEFRY the 1-based site counter; PHILOX the 0-based index among the sampled sites
  uint32_t fold_of(int q) const {
    if (impl == 0) return (uint32_t)(q + 1);
    uint32_t d = 0;
    for (int p = 0; p < q; ++p) d += sites[p].observed ? 0u : 1u;
    return d;
  }

  void run() {
    const std::string I = std::to_string(impl);
    int cur_blk = -1;
    for (int q = 0; q < n_sites; ++q) {
      const CSiteT& st = sites[q];
      const std::string Q = std::to_string(q);
      const uint32_t fold = fold_of(q);
      o << ind << "// site " << q << " dist " << st.dist << (st.observed ? " observed" : " latent") << "\n";
      std::string row;
      if (st.dist == GJX_DIST_CATEGORICAL) {
        std::string rr;
        if (st.a0.kind == GJX_ARG_SITE) rr = val_i32(st.a0.ref_site);
        else if (st.a0.kind == GJX_ARG_CONST) rr = "(int32_t)__builtin_rintf(" + flit(st.a0.offset) + ")";
        else rr = "(int32_t)__builtin_rintf(" + arg(st.a0) + ")";
        o << ind << "int32_t rr" << Q << " = " << rr << "; rr" << Q << " = rr" << Q << " < 0 ? 0 : (rr" << Q
          << " >= " << st.n_rows << " ? " << st.n_rows - 1 << " : rr" << Q << ");\n";
        o << ind << "const float* row" << Q << " = " << plit(st.logits) << " + (size_t)rr" << Q << " * " << st.n_cat << ";\n";
        row = "row" + Q;
      } else {
        o << ind << "const float a0_" << Q << " = " << arg(st.a0) << ";\n";
        if (st.dist != GJX_DIST_BERNOULLI) o << ind << "const float a1_" << Q << " = " << arg(st.a1) << ";\n";
      }
      const bool isint = is_int(st);
      if (st.observed) {
        std::string ov;
        if (st.obs.kind == GJX_ARG_CONST) ov = flit(st.obs.offset);
        else if (st.obs.kind == GJX_ARG_OBS) ov = "a.obs[" + std::to_string(st.obs.ref) + "]";
        else ov = "cols.in[" + std::to_string(st.obs.ref) + "][i]";
        if (isint) o << ind << "const int32_t vi" << Q << " = (int32_t)__builtin_rintf(" << ov << ");\n";
        else o << ind << "const float vf" << Q << " = " << ov << ";\n";
      } else {
        const bool one_word = st.dist == GJX_DIST_NORMAL || st.dist == GJX_DIST_BERNOULLI ||
                              (st.dist == GJX_DIST_CATEGORICAL && st.cat_mode == 1);
        if (one_word) {
          if (impl == 1) {
            // word fold&3 of the packed draw block fold>>2 of the particle / slot key
            const int blk = (int)(fold >> 2);
            const uint32_t word = fold & 3u;
            if (blk != cur_blk) {
              cur_blk = blk;
              o << ind << "uint32_t pw" << blk << "_0, pw" << blk << "_1, pw" << blk << "_2, pw" << blk << "_3;\n";
              o << ind << "philox4x32(pkey.k0, pkey.k1, pkey.l0, pkey.l1, " << blk << "u, kTagDraw, pw" << blk << "_0, pw" << blk
                << "_1, pw" << blk << "_2, pw" << blk << "_3);\n";
            }
            o << ind << "const uint32_t bits" << Q << " = pw" << blk << "_" << word << ";\n";
          } else {
            o << ind << "const uint32_t bits" << Q << " = Stream<0>(pkey, true, " << fold << "u).bits32(0);\n";
          }
        }
        switch (st.dist) {
          case GJX_DIST_NORMAL:
            o << ind << "const float t" << Q << " = a1_" << Q << " * std_normal(bits" << Q << ");\n";
            o << ind << "const float vf" << Q << " = a0_" << Q << " + t" << Q << ";\n";
            break;
          case GJX_DIST_BERNOULLI:
            o << ind << "const int32_t vi" << Q << " = uniform01(bits" << Q << ") < a0_" << Q << " ? 1 : 0;\n";
            break;
          case GJX_DIST_GAMMA:
            o << ind << "const Stream<" << I << "> strm" << Q << "(pkey, true, " << fold << "u);\n";
            o << ind << "const float vf" << Q << " = std_gamma<" << I << ">(strm" << Q << ", 0, a0_" << Q << ") / a1_" << Q << ";\n";
            break;
          case GJX_DIST_BETA:
            o << ind << "const Stream<" << I << "> strm" << Q << "(pkey, true, " << fold << "u);\n";
            o << ind << "const float g1_" << Q << " = std_gamma<" << I << ">(strm" << Q << ", 0, a0_" << Q << ");\n";
            o << ind << "const float g2_" << Q << " = std_gamma<" << I << ">(strm" << Q << ", 1, a1_" << Q << ");\n";
            o << ind << "const float vf" << Q << " = g1_" << Q << " / (g1_" << Q << " + g2_" << Q << ");\n";
            break;
          default:
            if (st.cat_mode == 0) {
              o << ind << "const Stream<" << I << "> strm" << Q << "(pkey, true, " << fold << "u);\n";
              o << ind << "const int32_t vi" << Q << " = jcat_gumbel<" << I << ">(" << row << ", " << st.n_cat << "u, strm" << Q << ");\n";
            } else {
              o << ind << "const int32_t vi" << Q << " = jcat_invcdf(" << row << ", " << st.n_cat << "u, bits" << Q << ");\n";
            }
        }
      }
      std::string lp;
      const std::string v = (isint ? "vi" : "vf") + Q;
      switch (st.dist) {
        case GJX_DIST_NORMAL:
          lp = st.pre ? "logpdf_normal_pre(" + v + ", a0_" + Q + ", " + flit(st.pre0) + ", " + flit(st.pre1) + ")"
                      : "logpdf_normal(" + v + ", a0_" + Q + ", a1_" + Q + ")";
          break;
        case GJX_DIST_GAMMA:
          lp = st.pre ? "logpdf_gamma_pre(" + v + ", a0_" + Q + ", a1_" + Q + ", " + flit(st.pre1) + ")"
                      : "logpdf_gamma(" + v + ", a0_" + Q + ", a1_" + Q + ")";
          break;
        case GJX_DIST_BETA:
          lp = st.pre ? "logpdf_beta_pre(" + v + ", a0_" + Q + ", a1_" + Q + ", " + flit(st.pre1) + ")"
                      : "logpdf_beta(" + v + ", a0_" + Q + ", a1_" + Q + ")";
          break;
        case GJX_DIST_BERNOULLI: lp = "logpdf_bernoulli(" + v + " != 0, a0_" + Q + ")"; break;
        default:
          lp = "((" + v + " < 0 || " + v + " >= " + std::to_string(st.n_cat) + ") ? -__builtin_inff() : " + row + "[" + v +
               "] - jrow_lse(" + row + ", " + std::to_string(st.n_cat) + "u))";
      }
      o << ind << "{ const float lp = " << lp << "; sc = sc + lp;" << (st.observed ? " w = w + lp;" : "") << " }\n";
      if (mode == 0 && st.out_col >= 0)
        o << ind << "reinterpret_cast<uint32_t*>(cols.out[" << st.out_col << "])[i] = "
          << (isint ? "(uint32_t)" + v : "f2u(" + v + ")") << ";\n";
    }
  }
};

inline void emit_prelude(std::ostringstream& o) {
  o << "#include \"gjx_device.hpp\"\nusing namespace gjx;\n";
  o << "__device__ __forceinline__ float jrow_max(const float* l, uint32_t K){ float m=l[0]; for(uint32_t c=1;c<K;++c) m = l[c]>m?l[c]:m; return m; }\n";
  o << "__device__ __forceinline__ float jrow_lse(const float* l, uint32_t K){ const float m=jrow_max(l,K); float acc=0.0f; for(uint32_t c=0;c<K;++c) acc = acc + m_exp(l[c]-m); return m + m_log(acc); }\n";
  o << "__device__ __forceinline__ int32_t jcat_invcdf(const float* l, uint32_t K, uint32_t bits){ const float m=jrow_max(l,K); uint64_t Q=0; for(uint32_t c=0;c<K;++c) Q += cat_fix(l[c],m); const uint64_t thr=((uint64_t)bits*Q)>>32; uint64_t C=0; for(uint32_t c=0;c<K;++c){ C += cat_fix(l[c],m); if (C>thr) return (int32_t)c; } return (int32_t)(K-1); }\n";
  o << "template <int IMPL> __device__ __forceinline__ int32_t jcat_gumbel(const float* l, uint32_t K, const Stream<IMPL>& st){ int32_t best=0; float bv=-__builtin_inff(); for(uint32_t c=0;c<K;++c){ const float v = l[c] + gumbel_from_bits(st.bits32(c)); if (v>bv || c==0){ bv=v; best=(int32_t)c; } } return best; }\n";
}

template <class CSiteT, class CArgT>
struct Gen {
  std::ostringstream o;
  int impl;
  const CSiteT* sites;
  int n_sites;
  int min_waves = 0;  // __launch_bounds__ waves-per-SIMD hint (0 = none)
  int rows_per_block = 1;
  bool laned = false;  // specialise for gjx_keys{mode 1, parent_lane 0} (PHILOX only)

  bool all_normal() const {
    for (int q = 0; q < n_sites; ++q) {
      if (sites[q].dist != GJX_DIST_NORMAL) return false;
      for (const CArgT* a : {&sites[q].a0, &sites[q].a1})
        if (a->kind == GJX_ARG_TABLE) return false;
    }
    return true;
  }
  std::string arg2(const CArgT& a) const {
    switch (a.kind) {
      case GJX_ARG_CONST: return "splat2(" + flit(a.offset) + ")";
      case GJX_ARG_SITE: return "((" + flit(a.scale) + " * vf" + std::to_string(a.ref_site) + ") + " + flit(a.offset) + ")";
      default:
        return "((" + flit(a.scale) + " * (f32x2){cols.in[" + std::to_string(a.ref) + "][i0], cols.in[" + std::to_string(a.ref) +
               "][i1]}) + " + flit(a.offset) + ")";
    }
  }
  // All-Normal plans: two particles per lane (rows r and r+1 of a 512-particle block) on packed f32.
  std::string run2() {
    const std::string I = std::to_string(impl);
    rows_per_block = 2;
    emit_prelude(o);
    o << "extern \"C\" __global__ __launch_bounds__(256) void gjx_plan_kernel_" << (impl == 0 ? "threefry" : "philox")
      << "(KeySrc ks, RunCols cols, float* score, float* logw, uint64_t n, float* max_partials, int32_t* row_e, uint64_t* row_s, LseTail tail) {\n";
    o << "  __shared__ float sh_red[4];\n  __shared__ uint64_t sh_sum[4];\n";
    o << "  for (uint64_t blk = blockIdx.x; blk * 512 < n; blk += gridDim.x) {\n";
    o << "    const uint64_t j0 = blk * 512 + threadIdx.x, j1 = j0 + 256;\n";
    o << "    const bool ok0 = j0 < n, ok1 = j1 < n;\n";
    o << "    const uint64_t i0 = ok0 ? j0 : n - 1, i1 = ok1 ? j1 : n - 1;  // surplus lanes redo the last particle, stores masked\n";
    if (laned)
      o << "    const uint64_t lnA = ks.first + i0 + 1u, lnB = ks.first + i1 + 1u;\n"
        << "    const Key pkA{ks.parent.k0, ks.parent.k1, (uint32_t)lnA, (uint32_t)(lnA >> 32)}, pkB{ks.parent.k0, ks.parent.k1, (uint32_t)lnB, (uint32_t)(lnB >> 32)};\n";
    else
      o << "    const Key pkA = key_at<" << I << ">(ks, i0), pkB = key_at<" << I << ">(ks, i1);\n";
    o << "    f32x2 w = splat2(0.0f), sc = splat2(0.0f);\n";
    int cur_blk = -1;
    for (int q = 0; q < n_sites; ++q) {
      const CSiteT& st = sites[q];
      const std::string Q = std::to_string(q);
      const uint32_t fold = SiteEmitter<CSiteT, CArgT>{o, impl, 0, sites, n_sites, ""}.fold_of(q);
      o << "    // site " << q << (st.observed ? " observed" : " latent") << "\n";
      o << "    const f32x2 a0_" << Q << " = " << arg2(st.a0) << ";\n";
      o << "    const f32x2 a1_" << Q << " = " << arg2(st.a1) << ";\n";
      if (st.observed) {
        if (st.obs.kind == GJX_ARG_CONST) o << "    const f32x2 vf" << Q << " = splat2(" << flit(st.obs.offset) << ");\n";
        else o << "    const f32x2 vf" << Q << " = (f32x2){cols.in[" << st.obs.ref << "][i0], cols.in[" << st.obs.ref << "][i1]};\n";
      } else {
        if (impl == 1) {
          const int b = (int)(fold >> 2);
          if (b != cur_blk) {
            cur_blk = b;
            for (const char* P : {"A", "B"}) {
              o << "    uint32_t pw" << P << b << "_0, pw" << P << b << "_1, pw" << P << b << "_2, pw" << P << b << "_3;\n";
              o << "    philox4x32(pk" << P << ".k0, pk" << P << ".k1, pk" << P << ".l0, pk" << P << ".l1, " << b << "u, kTagDraw, pw" << P << b << "_0, pw" << P << b
                << "_1, pw" << P << b << "_2, pw" << P << b << "_3);\n";
            }
          }
          o << "    const uint32_t bA" << Q << " = pwA" << b << "_" << (fold & 3u) << ", bB" << Q << " = pwB" << b << "_" << (fold & 3u) << ";\n";
        } else {
          o << "    const uint32_t bA" << Q << " = Stream<0>(pkA, true, " << fold << "u).bits32(0), bB" << Q
            << " = Stream<0>(pkB, true, " << fold << "u).bits32(0);\n";
        }
        o << "    const f32x2 t" << Q << " = a1_" << Q << " * std_normal2(bA" << Q << ", bB" << Q << ");\n";
        o << "    const f32x2 vf" << Q << " = a0_" << Q << " + t" << Q << ";\n";
      }
      const std::string lp = st.pre ? "logpdf_normal_pre2(vf" + Q + ", a0_" + Q + ", " + flit(st.pre0) + ", " + flit(st.pre1) + ")"
                                    : "logpdf_normal2(vf" + Q + ", a0_" + Q + ", a1_" + Q + ")";
      o << "    { const f32x2 lp = " << lp << "; sc = sc + lp;" << (st.observed ? " w = w + lp;" : "") << " }\n";
      if (st.out_col >= 0) {
        o << "    if (ok0) reinterpret_cast<float*>(cols.out[" << st.out_col << "])[j0] = vf" << Q << ".x;\n";
        o << "    if (ok1) reinterpret_cast<float*>(cols.out[" << st.out_col << "])[j1] = vf" << Q << ".y;\n";
      }
    }
    o << "    if (ok0) { logw[j0] = w.x; if (score) score[j0] = sc.x; }\n";
    o << "    if (ok1) { logw[j1] = w.y; if (score) score[j1] = sc.y; }\n";
    for (int r = 0; r < 2; ++r) {
      const std::string R = std::to_string(r), W = r == 0 ? "w.x" : "w.y", OK = r == 0 ? "ok0" : "ok1";
      o << "    if ((max_partials || row_e) && (blk * 512 + " << (r * 256) << ") < n) {\n";
      o << "      const float bm = block_max(" << OK << " ? " << W << " : -__builtin_inff(), sh_red);\n";
      o << "      if (max_partials && threadIdx.x == 0) max_partials[blk * 2 + " << R << "] = bm;\n";
      o << "      if (row_e) {\n        const int32_t eb = row_anchor(bm);\n";
      o << "        const uint64_t sb = block_sum(" << OK << " ? rowfix(" << W << ", eb) : 0, sh_sum);\n";
      o << "        if (threadIdx.x == 0) lse_store_row(row_e, row_s, blk * 2 + " << R << ", eb, sb, tail.tickets != nullptr);\n";
      o << "      }\n    }\n";
    }
    o << "  }\n  if (row_e) lse_tail(row_e, row_s, (n + 255) / 256, tail);\n}\n";
    return o.str();
  }

  std::string run() {
    const std::string I = std::to_string(impl);
    // Packed-f32 form (two particles per lane): bit-identical, but measured SLOWER on MI355X for the
    // 10-latent model (29.1 vs 26.8 us) — the kernel is bound by the integer cipher and by issue slots,
    // not by f32 throughput — so it is opt-in (GJX_JIT_PACKED=1) and kept for the parity tests.
    const char* e2 = std::getenv("GJX_JIT_PACKED");
    if (all_normal() && e2 && e2[0] == '1') return run2();
    emit_prelude(o);
    // One workgroup per 256-particle row (grid-stride): short blocks keep every SIMD's wave slots
    // full even at 1e6 particles (15 rows per lane), where a 4-row block would serialise its rows.
    o << "extern \"C\" __global__ __launch_bounds__(256" << (min_waves > 0 ? ", " + std::to_string(min_waves) : std::string())
      << ") void gjx_plan_kernel_" << (impl == 0 ? "threefry" : "philox")
      << "(KeySrc ks, RunCols cols, float* score, float* logw, uint64_t n, float* max_partials, int32_t* row_e, uint64_t* row_s, LseTail tail) {\n";
    o << "  __shared__ float sh_red[4];\n  __shared__ uint64_t sh_sum[4];\n";
    o << "  for (uint64_t row = blockIdx.x; row * 256 < n; row += gridDim.x) {\n";
    o << "    float tmax = -__builtin_inff();\n    bool live = false;\n";
    o << "    {\n";
    o << "      const uint64_t i = row * 256 + threadIdx.x;\n";
    o << "      if (i < n) {\n";
    if (laned)  // lazy children of a lane-0 PHILOX key: the cipher key is uniform over the launch
      o << "        const uint64_t lane = ks.first + i + 1u;\n        const Key pkey{ks.parent.k0, ks.parent.k1, (uint32_t)lane, (uint32_t)(lane >> 32)};\n";
    else
      o << "        const Key pkey = key_at<" << I << ">(ks, i);\n";
    o << "        float w = 0.0f, sc = 0.0f;\n";
    SiteEmitter<CSiteT, CArgT> em{o, impl, 0, sites, n_sites, "        "};
    em.run();
    o << "        logw[i] = w;\n        if (score) score[i] = sc;\n        tmax = w;\n        live = true;\n";
    o << "      }\n    }\n";
    o << "    if (max_partials || row_e) {\n";
    o << "      const float bm = block_max(tmax, sh_red);\n";
    o << "      if (max_partials && threadIdx.x == 0) max_partials[row] = bm;\n";
    o << "      if (row_e) {\n";
    o << "        const int32_t eb = row_anchor(bm);\n";
    o << "        const uint64_t sb = block_sum(live ? rowfix(tmax, eb) : 0, sh_sum);\n";
    o << "        if (threadIdx.x == 0) lse_store_row(row_e, row_s, row, eb, sb, tail.tickets != nullptr);\n";
    o << "      }\n    }\n";
    o << "  }\n  if (row_e) lse_tail(row_e, row_s, (n + 255) / 256, tail);\n}\n";
    return o.str();
  }
};

// Plan-driven bootstrap SMC: a generated policy inside the fused resample kernel (step) and a plain
// per-slot kernel (init).  State columns are staged per source tile in LDS like the fixed models.
template <class CSiteT, class CArgT>
struct GenSmc {
  std::ostringstream o;
  int impl;
  const CSiteT* init_sites;
  int n_init;
  const CSiteT* step_sites;
  int n_step;
  const CArgT* init_state;
  const CArgT* next_state;
  int n_state;

  std::string run() {
    const std::string I = std::to_string(impl), D = std::to_string(n_state);
    emit_prelude(o);
    SiteEmitter<CSiteT, CArgT> es{o, impl, 1, step_sites, n_step, "    "};
    SiteEmitter<CSiteT, CArgT> ei{o, impl, 1, init_sites, n_init, "        "};
    // ---- step policy
    o << "struct GenPolicy {\n  PlanPolicyArgs a;\n  float* xs[" << D << "];\n  float xr[" << D << "][kPer];\n";
    o << "  struct Out { float s[" << D << "]; float lw; };\n";
    o << "  __device__ __forceinline__ void fetch_source(uint64_t base, uint64_t n, int tid) {\n";
    o << "    for (int k = 0; k < " << D << "; ++k)\n      for (int r = 0; r < kPer; ++r) { const uint64_t i = base + (uint64_t)r * 256 + tid; xr[k][r] = i < n ? a.prev_state[k][i] : 0.0f; }\n  }\n";
    o << "  __device__ __forceinline__ void stage_source(int tid) {\n    __shared__ float tile[" << D << "][kTile];\n";
    o << "    for (int k = 0; k < " << D << "; ++k) { xs[k] = tile[k]; for (int r = 0; r < kPer; ++r) tile[k][r * 256 + tid] = xr[k][r]; }\n  }\n";
    o << "  __device__ __forceinline__ float compute(int64_t j, int src_local, Out& out) const {\n";
    for (int k = 0; k < n_state; ++k) o << "    const float st_" << k << " = xs[" << k << "][src_local];\n";
    if (es.needs_pk()) o << "    const Key pkey = slot_key<" << I << ">(a.step_key, (uint64_t)j);\n";
    o << "    float w = 0.0f, sc = 0.0f;\n";
    es.run();
    for (int k = 0; k < n_state; ++k) o << "    out.s[" << k << "] = " << es.arg(next_state[k]) << ";\n";
    o << "    (void)sc;\n    out.lw = w;\n    return w;\n  }\n";
    o << "  __device__ __forceinline__ void store(int64_t j, int64_t out_lo, uint64_t src, const Out& out) const {\n";
    o << "    for (int k = 0; k < " << D << "; ++k) a.state_out[k][j - out_lo] = out.s[k];\n";
    o << "    a.logw_out[j - out_lo] = out.lw;\n    if (a.anc_out) a.anc_out[j - out_lo] = (int32_t)src;\n  }\n};\n";
    o << "extern \"C\" __global__ __launch_bounds__(256) void gjx_smc_step_kernel(ResampleArgs A, PlanPolicyArgs PA, float* max_partials) {\n";
    o << "  GenPolicy P;\n  P.a = PA;\n  resample_body<" << I << ">(A, P, max_partials);\n}\n";
    // ---- init kernel: one workgroup per global tile, like k_lgssm_init
    o << "extern \"C\" __global__ __launch_bounds__(256) void gjx_smc_init_kernel(PlanPolicyArgs a, uint64_t first_slot, uint64_t n_local, float* max_partials) {\n";
    o << "  __shared__ float shf[4];\n  const uint64_t gbase = (uint64_t)blockIdx.x * kTile;\n  float tmax = -__builtin_inff();\n";
    o << "  if (gbase >= first_slot && gbase < first_slot + n_local) {\n    for (int r = 0; r < kPer; ++r) {\n";
    o << "      const uint64_t j = gbase + (uint64_t)r * 256 + threadIdx.x;\n      if (j < first_slot + n_local) {\n";
    if (ei.needs_pk()) o << "        const Key pkey = slot_key<" << I << ">(a.step_key, j);\n";
    o << "        float w = 0.0f, sc = 0.0f;\n";
    ei.run();
    for (int k = 0; k < n_state; ++k) o << "        a.state_out[" << k << "][j - first_slot] = " << ei.arg(init_state[k]) << ";\n";
    o << "        (void)sc;\n        a.logw_out[j - first_slot] = w;\n        if (a.anc_out) a.anc_out[j - first_slot] = (int32_t)j;\n";
    o << "        tmax = w > tmax ? w : tmax;\n      }\n    }\n  }\n";
    o << "  const float bm = block_max(tmax, shf);\n  if (threadIdx.x == 0) max_partials[blockIdx.x] = bm;\n}\n";
    return o.str();
  }
};

struct Compiled {
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  int state = 0;  // 0 untried, 1 ready, -1 failed
  int rows_per_block = 1;  // 256-particle rows one workgroup processes per grid-stride iteration
};

inline bool enabled() {
  const char* e = std::getenv("GJX_PLAN_JIT");
  return !(e && e[0] == '0');
}

// Compile `src` for gfx950; on success `code` holds the code object.
inline bool compile_to_code(const std::string& src, std::string* code) {
  hiprtcProgram prog;
  const char* hn[] = {"gjx_device.hpp"};
  const char* hs[] = {kDeviceHeader};
  if (hiprtcCreateProgram(&prog, src.c_str(), "gjx_plan.hip", 1, hs, hn) != HIPRTC_SUCCESS) return false;
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17"};
  const hiprtcResult r = hiprtcCompileProgram(prog, 4, opts);
  if (r != HIPRTC_SUCCESS) {
    if (std::getenv("GJX_PLAN_JIT_VERBOSE")) {
      size_t ls = 0;
      hiprtcGetProgramLogSize(prog, &ls);
      std::string log(ls, 0);
      hiprtcGetProgramLog(prog, &log[0]);
      std::fprintf(stderr, "[gjx] plan specialisation failed to compile:\n%s\n", log.c_str());
    }
    hiprtcDestroyProgram(&prog);
    return false;
  }
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  code->assign(cs, 0);
  hiprtcGetCode(prog, &(*code)[0]);
  hiprtcDestroyProgram(&prog);
  return true;
}
inline bool compile_only(const std::string& src) {
  std::string code;
  return compile_to_code(src, &code) && !code.empty();
}
// ... and load it on the current device.  Identical sources (same model, same constants) share
// one module process-wide, so re-creating a plan does not recompile.
inline bool compile_module(const std::string& src, hipModule_t* mod_out) {
  static std::mutex mu;
  static std::unordered_map<std::string, hipModule_t> cache;
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(src);
  if (it != cache.end()) {
    *mod_out = it->second;
    return true;
  }
  std::string code;
  if (!compile_to_code(src, &code)) return false;
  hipModule_t mod = nullptr;
  if (hipModuleLoadData(&mod, code.data()) != hipSuccess) return false;
  cache.emplace(src, mod);
  *mod_out = mod;
  return true;
}
inline bool compile(const std::string& src, int impl, Compiled* out) {
  hipModule_t mod = nullptr;
  if (!compile_module(src, &mod)) return false;
  hipFunction_t fn = nullptr;
  if (hipModuleGetFunction(&fn, mod, impl == 0 ? "gjx_plan_kernel_threefry" : "gjx_plan_kernel_philox") != hipSuccess)
    return false;
  out->mod = nullptr;  // owned by the cache
  out->fn = fn;
  return true;
}
struct CompiledSmc {
  hipFunction_t step = nullptr, init = nullptr;
  int state = 0;  // 0 untried, 1 ready, -1 failed
};
inline bool compile_smc(const std::string& src, CompiledSmc* out) {
  hipModule_t mod = nullptr;
  if (!compile_module(src, &mod)) return false;
  if (hipModuleGetFunction(&out->step, mod, "gjx_smc_step_kernel") != hipSuccess) return false;
  if (hipModuleGetFunction(&out->init, mod, "gjx_smc_init_kernel") != hipSuccess) return false;
  return true;
}

}  // namespace gjx_jit
