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

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <sstream>
#include <string>
#include <unordered_map>
#include <cstring>
#include <utility>
#include <vector>

#include <dirent.h>
#include <dlfcn.h>
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cerrno>

extern "C" char** environ;

namespace gjx_jit {

// The device header, embedded at build time (see __graft_entry__.build()).
static const char kDeviceHeader[] =
#include "gjx_device_embed.inc"
    ;

inline std::string flit(float f) {
  char b[64];
  // (a NaN / infinite literal is kept out of constant propagation: gjx_device.hpp opq)
  std::snprintf(b, sizeof b, (f - f == 0.0f) ? "u2f(0x%08xu)" : "opq(u2f(0x%08xu))", gjx::f2u(f));
  return b;
}
// a finite literal that must still stay opaque (a parameter outside its domain, an observation outside the support)
inline std::string flit_opaque(float f) { return (f - f == 0.0f) ? "opq(" + flit(f) + ")" : flit(f); }
// Device pointers of a generated kernel: entry k of the `tabs` kernel argument (gjx_device.hpp PlanTables), numbered in
// order of first use while the source is generated — the SOURCE holds indices, never addresses, so it is the same for
// every plan of the same structure; the plan keeps the addresses and passes them at launch.  (Beyond kMaxPlanTables
// distinct tables — 48: never seen — the address is written into the source as before.)
struct TableReg {
  std::vector<const void*> ptrs;
  int index_of(const void* p) {
    for (size_t i = 0; i < ptrs.size(); ++i)
      if (ptrs[i] == p) return (int)i;
    if ((int)ptrs.size() >= gjx::kMaxPlanTables) return -1;
    ptrs.push_back(p);
    return (int)ptrs.size() - 1;
  }
  gjx::PlanTables tables() const {
    gjx::PlanTables t;
    std::memset(&t, 0, sizeof t);
    for (size_t i = 0; i < ptrs.size(); ++i) t.p[i] = ptrs[i];
    return t;
  }
};
inline TableReg*& active_tables() {
  static thread_local TableReg* reg = nullptr;
  return reg;
}
struct TableScope {  // the registry of one source generation
  TableReg reg;
  TableReg* prev;
  TableScope() : prev(active_tables()) { active_tables() = &reg; }
  ~TableScope() { active_tables() = prev; }
};
inline std::string plit_as(const char* type, const void* p) {
  char b[96];
  const int k = active_tables() ? active_tables()->index_of(p) : -1;
  if (k >= 0) std::snprintf(b, sizeof b, "((const %s*)tabs.p[%d])", type, k);
  else std::snprintf(b, sizeof b, "((const %s*)0x%llxull)", type, (unsigned long long)(uintptr_t)p);
  return b;
}
inline std::string plit(const void* p) { return plit_as("float", p); }

// Emits the walk of one site table as straight-line HIP, in exactly the interpreter's operation
// order.  mode 0 = importance (particle index `i`, particle key `pkey`, input columns, score and
// output columns); mode 1 = SMC step / init (slot `j`, `a.step_key`, ancestor state `st_k`,
// observation constants `a.obs[]`, weight only).  `sfx` is appended to every per-particle name, so two
// emitters ("A", "B") can interleave the walks of the two particles a lane owns.
// Nested `@gen` calls (gjx.h gjx_scope): which key every site draws under and with which fold.  Derived once per plan
// from the scopes' ranges: every `@` site and every call takes the next counter of the scope it sits in (THREEFRY: the
// 1-based counter over all of them; PHILOX: the 0-based index among those that consume randomness — unobserved sites and
// calls), a callee numbers its own sites afresh under fold_in(caller's key, the call's counter).
struct ScopeInfo {
  int n_scopes = 0;
  int site_scope[GJX_MAX_SITES];
  uint32_t fold_t[GJX_MAX_SITES], fold_p[GJX_MAX_SITES];
  int parent[GJX_MAX_SCOPES + 1], begin[GJX_MAX_SCOPES + 1], end[GJX_MAX_SCOPES + 1];
  uint32_t s_fold_t[GJX_MAX_SCOPES + 1], s_fold_p[GJX_MAX_SCOPES + 1];
};
inline bool derive_scopes(const gjx_site* sites, int n_sites, const gjx_scope* sc, int n_sc, ScopeInfo& out) {
  if (n_sc < 0 || n_sc > GJX_MAX_SCOPES || (n_sc && !sc)) return false;
  struct Frame { int id, end; uint32_t ct, dr; };
  Frame fr[8];
  int depth = 0, next = 0;
  fr[depth++] = Frame{0, n_sites, 1u, 0u};
  out.n_scopes = n_sc;
  out.parent[0] = -1;
  for (int q = 0; q <= n_sites; ++q) {
    while (next < n_sc && sc[next].begin == q) {  // the calls made at this position, in call order
      const gjx_scope& k = sc[next];
      if (k.parent < 0 || k.parent > next || k.end < k.begin || k.end > n_sites) return false;
      while (depth > 0 && fr[depth - 1].id != k.parent) {
        if (fr[depth - 1].end > q) return false;  // the caller is not the innermost open scope
        --depth;
      }
      if (depth == 0 || k.end > fr[depth - 1].end || depth >= 5) return false;
      out.parent[next + 1] = k.parent;
      out.begin[next + 1] = k.begin;
      out.end[next + 1] = k.end;
      out.s_fold_t[next + 1] = fr[depth - 1].ct++;
      out.s_fold_p[next + 1] = fr[depth - 1].dr++;
      fr[depth++] = Frame{next + 1, k.end, 1u, 0u};
      ++next;
    }
    if (next < n_sc && sc[next].begin < q) return false;  // not in call order
    while (depth > 1 && fr[depth - 1].end <= q) --depth;
    if (q == n_sites) break;
    out.site_scope[q] = fr[depth - 1].id;
    out.fold_t[q] = fr[depth - 1].ct++;
    out.fold_p[q] = fr[depth - 1].dr;
    if (!sites[q].observed) fr[depth - 1].dr++;
  }
  return next == n_sc;
}

template <class CSiteT, class CArgT>
struct SiteEmitter {
  std::ostringstream& o;
  int impl, mode;
  const CSiteT* sites;
  int n_sites;
  const char* ind;
  std::string sfx = "";
  int cur_blk = -1;
  bool store_values = true;  // false: the caller stores (the paired kernel writes both particles at once)
  bool ext_bits = false;     // true: the caller defines bits<q><sfx> of one-word draws (SMC quads share a block)
  const ScopeInfo* sc = nullptr;  // nested calls: per-site key scope and fold (null: a flat body, the implicit numbering)

  int scope_of(int q) const { return sc ? sc->site_scope[q] : 0; }
  std::string key_of_scope(int k) const { return k == 0 ? "pkey" + sfx : "skey" + std::to_string(k) + sfx; }
  std::string key_of(int q) const { return key_of_scope(scope_of(q)); }
  // the keys of the nested scopes, each fold_in(caller's key, the counter its call took); after pkey<sfx> is defined
  void emit_scope_keys() {
    if (!sc) return;
    for (int k = 1; k <= sc->n_scopes; ++k)
      o << ind << "const Key " << key_of_scope(k) << " = fold_in<" << impl << ">(" << key_of_scope(sc->parent[k]) << ", "
        << (impl == 0 ? sc->s_fold_t[k] : sc->s_fold_p[k]) << "u); (void)" << key_of_scope(k) << ";\n";
    for (int k = 1; k <= sc->n_scopes; ++k) o << ind << "float " << acc("w", k) << " = 0.0f, " << acc("sc", k) << " = 0.0f;\n";
  }

  static bool is_int(const CSiteT& s) { return s.dist >= GJX_DIST_BERNOULLI; }
  std::string nm(const char* base, int q) const { return std::string(base) + std::to_string(q) + sfx; }
  std::string val_f32(int site) const { return is_int(sites[site]) ? "(float)" + nm("vi", site) : nm("vf", site); }
  std::string val_i32(int site) const {
    return is_int(sites[site]) ? nm("vi", site) : "(int32_t)__builtin_rintf(" + nm("vf", site) + ")";
  }
  std::string arg(const CArgT& a) const {
    switch (a.kind) {
      case GJX_ARG_CONST: return flit(a.offset);
      case GJX_ARG_SITE: return "((" + flit(a.scale) + " * " + val_f32(a.ref_site) + ") + " + flit(a.offset) + ")";
      case GJX_ARG_INPUT:
        return "((" + flit(a.scale) + " * cols.in[" + std::to_string(a.ref) + "][li" + sfx + "]) + " + flit(a.offset) + ")";
      case GJX_ARG_STATE: return "((" + flit(a.scale) + " * st_" + std::to_string(a.ref) + sfx + ") + " + flit(a.offset) + ")";
      case GJX_ARG_OBS: return "((" + flit(a.scale) + " * a.obs[" + std::to_string(a.ref) + "]) + " + flit(a.offset) + ")";
      case GJX_ARG_PARAM: return "((" + flit(a.scale) + " * prm.p[" + std::to_string(a.ref) + "]) + " + flit(a.offset) + ")";
      case GJX_ARG_EXPR: {
        // the postfix program as ONE parenthesised f32 expression: every operator rounds once, in program order (the
        // translation unit is compiled with -ffp-contract=off: no fusion, no re-association)
        const gjx_expr_op* ops = reinterpret_cast<const gjx_expr_op*>(a.table);
        std::vector<std::string> st;
        for (int k = 0; k < a.ref; ++k) {
          const std::string r = std::to_string(ops[k].ref);
          switch (ops[k].op) {
            case GJX_EXPR_CONST: st.push_back(flit(ops[k].value)); break;
            case GJX_EXPR_SITE: st.push_back(val_f32(ops[k].ref)); break;
            case GJX_EXPR_INPUT: st.push_back("cols.in[" + r + "][li" + sfx + "]"); break;
            case GJX_EXPR_PARAM: st.push_back("prm.p[" + r + "]"); break;
            case GJX_EXPR_STATE: st.push_back("st_" + r + sfx); break;
            case GJX_EXPR_OBS: st.push_back("a.obs[" + r + "]"); break;
            case GJX_EXPR_NEG: st.back() = "(-" + st.back() + ")"; break;
            case GJX_EXPR_EXP: st.back() = "e_exp(" + st.back() + ")"; break;
            case GJX_EXPR_LOG: st.back() = "m_log(" + st.back() + ")"; break;
            case GJX_EXPR_SQRT: st.back() = "__builtin_sqrtf(" + st.back() + ")"; break;
            case GJX_EXPR_ABS: st.back() = "__builtin_fabsf(" + st.back() + ")"; break;
            case GJX_EXPR_LT: case GJX_EXPR_LE: case GJX_EXPR_EQ: {
              const std::string b = st.back();
              st.pop_back();
              const char* op = ops[k].op == GJX_EXPR_LT ? " < " : (ops[k].op == GJX_EXPR_LE ? " <= " : " == ");
              st.back() = "((" + st.back() + op + b + ") ? 1.0f : 0.0f)";
              break;
            }
            case GJX_EXPR_SELECT: {
              const std::string f = st.back();
              st.pop_back();
              const std::string t = st.back();
              st.pop_back();
              st.back() = "((" + st.back() + " != 0.0f) ? " + t + " : " + f + ")";
              break;
            }
            case GJX_EXPR_MAX: case GJX_EXPR_MIN: {
              const std::string b = st.back();
              st.pop_back();
              st.back() = std::string(ops[k].op == GJX_EXPR_MAX ? "e_max(" : "e_min(") + st.back() + ", " + b + ")";
              break;
            }
            default: {
              const std::string b = st.back();
              st.pop_back();
              const char* op = ops[k].op == GJX_EXPR_ADD ? " + " : (ops[k].op == GJX_EXPR_SUB ? " - " : (ops[k].op == GJX_EXPR_MUL ? " * " : " / "));
              st.back() = "(" + st.back() + op + b + ")";
            }
          }
        }
        return st.back();
      }
      default: return plit(a.table) + "[" + val_i32(a.ref_site) + "]";
    }
  }
  // does any site draw (and so need the per-particle key)?
  bool needs_pk() const {
    for (int q = 0; q < n_sites; ++q)
      if (!sites[q].observed) return true;
    return false;
  }
  bool needs_stream_key() const {
    for (int q = 0; q < n_sites; ++q)
      if (!sites[q].observed && !one_word(sites[q])) return true;
    return false;
  }
  // fold of site q: THREEFRY the 1-based site counter; PHILOX the 0-based index among the sampled sites
  uint32_t fold_of(int q) const {
    if (sc) return impl == 0 ? sc->fold_t[q] : sc->fold_p[q];
    if (impl == 0) return (uint32_t)(q + 1);
    uint32_t d = 0;
    for (int p = 0; p < q; ++p) d += sites[p].observed ? 0u : 1u;
    return d;
  }
  bool one_word(const CSiteT& st) const {
    return st.dist == GJX_DIST_NORMAL || st.dist == GJX_DIST_BERNOULLI ||
           (st.dist == GJX_DIST_CATEGORICAL && st.cat_mode == 1);
  }
  // Importance plans under PHILOX pair particles at Normal sites (gjx_device.hpp bm_pair / site_normal);
  // SMC steps keep the single-draw inverse-CDF form (one latent per slot-step: a pair would cost two blocks).
  // Scan steps (mode 2) are importance walks under the step's chained key: same draws as mode 0.
  bool pairs_normals() const { return impl == 1 && mode != 1; }

  // Part 1 of site q: arguments, the observed value, or the draw word(s).
  void head(int q) {
    const CSiteT& st = sites[q];
    const std::string Q = std::to_string(q) + sfx;
    const uint32_t fold = fold_of(q);
    o << ind << "// site " << q << sfx << " dist " << st.dist << (st.observed ? " observed" : " latent") << "\n";
    if (st.dist == GJX_DIST_CATEGORICAL) {
      std::string rr;
      if (st.a0.kind == GJX_ARG_SITE) rr = val_i32(st.a0.ref_site);
      else if (st.a0.kind == GJX_ARG_CONST) rr = "(int32_t)__builtin_rintf(" + flit(st.a0.offset) + ")";
      else rr = "(int32_t)__builtin_rintf(" + arg(st.a0) + ")";
      o << ind << "int32_t rr" << Q << " = " << rr << "; rr" << Q << " = rr" << Q << " < 0 ? 0 : (rr" << Q
        << " >= " << st.n_rows << " ? " << st.n_rows - 1 << " : rr" << Q << ");\n";
      o << ind << "const float* row" << Q << " = " << plit(st.logits) << " + (size_t)rr" << Q << " * " << st.n_cat << ";\n";
    } else {
      // a CONSTANT argument outside its domain (scale / rate / concentration <= 0, a probability outside [0, 1]) or an
      // observed constant outside the support would fold into a NaN / +inf constant log-density: kept opaque (opq)
      auto carg = [&](const CArgT& a, int which) {
        float v = a.offset;
        if (a.kind == GJX_ARG_EXPR) {  // a program over literals only is a constant too: its value, by the program's own steps
          const gjx_expr_op* ops = reinterpret_cast<const gjx_expr_op*>(a.table);
          float stk[8];
          int d = 0;
          for (int k = 0; k < a.ref; ++k) {
            if (ops[k].op >= GJX_EXPR_SITE && ops[k].op <= GJX_EXPR_OBS) return arg(a);  // (not a constant)
            if (ops[k].op == GJX_EXPR_CONST) { stk[d++] = ops[k].value; continue; }
            if (ops[k].op == GJX_EXPR_NEG) { stk[d - 1] = -stk[d - 1]; continue; }
            if (ops[k].op == GJX_EXPR_EXP) { stk[d - 1] = gjx::e_exp(stk[d - 1]); continue; }
            if (ops[k].op == GJX_EXPR_LOG) { stk[d - 1] = gjx::m_log(stk[d - 1]); continue; }
            if (ops[k].op == GJX_EXPR_SQRT) { stk[d - 1] = __builtin_sqrtf(stk[d - 1]); continue; }
            if (ops[k].op == GJX_EXPR_ABS) { stk[d - 1] = __builtin_fabsf(stk[d - 1]); continue; }
            if (ops[k].op == GJX_EXPR_SELECT) { const float r = stk[d - 3] != 0.0f ? stk[d - 2] : stk[d - 1]; d -= 2; stk[d - 1] = r; continue; }
            if (ops[k].op == GJX_EXPR_LT || ops[k].op == GJX_EXPR_LE || ops[k].op == GJX_EXPR_EQ) {
              const float y = stk[--d], x = stk[d - 1];
              stk[d - 1] = (ops[k].op == GJX_EXPR_LT ? x < y : (ops[k].op == GJX_EXPR_LE ? x <= y : x == y)) ? 1.0f : 0.0f;
              continue;
            }
            const float y = stk[--d], x = stk[d - 1];
            stk[d - 1] = ops[k].op == GJX_EXPR_ADD ? x + y : ops[k].op == GJX_EXPR_SUB ? x - y : ops[k].op == GJX_EXPR_MUL ? x * y
                         : ops[k].op == GJX_EXPR_MAX ? gjx::e_max(x, y) : ops[k].op == GJX_EXPR_MIN ? gjx::e_min(x, y) : x / y;
          }
          v = stk[0];
        } else if (a.kind != GJX_ARG_CONST) {
          return arg(a);
        }
        bool ok = v - v == 0.0f;
        if (st.dist == GJX_DIST_NORMAL) ok = ok && (which == 0 || v > 0.0f);
        else if (st.dist == GJX_DIST_BERNOULLI) ok = ok && v >= 0.0f && v <= 1.0f;
        else ok = ok && v > 0.0f;
        if (a.kind == GJX_ARG_EXPR) return ok ? arg(a) : "opq(" + arg(a) + ")";
        return ok ? flit(v) : flit_opaque(v);
      };
      o << ind << "const float a0_" << Q << " = " << carg(st.a0, 0) << ";\n";
      if (st.dist != GJX_DIST_BERNOULLI) o << ind << "const float a1_" << Q << " = " << carg(st.a1, 1) << ";\n";
    }
    if (st.observed) {
      std::string ov;
      if (st.obs.kind == GJX_ARG_CONST) {
        const float v = st.obs.offset;
        const bool in_support = st.dist == GJX_DIST_GAMMA ? v > 0.0f : (st.dist == GJX_DIST_BETA ? (v > 0.0f && v < 1.0f) : true);
        ov = in_support ? flit(v) : flit_opaque(v);
      }
      else if (st.obs.kind == GJX_ARG_OBS) ov = "a.obs[" + std::to_string(st.obs.ref) + "]";
      else if (st.obs.kind == GJX_ARG_PARAM) ov = arg(st.obs);
      else ov = "cols.in[" + std::to_string(st.obs.ref) + "][li" + sfx + "]";
      if (is_int(st)) o << ind << "const int32_t vi" << Q << " = (int32_t)__builtin_rintf(" << ov << ");\n";
      else o << ind << "const float vf" << Q << " = " << ov << ";\n";
      return;
    }
    if (!one_word(st) || (ext_bits && scope_of(q) == 0)) return;  // (a callee's sites draw under their own lone keys)
    if (impl == 1) {  // (the generic form: kernels that own whole pairs / quads define bits themselves, ext_bits)
      o << ind << "const uint32_t bits" << Q << " = philox_single_draw(" << key_of(q) << ", " << fold << "u);\n";
    } else {
      o << ind << "const uint32_t bits" << Q << " = Stream<0>(" << key_of(q) << ", true, " << fold << "u).bits32(0);\n";
    }
  }

  // Part 2 of site q: the sampled value (Normal sites take their standard normal from `eps`: an expression
  // or, empty, this particle's own derivation), the log-density, the accumulators and the stored column.
  bool presampled = false;  // tail(): a Gamma / Beta site's value vf<q> has been defined by the caller
  void tail(int q, const std::string& eps = "") {
    const CSiteT& st = sites[q];
    const std::string I = std::to_string(impl), Q = std::to_string(q) + sfx, K = key_of(q);
    const uint32_t fold = fold_of(q);
    const std::string row = "row" + Q;
    const bool isint = is_int(st);
    // (presampled: the pair / quad forms draw the gammas of a lane's particles together — std_gamma_multi — and define vf<q> themselves)
    if (!st.observed && !(presampled && (st.dist == GJX_DIST_GAMMA || st.dist == GJX_DIST_BETA))) {
      switch (st.dist) {
        case GJX_DIST_NORMAL: {
          std::string e = eps;
          if (e.empty())
            e = pairs_normals() ? "site_normal<1>(Stream<1>(" + K + ", true, " + std::to_string(fold) + "u))"
                                : "std_normal(bits" + Q + ")";
          o << ind << "const float t" << Q << " = a1_" << Q << " * " << e << ";\n";
          o << ind << "const float vf" << Q << " = a0_" << Q << " + t" << Q << ";\n";
          break;
        }
        case GJX_DIST_BERNOULLI:
          o << ind << "const int32_t vi" << Q << " = uniform01(bits" << Q << ") < a0_" << Q << " ? 1 : 0;\n";
          break;
        case GJX_DIST_GAMMA:
          o << ind << "const Stream<" << I << "> strm" << Q << "(" << K << ", true, " << fold << "u);\n";
          o << ind << "const float vf" << Q << " = std_gamma<" << I << ">(strm" << Q << ", 0, a0_" << Q << ") / a1_" << Q << ";\n";
          break;
        case GJX_DIST_BETA:
          o << ind << "const Stream<" << I << "> strm" << Q << "(" << K << ", true, " << fold << "u);\n";
          o << ind << "const float g1_" << Q << " = std_gamma<" << I << ">(strm" << Q << ", 0, a0_" << Q << ");\n";
          o << ind << "const float g2_" << Q << " = std_gamma<" << I << ">(strm" << Q << ", 1, a1_" << Q << ");\n";
          o << ind << "const float vf" << Q << " = g1_" << Q << " / (g1_" << Q << " + g2_" << Q << ");\n";
          break;
        default:
          if (st.cat_mode == 0) {
            o << ind << "const Stream<" << I << "> strm" << Q << "(" << K << ", true, " << fold << "u);\n";
            o << ind << "const int32_t vi" << Q << " = jcat_gumbel<" << I << ">(" << row << ", " << st.n_cat << "u, strm" << Q << ");\n";
          } else if (st.cat_ent) {  // the row's prepared {CDF, log-probability} entries, entered at the guide of the draw's
                                    // top byte (same integers as the two-pass walk: the same category); the entry the walk
                                    // ends on carries the log-density
            o << ind << "uint32_t lpb" << Q << ";\n";
            o << ind << "const int32_t vi" << Q << " = jcat_invcdf_gb(" << plit_as("uint4", st.cat_guide4) << " + ((size_t)rr" << Q << " << " << st.cat_gbits << "), "
              << plit_as("uint2", st.cat_ent) << " + (size_t)rr" << Q << " * " << st.n_cat << ", " << st.n_cat << "u, bits" << Q << ", " << (32 - st.cat_gbits) << ", lpb" << Q << ");\n";
          } else {
            o << ind << "const int32_t vi" << Q << " = jcat_invcdf(" << row << ", " << st.n_cat << "u, bits" << Q << ");\n";
          }
      }
    }
    std::string lp;
    const std::string v = (isint ? "vi" : "vf") + Q;
    // hoisted per-site constants: literals (pre == 1) or derived from the launch's parameters (pre == 2)
    const std::string p0 = st.pre == 2 ? "prm.d[" + std::to_string(2 * q) + "]" : flit(st.pre0);
    const std::string p1 = st.pre == 2 ? "prm.d[" + std::to_string(2 * q + 1) + "]" : flit(st.pre1);
    switch (st.dist) {
      case GJX_DIST_NORMAL:
        lp = st.pre ? "logpdf_normal_pre(" + v + ", a0_" + Q + ", " + p0 + ", " + p1 + ")"
                    : "logpdf_normal(" + v + ", a0_" + Q + ", a1_" + Q + ")";
        break;
      case GJX_DIST_GAMMA:
        lp = st.pre ? "logpdf_gamma_pre(" + v + ", a0_" + Q + ", a1_" + Q + ", " + p1 + ")"
                    : "logpdf_gamma(" + v + ", a0_" + Q + ", a1_" + Q + ")";
        break;
      case GJX_DIST_BETA:
        lp = st.pre ? "logpdf_beta_pre(" + v + ", a0_" + Q + ", a1_" + Q + ", " + p1 + ")"
                    : "logpdf_beta(" + v + ", a0_" + Q + ", a1_" + Q + ")";
        break;
      case GJX_DIST_BERNOULLI: lp = "logpdf_bernoulli(" + v + " != 0, a0_" + Q + ")"; break;
      default:
        if (st.cat_ent && !st.observed && st.cat_mode != 0)
          lp = "u2f(lpb" + Q + ")";  // a drawn category is in range
        else
          lp = "((" + v + " < 0 || " + v + " >= " + std::to_string(st.n_cat) + ") ? -__builtin_inff() : " +
               (st.cat_logp_t ? plit(st.cat_logp_t) + "[(size_t)" + v + " * " + std::to_string(st.n_rows) + " + rr" + Q + "]"
                : st.cat_ent ? "u2f(" + plit_as("uint2", st.cat_ent) + "[(size_t)rr" + Q + " * " + std::to_string(st.n_cat) + " + " + v + "].y)"
                             : row + "[" + v + "] - jrow_lse(" + row + ", " + std::to_string(st.n_cat) + "u)") + ")";
    }
    // an observed site whose arguments and value are ALL compile-time constants has a compile-time log-density: opaque, so
    // that no overflow of valid constants (-inf, +inf) becomes a constant log-weight either
    auto is_const = [](const CArgT& a) {  // a literal, or a program over literals only
      if (a.kind == GJX_ARG_CONST) return true;
      if (a.kind != GJX_ARG_EXPR) return false;
      const gjx_expr_op* ops = reinterpret_cast<const gjx_expr_op*>(a.table);
      for (int k = 0; k < a.ref; ++k)
        if (ops[k].op >= GJX_EXPR_SITE && ops[k].op <= GJX_EXPR_OBS) return false;
      return true;
    };
    const bool all_const = st.observed && st.obs.kind == GJX_ARG_CONST && is_const(st.a0) &&
                           (st.dist == GJX_DIST_BERNOULLI || st.dist == GJX_DIST_CATEGORICAL || is_const(st.a1));
    if (all_const) lp = "opq(" + lp + ")";
    const std::string aw = acc("w", scope_of(q)), as = acc("sc", scope_of(q));
    o << ind << "{ const float lp = " << lp << "; " << as << " = " << as << " + lp;"
      << (st.observed ? " " + aw + " = " + aw + " + lp;" : "") << " }\n";
    if ((mode == 0 || mode == 2) && st.out_col >= 0 && store_values)
      o << ind << "reinterpret_cast<uint32_t*>(cols.out[" << st.out_col << "])[" << (mode == 2 ? "oi" : "i") << sfx << "] = "
        << (isint ? "(uint32_t)" + v : "f2u(" + v + ")") << ";\n";
  }

  // a callee's weight and score are ITS totals, added to the caller's when the call returns (static.py:374-380: the
  // caller adds `w`; StaticTrace.get_score sums the sub-traces' scores): one accumulator pair per scope
  std::string acc(const char* base, int k) const { return k == 0 ? std::string(base) + sfx : std::string(base) + "_s" + std::to_string(k) + sfx; }
  void close_scopes(int pos) {  // the calls that have returned once the sites before `pos` are done (inner ones first)
    if (!sc) return;
    for (int k = sc->n_scopes; k >= 1; --k)
      if (sc->end[k] == pos && sc->begin[k] < pos)
        o << ind << acc("w", sc->parent[k]) << " = " << acc("w", sc->parent[k]) << " + " << acc("w", k) << "; "
          << acc("sc", sc->parent[k]) << " = " << acc("sc", sc->parent[k]) << " + " << acc("sc", k) << ";\n";
  }
  void run() {
    emit_scope_keys();
    for (int q = 0; q < n_sites; ++q) {
      head(q);
      tail(q);
      close_scopes(q + 1);
    }
  }
};

// bm_lds: the kernels of this source stage the Box-Muller tables in LDS (each calls bm_stage() at entry; gjx_device.hpp)
inline void emit_prelude(std::ostringstream& o, bool fast_math = false, bool bm_lds = false) {
  if (bm_lds) o << "#define GJX_BM_LDS 1\n";
  if (fast_math) o << "#define GJX_FAST_MATH 1\n";  // gjx.h GJX_PLAN_FAST_MATH: hardware transcendentals (gjx_device.hpp d_log / d_exp / bm_pair)
  o << "#include \"gjx_device.hpp\"\nusing namespace gjx;\n";
  o << "__device__ __forceinline__ float jrow_max(const float* l, uint32_t K){ float m=l[0]; for(uint32_t c=1;c<K;++c) m = l[c]>m?l[c]:m; return m; }\n";
  o << "__device__ __forceinline__ float jrow_lse(const float* l, uint32_t K){ const float m=jrow_max(l,K); float acc=0.0f; for(uint32_t c=0;c<K;++c) acc = acc + m_exp(l[c]-m); return m + m_log(acc); }\n";
  o << "__device__ __forceinline__ int32_t jcat_invcdf(const float* l, uint32_t K, uint32_t bits){ const float m=jrow_max(l,K); uint64_t Q=0; for(uint32_t c=0;c<K;++c) Q += cat_fix(l[c],m); const uint64_t thr=((uint64_t)bits*Q)>>32; uint64_t C=0; for(uint32_t c=0;c<K;++c){ C += cat_fix(l[c],m); if (C>thr) return (int32_t)c; } return (int32_t)(K-1); }\n";
  // the guide bucket: one scattered 16-byte load per draw from a table of up to 2 MB (L2-resident).  As a BUFFER load
  // (resource anchored 1 GiB below the first lane's address; every lane of the wave reads the same table, so all offsets
  // are in range) the scan of the 256-state HMM runs 4.02 -> 2.39 ms per 5e8 particle-steps against the same load as
  // global_load_dwordx4; the sc0 / sc1 bits make no further difference, a non-temporal global load is 2x slower and an
  // agent-scope pair of 8-byte loads is even (profiles/r04_ab/README.md).  GJX_GUIDE_LOAD=global keeps the old form for A/B.
  {
    const char* gl = std::getenv("GJX_GUIDE_LOAD");
    if (gl && !strcmp(gl, "global"))
      o << "__device__ __forceinline__ uint4 jguide_load(const uint4* p){ return *p; }\n";
    else
      o << "typedef unsigned jv4u_t __attribute__((ext_vector_type(4)));\n__device__ __forceinline__ uint4 jguide_load(const uint4* p){ const uint64_t a=(uint64_t)(uintptr_t)p; const uint64_t first=((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a>>32))<<32)|(uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a); const uint64_t base=first-(1ull<<30); __amdgpu_buffer_rsrc_t rs=__builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)base,0,0x7fffffff,0x00020000); const jv4u_t v=__builtin_amdgcn_raw_buffer_load_b128(rs,(int)(uint32_t)(a-base),0,0); return make_uint4(v[0],v[1],v[2],v[3]); }\n";
  }
  o << "__device__ __forceinline__ int32_t jcat_invcdf_gb(const uint4* guide, const uint2* ent, uint32_t K, uint32_t bits, int sh, uint32_t& lpb){ const uint4 g=jguide_load(guide+(bits>>sh)); const uint32_t c0=g.y&511u, c1=(g.y>>9)&511u; if (bits<=g.x){ lpb=g.z; return (int32_t)c0; } if (!(g.y>>18)){ lpb=g.w; return (int32_t)c1; } const uint64_t thr=((uint64_t)bits*(uint64_t)ent[K-1].x)>>32; uint32_t c=c1; uint2 e=ent[c]; while (c<K-1 && (uint64_t)e.x<=thr){ ++c; e=ent[c]; } lpb=e.y; return (int32_t)c; }\n";
  o << "template <int IMPL> __device__ __forceinline__ int32_t jcat_gumbel(const float* l, uint32_t K, const Stream<IMPL>& st){ int32_t best=0; float bv=-__builtin_inff(); for(uint32_t c=0;c<K;++c){ const float v = l[c] + gumbel_from_bits(st.bits32(c)); if (v>bv || c==0){ bv=v; best=(int32_t)c; } } return best; }\n";
}

// Write-through (sc1) column stores in the quad kernels when a launch holds ONE pass (store16_out, gjx_device.hpp): the
// kernel's end no longer waits for the write-back of its 48 MB — 4.94e10 -> 5.47e10 particles/s at one pass per launch;
// with many passes per launch the write-back overlaps the following passes and plain stores are 1 % faster, hence the
// run-time flag.  GJX_JIT_WT=0: plain stores always.
inline bool wt_stores_knob() {
  const char* e = std::getenv("GJX_JIT_WT");
  return !(e && e[0] == '0');
}
// The sites of a lane that owns NP whole PAIRS of adjacent particles (suffixes A, B [, C, D]) under PHILOX: one cipher
// block per pair and two draws (pk0 / pk1: the cipher key; pair0 [, pair1]: the pairs' counter words — defined by the
// caller), one Box-Muller transform per pair and Normal site, every stored column one vector store at `store_at`.
// GJX_JIT_GAMMA_MULTI=0: every particle of a lane loops over its own rejection attempts (the r01-r03 form; A/B knob)
inline bool gamma_multi_knob() {
  const char* e = std::getenv("GJX_JIT_GAMMA_MULTI");
  return !(e && e[0] == '0');
}
template <class CSiteT, class CArgT>
inline void emit_pair_lane_sites(std::ostringstream& o, std::vector<SiteEmitter<CSiteT, CArgT>>& em, const CSiteT* sites, int n_sites,
                                 int NP, const std::string& ind, const std::string& store_at) {
  const char* sfx[4] = {"A", "B", "C", "D"};
  const int P = 2 * NP;
int cur_pair_blk = -1;
  for (int q = 0; q < n_sites; ++q) {
    const CSiteT& st = sites[q];
    for (int u = 0; u < P; ++u) em[u].head(q);
    const std::string Q = std::to_string(q);
    if (!st.observed && em[0].one_word(st) && em[0].scope_of(q) == 0) {  // (a callee's sites: every particle's own lone key)
      // draw f of a pair: block f >> 1 holds words (even, odd particle) of draw 2 (f >> 1) and of draw 2 (f >> 1) + 1
      const uint32_t f = em[0].fold_of(q);
      const int blk = (int)(f >> 1);
      const std::string B = std::to_string(blk);
      if (blk != cur_pair_blk) {
        cur_pair_blk = blk;
        for (int pi = 0; pi < NP; ++pi) {
          const std::string V = "pp" + std::to_string(pi) + "_" + B;
          o << ind << "uint32_t " << V << "_0, " << V << "_1, " << V << "_2, " << V << "_3;\n";
          o << ind << "philox4x32(pk0, pk1, (uint32_t)pair" << pi << ", (uint32_t)(pair" << pi << " >> 32), " << blk << "u, kTagPair, " << V
            << "_0, " << V << "_1, " << V << "_2, " << V << "_3);\n";
        }
      }
      for (int pi = 0; pi < NP; ++pi) {
        const std::string V = "pp" + std::to_string(pi) + "_" + B;
        o << ind << "const uint32_t bits" << Q << sfx[2 * pi] << " = " << V << "_" << ((f & 1u) << 1) << ", bits" << Q << sfx[2 * pi + 1]
          << " = " << V << "_" << (((f & 1u) << 1) | 1u) << ";\n";
      }
    }
    if (!st.observed && st.dist == GJX_DIST_NORMAL && em[0].scope_of(q) == 0) {
      for (int pi = 0; pi < NP; ++pi) {
        const std::string Z = Q + "_" + std::to_string(pi);
        o << ind << "float zc" << Z << ", zs" << Z << ";\n      bm_pair(bits" << Q << sfx[2 * pi] << ", bits" << Q << sfx[2 * pi + 1] << ", zc" << Z
          << ", zs" << Z << ");\n";
      }
      for (int pi = 0; pi < NP; ++pi) {
        const std::string Z = Q + "_" + std::to_string(pi);
        em[2 * pi].tail(q, "zc" + Z);
        em[2 * pi + 1].tail(q, "zs" + Z);
      }
    } else if (!st.observed && (st.dist == GJX_DIST_GAMMA || st.dist == GJX_DIST_BETA) && gamma_multi_knob()) {
      // the lane's P particles draw their gammas TOGETHER (gjx_device.hpp std_gamma_multi: the first attempts straight-line,
      // the retries in one shared loop): the same values as P std_gamma calls, fewer divergent wave-trips
      const std::string I = std::to_string(em[0].impl), PS = std::to_string(P);
      const uint32_t fold = em[0].fold_of(q);
      o << ind << "float vg0_" << Q << "[" << PS << "]" << (st.dist == GJX_DIST_BETA ? ", vg1_" + Q + "[" + PS + "]" : std::string()) << ";\n";
      o << ind << "{\n" << ind << "  const Stream<" << I << "> gs[" << PS << "] = {";
      for (int u = 0; u < P; ++u) o << (u ? ", " : "") << "Stream<" << I << ">(" << em[u].key_of(q) << ", true, " << fold << "u)";
      o << "};\n";
      o << ind << "  const float gc0[" << PS << "] = {";
      for (int u = 0; u < P; ++u) o << (u ? ", " : "") << "a0_" << Q << sfx[u];
      o << "};\n" << ind << "  std_gamma_multi<" << I << ", " << PS << ">(gs, 0, gc0, vg0_" << Q << ");\n";
      if (st.dist == GJX_DIST_BETA) {
        o << ind << "  const float gc1[" << PS << "] = {";
        for (int u = 0; u < P; ++u) o << (u ? ", " : "") << "a1_" << Q << sfx[u];
        o << "};\n" << ind << "  std_gamma_multi<" << I << ", " << PS << ">(gs, 1, gc1, vg1_" << Q << ");\n";
      }
      o << ind << "}\n";
      for (int u = 0; u < P; ++u) {
        if (st.dist == GJX_DIST_GAMMA)
          o << ind << "const float vf" << Q << sfx[u] << " = vg0_" << Q << "[" << u << "] / a1_" << Q << sfx[u] << ";\n";
        else
          o << ind << "const float vf" << Q << sfx[u] << " = vg0_" << Q << "[" << u << "] / (vg0_" << Q << "[" << u << "] + vg1_" << Q << "[" << u << "]);\n";
        em[u].presampled = true;
        em[u].tail(q);
        em[u].presampled = false;
      }
    } else {
      for (int u = 0; u < P; ++u) em[u].tail(q);
    }
    for (int u = 0; u < P; ++u) em[u].close_scopes(q + 1);
    if (st.out_col >= 0) {  // the lane's particles are adjacent in the column: one 8- / 16-byte store per lane
      const bool isint = SiteEmitter<CSiteT, CArgT>::is_int(st);
      std::string vals;
      for (int u = 0; u < P; ++u)
        vals += (u ? ", " : "") + (isint ? "(uint32_t)vi" + Q + sfx[u] : "f2u(vf" + Q + sfx[u] + ")");
      o << "#ifndef GJX_EXP_NO_VALUE_STORES\n";
      if (P == 4 && wt_stores_knob())
        o << ind << "store16_out(reinterpret_cast<uint32_t*>(cols.out[" << st.out_col << "]) + " << store_at << ", make_uint4(" << vals << "), wt_one_pass);\n";
      else
      o << ind << "*reinterpret_cast<uint" << P << "*>(reinterpret_cast<uint32_t*>(cols.out[" << st.out_col << "]) + " << store_at << ") = make_uint" << P
        << "(" << vals << ");\n";
      o << "#endif\n";
    }
  }
}

// The importance kernel of a plan.  Two forms:
//  * generic: one 256-particle row per 256-thread workgroup, one particle per lane, any key batch;
//  * paired (PHILOX, lazy children of a lane-0 key, even first index): one row per 128-thread workgroup,
//    TWO ADJACENT PARTICLES per lane.  The lane derives both particles' words anyway, so every Normal site
//    costs one shared Box-Muller transform for the pair, and the cipher key (the parent's) is uniform over
//    the launch: the round keys live in scalar registers.
template <class CSiteT, class CArgT>
struct Gen {
  std::ostringstream o;
  int impl;
  const CSiteT* sites;
  int n_sites;
  int min_waves = 0;  // __launch_bounds__ waves-per-SIMD hint (0 = none)
  int block = 256;    // threads per workgroup of the generated kernel
  int rows_per_block = 1;
  bool laned = false; // the paired form (see above)
  bool fast_math = false;  // GJX_PLAN_FAST_MATH
  const ScopeInfo* sc = nullptr;  // nested calls (null: a flat body)
  int pairs_per_lane = 1;  // paired form: 1 = two adjacent particles per lane (128-thread workgroup per 256-particle row);
                           // 2 = FOUR adjacent particles per lane: one WAVE owns the whole row, so the row statistics are
                           // wave reductions (DPP only: no LDS, no barrier) and every column store is 16 bytes per lane

  static const char* signature() {
    return "(KeySrc ks, RunCols cols, float* score, float* logw, uint64_t n, float* max_partials, int32_t* row_e, "
           "uint64_t* row_s, LseTail tail, PassBatch bt, PlanParams prm, PlanTables tabs) {\n";
  }
  const char* kname() const { return impl == 0 ? "gjx_plan_kernel_threefry" : "gjx_plan_kernel_philox"; }
  // PHILOX Normal sites draw through the Box-Muller tables: this source's kernels stage them in LDS (gjx_device.hpp bm_stage)
  bool bm_lds() const {
    if (impl != 1 || fast_math) return false;
    for (int q = 0; q < n_sites; ++q)
      if (!sites[q].observed && sites[q].dist == GJX_DIST_NORMAL) return true;
    return false;
  }

  std::string run_paired() {
    // NP pairs of adjacent particles per lane; a 256-particle row is 128 / NP lanes.  R rows per workgroup.
    const int NP = pairs_per_lane == 2 ? 2 : 1;
    const int lanes_per_row = 128 / NP;
    // rows per workgroup: 1 when the walk stores value columns (measured r02); a walk without any (an estimate of the
    // log-marginal alone: row sums out, nothing else) runs 4 one-wave rows per workgroup — 256 lanes fold the row sums in
    // the fused tail in one trip of loads instead of 8 by a single wave (host-API call 52.5 -> 47.6 us at 1e6 particles)
    bool any_col = false;
    for (int q = 0; q < n_sites; ++q) any_col = any_col || sites[q].out_col >= 0;
    int R = (NP == 2 && !any_col) ? 4 : 1;
    if (const char* e = std::getenv("GJX_JIT_PAIR_ROWS")) R = atoi(e) == 2 ? 2 : (atoi(e) == 4 ? 4 : 1);
    block = lanes_per_row * R;
    rows_per_block = R;
    const int waves_per_row = lanes_per_row / 64;  // 2 (pairs) or 1 (quads)
    emit_prelude(o, fast_math, bm_lds());
    o << "extern \"C\" __global__ __launch_bounds__(" << block << (min_waves > 0 ? ", " + std::to_string(min_waves) : std::string())
      << ") void " << kname() << signature();
    if (waves_per_row > 1) o << "  __shared__ float sh_red[" << waves_per_row * R << "];\n  __shared__ uint64_t sh_sum[" << waves_per_row * R << "];\n";
    if (R == 1) o << "  const int wv = threadIdx.x >> 6, pr = 0, tr = threadIdx.x;  // (block-uniform row: the cipher key stays scalar)\n";
    else o << "  const int wv = threadIdx.x >> 6, pr = threadIdx.x / " << lanes_per_row << ", tr = threadIdx.x % " << lanes_per_row << ";\n";
    o << "  (void)wv;\n";
    // the kernel's scalar arguments in ONE round of loads (gjx_device.hpp: resample_args_anchor has the measurement)
    o << "  asm volatile(\"\" :: \"s\"(n), \"s\"(score), \"s\"(logw), \"s\"(max_partials), \"s\"(row_e), \"s\"(row_s), \"s\"(bt.n_pass), \"s\"(bt.rows_per_pass), \"s\"(bt.pass_stride), \"s\"(bt.row_stride), \"s\"(ks.first), \"s\"(ks.parent.k0), \"s\"(ks.parent.k1), \"s\"(tail.tickets));\n";
    {
      bool seen[64] = {};  // ... and the value columns this plan stores (RunCols::out has 64)
      for (int q = 0; q < n_sites; ++q)
        if (sites[q].out_col >= 0 && sites[q].out_col < 64 && !seen[sites[q].out_col]) {
          seen[sites[q].out_col] = true;
          o << "  asm volatile(\"\" :: \"s\"(cols.out[" << sites[q].out_col << "]));\n";
        }
    }
    if (bm_lds()) o << "  bm_stage();\n";
    o << "  const uint64_t rows_all = (uint64_t)bt.n_pass * bt.rows_per_pass;\n";
    o << "  const bool wt_one_pass = bt.n_pass <= 1u; (void)wt_one_pass;\n";
    o << "  for (uint64_t g0 = (uint64_t)blockIdx.x * " << R << "; g0 < rows_all; g0 += (uint64_t)gridDim.x * " << R << ") {\n";
    o << "    const uint64_t gr = g0 + pr;\n";
    o << "    const uint32_t pass = (uint32_t)(gr / bt.rows_per_pass);\n";
    o << "    const uint64_t row = gr - (uint64_t)pass * bt.rows_per_pass;\n";
    o << "    const bool live_row = gr < rows_all;\n";
    o << "    const uint32_t pk0 = bt.n_pass > 1 ? bt.parent[pass < bt.n_pass ? pass : 0][0] : ks.parent.k0;\n";
    o << "    const uint32_t pk1 = bt.n_pass > 1 ? bt.parent[pass < bt.n_pass ? pass : 0][1] : ks.parent.k1;\n";
    o << "    const uint64_t po = (uint64_t)pass * bt.pass_stride, ro = (uint64_t)pass * bt.row_stride;  // this pass's outputs\n";
    const char* sfx[4] = {"A", "B", "C", "D"};
    const int P = 2 * NP;  // particles per lane
    o << "    const uint64_t iA = row * 256 + " << P << " * (uint64_t)tr;\n";
    for (int u = 1; u < P; ++u) o << "    const uint64_t i" << sfx[u] << " = iA + " << u << ";\n";
    // n is a multiple of P in this form (checked by the host): all particles of a lane exist or none
    o << "    const bool ok = live_row && iA < n;\n";
    for (int u = 0; u < P; ++u) o << "    const uint64_t li" << sfx[u] << " = i" << sfx[u] << "; (void)li" << sfx[u] << ";\n";
    for (int u = 0; u < P; ++u) o << "    float w" << sfx[u] << " = 0.0f, sc" << sfx[u] << " = 0.0f;\n";
    o << "    if (ok) {\n";
    o << "      const uint64_t lnA = ks.first + iA + 1u;\n";
    for (int u = 0; u < P; ++u) {
      if (u) o << "      const uint64_t ln" << sfx[u] << " = lnA + " << u << "u;\n";
      o << "      const Key pkey" << sfx[u] << "{pk0, pk1, (uint32_t)ln" << sfx[u] << ", (uint32_t)(ln" << sfx[u] << " >> 32)}; (void)pkey" << sfx[u] << ";\n";
    }
    std::vector<SiteEmitter<CSiteT, CArgT>> em;
    for (int u = 0; u < P; ++u) {
      em.push_back(SiteEmitter<CSiteT, CArgT>{o, impl, 0, sites, n_sites, "      ", sfx[u]});
      em.back().store_values = false;
      em.back().ext_bits = true;  // the lane owns whole pairs: their single-word draws come from the pairs' blocks
      em.back().sc = sc;
    }
    for (int u = 0; u < P; ++u) em[u].emit_scope_keys();
    o << "      const uint64_t pair0 = (lnA - 1u) >> 1;\n";
    if (NP == 2) o << "      const uint64_t pair1 = pair0 + 1u;\n";
    emit_pair_lane_sites<CSiteT, CArgT>(o, em, sites, n_sites, NP, "      ", "po + iA");
    {
      std::string ws, ss;
      for (int u = 0; u < P; ++u) { ws += (u ? ", w" : "w") + std::string(sfx[u]); ss += (u ? ", sc" : "sc") + std::string(sfx[u]); }
      if (P == 4 && wt_stores_knob()) {
        std::string wb, sb;
        for (int u = 0; u < P; ++u) { wb += (u ? ", f2u(w" : "f2u(w") + std::string(sfx[u]) + ")"; sb += (u ? ", f2u(sc" : "f2u(sc") + std::string(sfx[u]) + ")"; }
        o << "      if (logw) store16_out(logw + po + iA, make_uint4(" << wb << "), wt_one_pass);\n";
        o << "      if (score) store16_out(score + po + iA, make_uint4(" << sb << "), wt_one_pass);\n";
      } else {
      o << "      if (logw) *reinterpret_cast<float" << P << "*>(logw + po + iA) = make_float" << P << "(" << ws << ");\n";
      o << "      if (score) *reinterpret_cast<float" << P << "*>(score + po + iA) = make_float" << P << "(" << ss << ");\n";
      }
    }
    o << "    }\n";
    o << "    if (max_partials || row_e) {\n";
    o << "      const float ninf = -__builtin_inff();\n";
    {
      std::string mx = "wA";
      for (int u = 1; u < P; ++u) mx = "(" + mx + " > w" + sfx[u] + " ? " + mx + " : w" + sfx[u] + ")";
      o << "      float bm = wave_max(ok ? " << mx << " : ninf);\n";
    }
    if (waves_per_row > 1) {
      o << "      __syncthreads();\n      if ((threadIdx.x & 63) == 0) sh_red[wv] = bm;\n      __syncthreads();\n";
      o << "      bm = sh_red[2 * pr] > sh_red[2 * pr + 1] ? sh_red[2 * pr] : sh_red[2 * pr + 1];\n";
    }
    o << "      if (max_partials && tr == 0 && live_row) max_partials[ro + row] = bm;\n";
    o << "      if (row_e) {\n";
    o << "        const int32_t eb = row_anchor(bm);\n";
    {
      std::string sm;
      for (int u = 0; u < P; ++u) sm += (u ? " + rowfix(w" : "rowfix(w") + std::string(sfx[u]) + ", eb)";
      o << "        uint64_t sb = wave_sum_52(ok ? (" << sm << ") : 0);  // (at most four weights below 2^32)\n";
    }
    if (waves_per_row > 1) {
      o << "        __syncthreads();\n        if ((threadIdx.x & 63) == 0) sh_sum[wv] = sb;\n        __syncthreads();\n";
      o << "        sb = sh_sum[2 * pr] + sh_sum[2 * pr + 1];\n";
    }
    o << "        if (tr == 0 && live_row) lse_store_row(row_e, row_s, ro + row, eb, sb, tail.tickets != nullptr);\n";
    o << "      }\n    }\n";
    o << "  }\n  if (row_e) lse_tail(row_e, row_s, (n + 255) / 256, tail);\n}\n";
    return o.str();
  }

  std::string run() {
    if (laned && impl == 1) return run_paired();
    const std::string I = std::to_string(impl);
    emit_prelude(o, fast_math, bm_lds());
    // One workgroup per 256-particle row (grid-stride): short blocks keep every SIMD's wave slots
    // full even at 1e6 particles (15 rows per lane), where a 4-row block would serialise its rows.
    o << "extern \"C\" __global__ __launch_bounds__(256" << (min_waves > 0 ? ", " + std::to_string(min_waves) : std::string())
      << ") void " << kname() << signature();
    o << "  __shared__ float sh_red[4];\n  __shared__ uint64_t sh_sum[4];\n";
    if (bm_lds()) o << "  bm_stage();\n";
    o << "  const uint64_t rows_all = (uint64_t)bt.n_pass * bt.rows_per_pass;\n";
    o << "  for (uint64_t gr = blockIdx.x; gr < rows_all; gr += gridDim.x) {\n";
    o << "    const uint32_t pass = (uint32_t)(gr / bt.rows_per_pass);\n";
    o << "    const uint64_t row = gr - (uint64_t)pass * bt.rows_per_pass;\n";
    o << "    const uint64_t po = (uint64_t)pass * bt.pass_stride, ro = (uint64_t)pass * bt.row_stride;\n";
    o << "    KeySrc kp = ks;\n";
    o << "    if (bt.n_pass > 1) { kp.parent.k0 = bt.parent[pass][0]; kp.parent.k1 = bt.parent[pass][1]; }\n";
    o << "    float tmax = -__builtin_inff();\n    bool live = false;\n";
    o << "    {\n";
    o << "      const uint64_t li = row * 256 + threadIdx.x, i = po + li;\n";
    o << "      const bool ok = li < n;\n";
    o << "      if (ok) {\n";
    o << "        const Key pkey = key_at<" << I << ">(kp, li);\n";
    o << "        float w = 0.0f, sc = 0.0f;\n";
    SiteEmitter<CSiteT, CArgT> em{o, impl, 0, sites, n_sites, "        "};
    em.sc = sc;
    em.run();
    o << "        if (logw) logw[i] = w;\n        if (score) score[i] = sc;\n        tmax = w;\n        live = true;\n";
    o << "      }\n    }\n";
    o << "    if (max_partials || row_e) {\n";
    o << "      const float bm = block_max(tmax, sh_red);\n";
    o << "      if (max_partials && threadIdx.x == 0) max_partials[ro + row] = bm;\n";
    o << "      if (row_e) {\n";
    o << "        const int32_t eb = row_anchor(bm);\n";
    o << "        const uint64_t sb = block_sum(live ? rowfix(tmax, eb) : 0, sh_sum);\n";
    o << "        if (threadIdx.x == 0) lse_store_row(row_e, row_s, ro + row, eb, sb, tail.tickets != nullptr);\n";
    o << "      }\n    }\n";
    o << "  }\n  if (row_e) lse_tail(row_e, row_s, (n + 255) / 256, tail);\n}\n";
    return o.str();
  }
};

// Importance over a Scan model (gjx_scan_run): one lane per particle walks all T steps — the key chain, the carry and
// the running weight / score stay in registers, every step stores its sampled values as one coalesced row of the
// time-major columns, and the epilogue is the importance kernel's (row maxima, row-anchored sums, in-launch fold).
template <class CSiteT, class CArgT>
struct GenScan {
  std::ostringstream o;
  int impl;
  const CSiteT* sites;
  int n_sites;
  const CArgT* next_state;
  int n_state, n_obs;
  bool fast_math = false;
  const ScopeInfo* sc = nullptr;  // nested calls inside the step kernel (null: a flat body)
  bool quad = false;  // PHILOX, the lazy children of a lane-0 key, n and the columns multiples of four: FOUR adjacent particles
                      // per lane (one wave per 256-particle row).  A step's keys are lanes (scan_step_key_philox), so the
                      // lane's two pairs share their cipher blocks and Box-Muller transforms exactly as in the importance
                      // kernel's quad form; row statistics are wave reductions, every store is 16 bytes per lane.
  int block = 256;
  const char* kname() const { return impl == 0 ? "gjx_scan_kernel_threefry" : "gjx_scan_kernel_philox"; }
  // PHILOX Normal sites draw through the Box-Muller tables: this source's kernels stage them in LDS (gjx_device.hpp bm_stage)
  bool bm_lds() const {
    if (impl != 1 || fast_math) return false;
    for (int q = 0; q < n_sites; ++q)
      if (!sites[q].observed && sites[q].dist == GJX_DIST_NORMAL) return true;
    return false;
  }
  std::string run_quad() {
    block = 64;
    const char* sfx[4] = {"A", "B", "C", "D"};
    const int P = 4;
    emit_prelude(o, fast_math, bm_lds());
    o << "struct StepObs { const float* obs; };\n";
    o << "extern \"C\" __global__ __launch_bounds__(64) void " << kname()
      << "(KeySrc ks, RunCols cols, ScanArgs sa, float* score, float* logw, float* max_partials, int32_t* row_e, uint64_t* row_s, LseTail tail, PlanTables tabs) {\n";
    if (bm_lds()) o << "  bm_stage();\n";
    o << "  const uint64_t n = sa.n, rows_all = (n + 255) / 256;\n";
    o << "  const bool wt_one_pass = false; (void)wt_one_pass;  // (a scan's stores spread over its T steps)\n";
    o << "  const uint32_t pk0 = ks.parent.k0, pk1 = ks.parent.k1;\n";
    o << "  for (uint64_t row = blockIdx.x; row < rows_all; row += gridDim.x) {\n";
    o << "    const uint64_t iA = row * 256 + 4 * (uint64_t)threadIdx.x;\n";
    o << "    const bool ok = iA < n;  // (n is a multiple of four in this form: all particles of a lane exist or none)\n";
    for (int u = 0; u < P; ++u) o << "    float wt" << sfx[u] << " = 0.0f, sct" << sfx[u] << " = 0.0f;\n";
    o << "    if (ok) {\n";
    o << "      const uint64_t lnA = ks.first + iA + 1u;\n";
    for (int u = 0; u < P; ++u)
      for (int k = 0; k < n_state; ++k)
        o << "      float st_" << k << sfx[u] << " = sa.carry0_cols[" << k << "] ? sa.carry0_cols[" << k << "][iA + " << u << "] : sa.carry0[" << k << "];\n";
    o << "      for (int32_t t = 0; t < sa.n_steps; ++t) {\n";
    o << "        const uint64_t lt = lnA + (((uint64_t)(uint32_t)t + 1u) << 40);  // the lanes of step t (scan_step_key_philox)\n";
    for (int u = 0; u < P; ++u)
      o << "        const Key pkey" << sfx[u] << "{pk0, pk1, (uint32_t)(lt + " << u << "u), (uint32_t)((lt + " << u << "u) >> 32)}; (void)pkey" << sfx[u] << ";\n";
    o << "        const uint64_t pair0 = (lt - 1u) >> 1, pair1 = pair0 + 1u;\n";
    o << "        StepObs a; a.obs = sa.obs + (size_t)t * " << n_obs << "; (void)a;\n";
    o << "        const uint64_t oiA = (uint64_t)t * sa.col_stride + iA; (void)oiA;\n";
    for (int u = 0; u < P; ++u) o << "        float w" << sfx[u] << " = 0.0f, sc" << sfx[u] << " = 0.0f;\n";
    std::vector<SiteEmitter<CSiteT, CArgT>> em;
    for (int u = 0; u < P; ++u) {
      em.push_back(SiteEmitter<CSiteT, CArgT>{o, impl, 2, sites, n_sites, "        ", sfx[u]});
      em.back().store_values = false;
      em.back().ext_bits = true;
      em.back().sc = sc;
    }
    for (int u = 0; u < P; ++u) em[u].emit_scope_keys();
    emit_pair_lane_sites<CSiteT, CArgT>(o, em, sites, n_sites, 2, "        ", "oiA");
    for (int u = 0; u < P; ++u)
      for (int k = 0; k < n_state; ++k) o << "        const float nx_" << k << sfx[u] << " = " << em[u].arg(next_state[k]) << ";\n";
    for (int u = 0; u < P; ++u)
      for (int k = 0; k < n_state; ++k) o << "        st_" << k << sfx[u] << " = nx_" << k << sfx[u] << ";\n";
    for (int u = 0; u < P; ++u) o << "        wt" << sfx[u] << " = wt" << sfx[u] << " + w" << sfx[u] << "; sct" << sfx[u] << " = sct" << sfx[u] << " + sc" << sfx[u] << ";\n";
    o << "      }\n";
    for (int k = 0; k < n_state; ++k)
      o << "      if (sa.carry_out[" << k << "]) *reinterpret_cast<float4*>(sa.carry_out[" << k << "] + iA) = make_float4(st_" << k << "A, st_" << k
        << "B, st_" << k << "C, st_" << k << "D);\n";
    o << "      if (logw) *reinterpret_cast<float4*>(logw + iA) = make_float4(wtA, wtB, wtC, wtD);\n";
    o << "      if (score) *reinterpret_cast<float4*>(score + iA) = make_float4(sctA, sctB, sctC, sctD);\n";
    o << "    }\n";
    o << "    if (max_partials || row_e) {\n";
    o << "      const float ninf = -__builtin_inff();\n";
    o << "      const float m4 = ((wtA > wtB ? wtA : wtB) > wtC ? (wtA > wtB ? wtA : wtB) : wtC);\n";
    o << "      const float bm = wave_max(ok ? (m4 > wtD ? m4 : wtD) : ninf);\n";
    o << "      if (max_partials && threadIdx.x == 0) max_partials[row] = bm;\n";
    o << "      if (row_e) {\n";
    o << "        const int32_t eb = row_anchor(bm);\n";
    o << "        const uint64_t sb = wave_sum(ok ? (rowfix(wtA, eb) + rowfix(wtB, eb) + rowfix(wtC, eb) + rowfix(wtD, eb)) : 0);\n";
    o << "        if (threadIdx.x == 0) lse_store_row(row_e, row_s, row, eb, sb, tail.tickets != nullptr);\n";
    o << "      }\n    }\n";
    o << "  }\n  if (row_e) lse_tail(row_e, row_s, (n + 255) / 256, tail);\n}\n";
    return o.str();
  }
  std::string run() {
    if (quad && impl == 1) return run_quad();
    const std::string I = std::to_string(impl);
    emit_prelude(o, fast_math, bm_lds());
    o << "struct StepObs { const float* obs; };\n";
    o << "extern \"C\" __global__ __launch_bounds__(256) void " << kname()
      << "(KeySrc ks, RunCols cols, ScanArgs sa, float* score, float* logw, float* max_partials, int32_t* row_e, uint64_t* row_s, LseTail tail, PlanTables tabs) {\n";
    o << "  __shared__ float sh_red[4];\n  __shared__ uint64_t sh_sum[4];\n";
    if (bm_lds()) o << "  bm_stage();\n";
    o << "  const uint64_t n = sa.n, rows_all = (n + 255) / 256;\n";
    o << "  for (uint64_t row = blockIdx.x; row < rows_all; row += gridDim.x) {\n";
    o << "    float tmax = -__builtin_inff();\n    bool live = false;\n";
    o << "    const uint64_t i = row * 256 + threadIdx.x;\n";
    o << "    if (i < n) {\n";
    o << "      Key " << (impl == 0 ? "pkey" : "pkey0") << " = key_at<" << I << ">(ks, i);\n";
    for (int k = 0; k < n_state; ++k)
      o << "      float st_" << k << " = sa.carry0_cols[" << k << "] ? sa.carry0_cols[" << k << "][i] : sa.carry0[" << k << "];\n";
    o << "      float wt = 0.0f, sct = 0.0f;\n";
    o << "      for (int32_t t = 0; t < sa.n_steps; ++t) {\n";
    if (impl == 0) o << "        pkey = fold_in<0>(pkey, (uint32_t)t);  // chained: the folded key is carried (scan.py:267-268)\n";
    else o << "        const Key pkey = scan_step_key_philox(pkey0, (uint32_t)t);  // a lane per (particle, step): no cipher block\n";
    o << "        StepObs a; a.obs = sa.obs + (size_t)t * " << n_obs << "; (void)a;\n";
    o << "        const uint64_t oi = (uint64_t)t * sa.col_stride + i; (void)oi;\n";
    o << "        float w = 0.0f, sc = 0.0f;\n";
    SiteEmitter<CSiteT, CArgT> em{o, impl, 2, sites, n_sites, "        "};
    em.sc = sc;
    em.run();
    for (int k = 0; k < n_state; ++k) o << "        const float nx_" << k << " = " << em.arg(next_state[k]) << ";\n";
    for (int k = 0; k < n_state; ++k) o << "        st_" << k << " = nx_" << k << ";\n";
    o << "        wt = wt + w;\n        sct = sct + sc;\n      }\n";
    for (int k = 0; k < n_state; ++k) o << "      if (sa.carry_out[" << k << "]) sa.carry_out[" << k << "][i] = st_" << k << ";\n";
    o << "      if (logw) logw[i] = wt;\n      if (score) score[i] = sct;\n      tmax = wt;\n      live = true;\n    }\n";
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
// per-slot kernel (init).  State columns are gathered from global memory by ancestor index like the fixed models.
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
  const ScopeInfo* sc_init = nullptr;  // nested calls inside init / step (null: flat bodies)
  const ScopeInfo* sc_step = nullptr;
  bool peers = false;  // r04: the step kernels of the peer transport (the source population lives in the peers' arenas)

  // PHILOX: the four consecutive slots jq .. jq+3 of a lane (jq a multiple of 4) walked together.  One-word draw
  // number f of the quad is ONE block, PH(ctr = (g_lo, g_hi, f, 'Q'), key = step key), g = jq / 4, slot u taking
  // word u (for f = 0 and one Normal site this is the fixed LGSSM filter, bit for bit); Normal sites are two
  // Box-Muller transforms over the quad's words; multi-word samplers keep their per-slot streams.
  // The draws of the quad that do not depend on the ancestors — the cipher block of every one-word draw of the body's own
  // sites and the Box-Muller transforms of its Normal sites — can run under the latency of the step's first loads: the
  // policy's prefetch() (as the hand-written LgssmPolicy has it) keeps them in members (registers).
  int quad_draws(const CSiteT* sites, int n_sites, const ScopeInfo* sc) const {
    int n = 0;
    SiteEmitter<CSiteT, CArgT> e{const_cast<std::ostringstream&>(o), impl, 1, sites, n_sites, "", ""};
    e.sc = sc;
    for (int q = 0; q < n_sites; ++q)
      if (!sites[q].observed && e.one_word(sites[q]) && e.scope_of(q) == 0) ++n;
    return n;
  }
  void emit_prefetch(const CSiteT* sites, int n_sites, const ScopeInfo* sc) {
    SiteEmitter<CSiteT, CArgT> e{o, impl, 1, sites, n_sites, "    ", ""};
    e.sc = sc;
    o << "  __device__ __forceinline__ void prefetch(int64_t jq) {\n    const uint64_t g = (uint64_t)jq >> 2;\n";
    int i = 0;
    for (int q = 0; q < n_sites; ++q) {
      const CSiteT& st = sites[q];
      if (st.observed || !e.one_word(st) || e.scope_of(q) != 0) continue;
      o << "    philox4x32(a.step_key.k0, a.step_key.k1, (uint32_t)g, (uint32_t)(g >> 32), " << e.fold_of(q) << "u, kTagQuad, pf_w[" << i
        << "][0], pf_w[" << i << "][1], pf_w[" << i << "][2], pf_w[" << i << "][3]);\n";
      if (st.dist == GJX_DIST_NORMAL) {
        o << "    bm_pair(pf_w[" << i << "][0], pf_w[" << i << "][1], pf_z[" << i << "][0], pf_z[" << i << "][1]);\n";
        o << "    bm_pair(pf_w[" << i << "][2], pf_w[" << i << "][3], pf_z[" << i << "][2], pf_z[" << i << "][3]);\n";
      }
      ++i;
    }
    o << "  }\n";
  }
  void emit_quad_body(const CSiteT* sites, int n_sites, const CArgT* state_args, bool step, bool prefetched = false) {
    const char* sf[4] = {"A", "B", "C", "D"};
    int pf_i = 0;
    std::vector<SiteEmitter<CSiteT, CArgT>> em;
    const ScopeInfo* sc = step ? sc_step : sc_init;
    for (int u = 0; u < 4; ++u) {
      em.push_back(SiteEmitter<CSiteT, CArgT>{o, impl, 1, sites, n_sites, "    ", sf[u]});
      em.back().ext_bits = true;
      em.back().sc = sc;
    }
    o << "    const uint64_t g = (uint64_t)jq >> 2;\n";
    for (int u = 0; u < 4; ++u) {
      if (step)
        for (int k = 0; k < n_state; ++k) o << "    const float st_" << k << sf[u] << " = src_load<kPeers>(a.prev_state[" << k << "], src[" << u << "], pd, tpr);\n";
      if (em[u].needs_stream_key() || sc) o << "    const Key pkey" << sf[u] << " = slot_key<1>(a.step_key, (uint64_t)jq + " << u << "u);\n";
      o << "    float w" << sf[u] << " = 0.0f, sc" << sf[u] << " = 0.0f;\n";
      em[u].emit_scope_keys();
    }
    for (int q = 0; q < n_sites; ++q) {
      const CSiteT& st = sites[q];
      const std::string Q = std::to_string(q);
      for (int u = 0; u < 4; ++u) em[u].head(q);
      const bool drawn = !st.observed && em[0].one_word(st) && em[0].scope_of(q) == 0;  // (a callee's sites: their own lone keys)
      const bool normal = drawn && st.dist == GJX_DIST_NORMAL;
      if (drawn && prefetched) {  // (computed by prefetch(): the same block, the same transforms)
        for (int u = 0; u < 4; ++u) o << "    const uint32_t bits" << Q << sf[u] << " = pf_w[" << pf_i << "][" << u << "]; (void)bits" << Q << sf[u] << ";\n";
        if (normal)
          for (int u = 0; u < 4; ++u) o << "    const float z" << Q << sf[u] << " = pf_z[" << pf_i << "][" << u << "];\n";
        ++pf_i;
      } else if (drawn) {
        o << "    uint32_t qw" << Q << "_0, qw" << Q << "_1, qw" << Q << "_2, qw" << Q << "_3;\n";
        o << "    philox4x32(a.step_key.k0, a.step_key.k1, (uint32_t)g, (uint32_t)(g >> 32), " << em[0].fold_of(q)
          << "u, kTagQuad, qw" << Q << "_0, qw" << Q << "_1, qw" << Q << "_2, qw" << Q << "_3);\n";
        for (int u = 0; u < 4; ++u) o << "    const uint32_t bits" << Q << sf[u] << " = qw" << Q << "_" << u << ";\n";
        if (normal) {
          o << "    float z" << Q << "A, z" << Q << "B, z" << Q << "C, z" << Q << "D;\n";
          o << "    bm_pair(bits" << Q << "A, bits" << Q << "B, z" << Q << "A, z" << Q << "B);\n";
          o << "    bm_pair(bits" << Q << "C, bits" << Q << "D, z" << Q << "C, z" << Q << "D);\n";
        }
      }
      for (int u = 0; u < 4; ++u) em[u].tail(q, normal ? "z" + Q + sf[u] : std::string());
      for (int u = 0; u < 4; ++u) em[u].close_scopes(q + 1);
    }
    for (int u = 0; u < 4; ++u) {
      for (int k = 0; k < n_state; ++k) o << "    out[" << u << "].s[" << k << "] = " << em[u].arg(state_args[k]) << ";\n";
      o << "    (void)sc" << sf[u] << ";\n    wq[" << u << "] = w" << sf[u] << ";\n";
    }
  }
  void emit_quad(const CSiteT* sites, int n_sites, const CArgT* state_args, bool step) {
    const int nq = quad_draws(sites, n_sites, step ? sc_step : sc_init);
    if (nq > 0) {
      o << "  uint32_t pf_w[" << nq << "][4];\n  float pf_z[" << nq << "][4];\n";
      emit_prefetch(sites, n_sites, step ? sc_step : sc_init);
    }
    o << "  __device__ __forceinline__ void compute_quad(int64_t jq, const uint32_t (&src)[4], Out (&out)[4], float (&wq)[4]) const {\n";
    emit_quad_body(sites, n_sites, state_args, step, nq > 0);
    o << "  }\n";
  }

  std::string run() {
    const std::string I = std::to_string(impl), D = std::to_string(n_state);
    emit_prelude(o);
    SiteEmitter<CSiteT, CArgT> es{o, impl, 1, step_sites, n_step, "    "};
    SiteEmitter<CSiteT, CArgT> ei{o, impl, 1, init_sites, n_init, "        "};
    es.sc = sc_step;
    ei.sc = sc_init;
    // ---- step policy
    o << "struct GenPolicy {\n  static constexpr bool kEmit = true;\n  static constexpr bool kPeers = " << (peers ? "true" : "false") << ";\n  PlanPolicyArgs a;\n  PlanTables tabs;\n";
    o << "  const int64_t* pd = nullptr;\n  uint32_t tpr = 1;\n";
    o << "  __device__ __forceinline__ void set_peers(const int64_t* d, uint32_t t) { pd = d; tpr = t; }\n";
    o << "  struct Out { float s[" << D << "]; };\n";
    o << "  __device__ __forceinline__ void select_filter(uint64_t off, Key k) {\n";
    o << "    for (int c = 0; c < " << D << "; ++c) { a.prev_state[c] += off; a.state_out[c] += off; }\n";
    o << "    if (a.anc_out) a.anc_out += off; a.step_key = k;\n  }\n";
    o << "  __device__ __forceinline__ float compute(int64_t j, uint32_t src_global, Out& out) const {\n";
    for (int k = 0; k < n_state; ++k) o << "    const float st_" << k << " = src_load<kPeers>(a.prev_state[" << k << "], src_global, pd, tpr);\n";
    if (es.needs_pk() || sc_step) o << "    const Key pkey = slot_key<" << I << ">(a.step_key, (uint64_t)j);\n";
    o << "    float w = 0.0f, sc = 0.0f;\n";
    es.run();
    for (int k = 0; k < n_state; ++k) o << "    out.s[" << k << "] = " << es.arg(next_state[k]) << ";\n";
    o << "    (void)sc;\n    return w;\n  }\n";
    if (impl == 1) emit_quad(step_sites, n_step, next_state, true);
    o << "  __device__ __forceinline__ void store(int64_t j, int64_t out_lo, uint32_t src, const Out& out) const {\n";
    o << "    for (int k = 0; k < " << D << "; ++k) a.state_out[k][j - out_lo] = out.s[k];\n";
    o << "    if (a.anc_out) a.anc_out[j - out_lo] = (int32_t)src;\n  }\n";
    // the lane's four consecutive slots at once: one 16-byte (write-through, for one-filter launches) store per column
    o << "  __device__ __forceinline__ void store_quad(int64_t jq, int64_t out_lo, const uint32_t (&anc)[4], const Out (&o)[4], const bool (&ok)[4]) const {\n";
    o << "    const int64_t k = jq - out_lo;\n";
    o << "    uintptr_t al = (uintptr_t)a.anc_out;\n";
    o << "    for (int c = 0; c < " << D << "; ++c) al |= (uintptr_t)a.state_out[c];\n";
    o << "    if ((al & 15) == 0 && ok[0] && ok[1] && ok[2] && ok[3]) {\n";
    o << "      for (int c = 0; c < " << D << "; ++c) store16_out(a.state_out[c] + k, make_uint4(f2u(o[0].s[c]), f2u(o[1].s[c]), f2u(o[2].s[c]), f2u(o[3].s[c])), a.wt != 0);\n";
    o << "      if (a.anc_out) store16_out(a.anc_out + k, make_uint4(anc[0], anc[1], anc[2], anc[3]), a.wt != 0);\n";
    o << "      return;\n    }\n";
    o << "    for (int u = 0; u < 4; ++u) if (ok[u]) store(jq + u, out_lo, anc[u], o[u]);\n  }\n};\n";
    // two instantiations, as for the hand-written filters: the every-step form carries no ESS decision / keep-your-particle path
    o << "extern \"C\" __global__ __launch_bounds__(256) void gjx_smc_step_kernel(ResampleArgs A, PlanPolicyArgs PA, PlanTables T) {\n";
    o << "  GenPolicy P;\n  P.a = PA;\n  P.tabs = T;\n  resample_body<" << I << ", GenPolicy, false, GenPolicy::kPeers>(A, P);\n}\n";
    o << "extern \"C\" __global__ __launch_bounds__(256) void gjx_smc_step_kernel_adaptive(ResampleArgs A, PlanPolicyArgs PA, PlanTables T) {\n";
    o << "  GenPolicy P;\n  P.a = PA;\n  P.tabs = T;\n  resample_body<" << I << ", GenPolicy, true, GenPolicy::kPeers>(A, P);\n}\n";
    // ---- init kernel: one workgroup per LOCAL tile, like k_lgssm_init
    if (impl == 1) {
      o << "struct GenInitOut { float s[" << D << "]; };\n";
      o << "__device__ __forceinline__ void init_quad(const PlanPolicyArgs& a, const PlanTables& tabs, int64_t jq, GenInitOut (&out)[4], float (&wq)[4]) {\n";
      emit_quad_body(init_sites, n_init, init_state, false);
      o << "}\n";
    }
    o << "extern \"C\" __global__ __launch_bounds__(256) void gjx_smc_init_kernel(PlanPolicyArgs a, uint64_t first_slot, uint64_t n_local, EmitOut em, FilterBatch fb, PlanTables tabs) {\n";
    o << "  uint64_t ltile = blockIdx.x;\n";
    o << "  if (fb.n_filters > 1) {  // several filters per launch: tile of filter f, its key, its outputs\n";
    o << "    const uint32_t f = (uint32_t)(ltile / fb.tiles);\n    ltile -= (uint64_t)f * fb.tiles;\n    a.step_key = fb.step_key[f];\n";
    o << "    for (int k = 0; k < " << D << "; ++k) a.state_out[k] += (uint64_t)f * fb.stride;\n";
    o << "    if (a.anc_out) a.anc_out += (uint64_t)f * fb.stride;\n    select_filter_emit(em, fb, f);\n  }\n";
    o << "  const uint64_t loc = ltile * kTile + 4 * (uint64_t)threadIdx.x;\n  const uint64_t gq = first_slot + loc;\n";
    o << "  float wq[4];\n  bool okq[4];\n  for (int u = 0; u < 4; ++u) okq[u] = loc + u < n_local;\n";
    if (impl == 1) {  // four consecutive slots per lane, one cipher block per one-word draw of the quad
      o << "  {\n    const int64_t jq = (int64_t)gq;\n    GenInitOut out[4];\n";
      o << "    init_quad(a, tabs, jq, out, wq);\n";
      o << "    for (int u = 0; u < 4; ++u) {\n      if (okq[u]) {\n";
      for (int k = 0; k < n_state; ++k) o << "        a.state_out[" << k << "][loc + u] = out[u].s[" << k << "];\n";
      o << "        if (a.anc_out) a.anc_out[loc + u] = (int32_t)(gq + u);\n      }\n    }\n  }\n";
    } else {
      o << "  for (int u = 0; u < 4; ++u) {\n    const uint64_t j = gq + u;\n    wq[u] = 0.0f;\n    {\n";
      if (ei.needs_pk() || sc_init) o << "        const Key pkey = slot_key<" << I << ">(a.step_key, j);\n";
      o << "        float w = 0.0f, sc = 0.0f;\n";
      ei.run();
      for (int k = 0; k < n_state; ++k) o << "        const float ns_" << k << " = " << ei.arg(init_state[k]) << ";\n";
      o << "        (void)sc;\n        wq[u] = w;\n        if (okq[u]) {\n";
      for (int k = 0; k < n_state; ++k) o << "          a.state_out[" << k << "][loc + u] = ns_" << k << ";\n";
      o << "          if (a.anc_out) a.anc_out[loc + u] = (int32_t)j;\n        }\n    }\n  }\n";
    }
    o << "  emit_init_tile(wq, okq, em, loc, first_slot / kTile + ltile);\n}\n";
    return o.str();
  }
};

struct Compiled {
  hipModule_t mod = nullptr;
  hipFunction_t fn = nullptr;
  int state = 0;  // 0 untried, 1 ready, -1 failed
  int block = 256;  // threads per workgroup of the compiled kernel
  int rows_per_block = 1;  // 256-particle rows per workgroup
  std::string key;  // the source this slot holds a reference on (module cache)
  gjx::PlanTables tabs;  // the device tables of THIS plan, in the order the source numbers them (kernel argument)
};

inline bool enabled() {
  const char* e = std::getenv("GJX_PLAN_JIT");
  return !(e && e[0] == '0');
}

// ---- the compiler runs in a CHILD process (gjx_jitc.cpp) ---------------------------------------------------------------
// hiprtc is the whole AMDGPU backend in-process: a backend crash on generated source would be an abort() of the caller
// (it happened once: a constant-folded NaN log-weight; the generator now keeps such constants opaque, but ANY other
// backend bug would do the same).  So generated kernels are compiled by `gjx_jitc`, a helper next to this library: a fresh
// process that never touches the GPU, started with posix_spawn (a child process — the caller is not replaced), source /
// header / code object / log in files of a private temporary directory.  A child that dies => false + a log line
// (callers return GJX_ERR_JIT).  GJX_JIT_INPROC=1 compiles in-process (the round-1/2 behaviour).  A helper that cannot be
// started is GJX_ERR_JIT too (r04: the silent in-process fallback is opt-in, GJX_JIT_INPROC_FALLBACK=1); gjx_jit_routes
// says which route produced the code objects of this process.
inline std::string jitc_path() {
  static const std::string path = [] {
    if (const char* e = std::getenv("GJX_JITC")) return std::string(e);
    Dl_info info;
    if (dladdr((const void*)&kDeviceHeader, &info) && info.dli_fname) {
      std::string p(info.dli_fname);
      const size_t k = p.rfind('/');
      p = (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/gjx_jitc";
      if (access(p.c_str(), X_OK) == 0) return p;
    }
    return std::string();
  }();
  return path;
}
inline bool write_file(const std::string& path, const char* data, size_t n) {
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) return false;
  const bool ok = std::fwrite(data, 1, n, f) == n;
  return std::fclose(f) == 0 && ok;
}
inline bool read_file(const std::string& path, std::string* out) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  char buf[1 << 16];
  size_t n;
  out->clear();
  while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out->append(buf, n);
  std::fclose(f);
  return true;
}
inline std::vector<std::string> compile_options() {
  std::vector<std::string> opts = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17"};
  // GJX_JIT_DEFINE=NAME[=VALUE]: one extra -D for the generated kernel (A/B knob for device-header variants)
  if (const char* d = std::getenv("GJX_JIT_DEFINE")) opts.push_back(std::string("-D") + d);
  // GJX_JIT_OPTS="opt opt ...": extra compiler options for the generated kernel (scheduling experiments)
  if (const char* e = std::getenv("GJX_JIT_OPTS")) {
    std::istringstream is(e);
    for (std::string t; is >> t;) opts.push_back(t);
  }
  return opts;
}
// Which route compiled what (gjx_jit_routes): child = code objects produced by gjx_jitc, inproc = by hiprtc inside this
// process (GJX_JIT_INPROC=1 / GJX_JIT_INPROC_FALLBACK=1 only), child_failures = children that died or rejected a source,
// spawn_failures = helpers that could not be started at all.
struct RouteCounters {
  std::atomic<uint64_t> child{0}, inproc{0}, child_failures{0}, spawn_failures{0};
};
inline RouteCounters& routes() {
  static RouteCounters r;
  return r;
}
// The private directory of THIS process (re-created after a fork: a child of the host program must not share its parent's
// files), removed at exit.  Files inside are named per compilation.
struct JitDir {
  std::string path;
  pid_t owner = 0;
  bool hdr_ok = false;
  void remove_all() {
    if (path.empty() || owner != getpid()) return;  // (a forked child never deletes its parent's directory)
    if (DIR* d = opendir(path.c_str())) {
      while (dirent* e = readdir(d)) {
        if (!std::strcmp(e->d_name, ".") || !std::strcmp(e->d_name, "..")) continue;
        unlink((path + "/" + e->d_name).c_str());
      }
      closedir(d);
    }
    rmdir(path.c_str());
    path.clear();
  }
  ~JitDir() {
    if (!std::getenv("GJX_JIT_KEEP_FILES")) remove_all();
  }
  // -> the directory of the calling process ("" on failure); called under the compile lock
  const std::string& get() {
    if (!path.empty() && owner == getpid()) return path;
    const char* t = std::getenv("TMPDIR");
    std::string d = std::string(t && *t ? t : "/tmp") + "/gjx_jit_XXXXXX";
    path = mkdtemp(&d[0]) ? d : std::string();
    owner = getpid();
    hdr_ok = !path.empty() && write_file(path + "/gjx_device.hpp", kDeviceHeader, sizeof(kDeviceHeader) - 1);
    return path;
  }
};
// The child's environment: the caller's without what would make the helper more than a compiler — a profiler's or
// sanitizer's preloaded library (under rocprofv3 LD_PRELOAD would initialise the GPU inside gjx_jitc and add its output),
// and the ROCm tool variables that go with it.
inline std::vector<char*> child_environ() {
  static const char* const drop[] = {"LD_PRELOAD=", "ROCP_", "ROCPROFILER_", "ROCPROF_", "HSA_TOOLS_", "ROCTX_", "ROCTRACER_"};
  std::vector<char*> env;
  for (char** e = environ; e && *e; ++e) {
    bool skip = false;
    for (const char* d : drop) skip = skip || std::strncmp(*e, d, std::strlen(d)) == 0;
    if (!skip) env.push_back(*e);
  }
  env.push_back(nullptr);
  return env;
}
// -> 1 compiled, 0 compilation failed or the child died (logged), -1 the child could not be started
inline int compile_in_child(const std::string& src, std::string* code) {
  const std::string helper = jitc_path();
  if (helper.empty()) return -1;
  static std::mutex mu;  // compilations are serialised by the module cache's lock anyway
  std::lock_guard<std::mutex> lock(mu);
  static JitDir jd;
  static uint64_t serial = 0;
  const std::string dir = jd.get();
  if (dir.empty() || !jd.hdr_ok) return -1;
  const std::string stem = dir + "/k" + std::to_string(++serial);
  const std::string fsrc = stem + ".hip", fhdr = dir + "/gjx_device.hpp", fout = stem + ".co", flog = stem + ".log";
  if (!write_file(fsrc, src.data(), src.size())) return -1;
  const std::vector<std::string> opts = compile_options();
  std::vector<char*> argv = {const_cast<char*>(helper.c_str()), const_cast<char*>(fsrc.c_str()), const_cast<char*>(fhdr.c_str()),
                             const_cast<char*>(fout.c_str()), const_cast<char*>(flog.c_str())};
  for (const std::string& o : opts) argv.push_back(const_cast<char*>(o.c_str()));
  argv.push_back(nullptr);
  std::vector<char*> env = child_environ();
  pid_t pid = 0;
  // a CHILD process (fork + exec inside posix_spawn): the caller itself is never replaced
  if (posix_spawn(&pid, helper.c_str(), nullptr, nullptr, argv.data(), env.data()) != 0) {
    unlink(fsrc.c_str());
    return -1;
  }
  int status = 0;
  while (waitpid(pid, &status, 0) < 0) {
    if (errno != EINTR) return -1;
  }
  const bool ok = WIFEXITED(status) && WEXITSTATUS(status) == 0 && read_file(fout, code) && !code->empty();
  if (ok) {
    routes().child++;
  } else if (WIFEXITED(status) && WEXITSTATUS(status) == 127) {
    // posix_spawn reports an exec failure of the child as exit status 127: the helper did not start
    unlink(fsrc.c_str());
    return -1;
  } else if (WIFSIGNALED(status)) {
    routes().child_failures++;
    std::fprintf(stderr, "[gjx] the plan compiler (gjx_jitc, pid %d) died with signal %d on a generated kernel; source kept in %s\n",
                 (int)pid, WTERMSIG(status), fsrc.c_str());
  } else {
    routes().child_failures++;
    std::string log;
    (void)read_file(flog, &log);
    std::fprintf(stderr, "[gjx] hiprtc: plan specialisation failed to compile (child status %d)%s\n", WIFEXITED(status) ? WEXITSTATUS(status) : -1,
                 std::getenv("GJX_PLAN_JIT_VERBOSE") ? ":" : "; GJX_PLAN_JIT_VERBOSE=1 prints the compiler log");
    if (std::getenv("GJX_PLAN_JIT_VERBOSE")) std::fprintf(stderr, "%s\n", log.c_str());
  }
  unlink(fout.c_str());
  unlink(flog.c_str());
  if (ok || !WIFSIGNALED(status)) unlink(fsrc.c_str());  // (the source of a crashed compilation is kept for the report)
  return ok ? 1 : 0;
}

// Compile `src` for gfx950; on success `code` holds the code object.
inline bool compile_to_code(const std::string& src, std::string* code) {
  // GJX_PLAN_JIT_DUMP_FILE=path: the source about to be compiled (overwritten per compilation: after a compiler crash the
  // file holds the offending kernel)
  if (const char* f = std::getenv("GJX_PLAN_JIT_DUMP_FILE")) {
    if (FILE* fp = fopen(f, "w")) {
      fwrite(src.data(), 1, src.size(), fp);
      fclose(fp);
    }
  }
  const char* inproc = std::getenv("GJX_JIT_INPROC");
  if (!(inproc && inproc[0] == '1')) {
    const int r = compile_in_child(src, code);
    if (r >= 0) return r == 1;
    routes().spawn_failures++;
    // no silent change of route: a caller that accepts the compiler inside its own address space says so
    const char* fb = std::getenv("GJX_JIT_INPROC_FALLBACK");
    if (!(fb && fb[0] == '1')) {
      std::fprintf(stderr, "[gjx] gjx_jitc (the plan compiler's child process) could not be started%s: GJX_ERR_JIT.  "
                           "GJX_JIT_INPROC_FALLBACK=1 (or GJX_JIT_INPROC=1) compiles inside the calling process instead\n",
                   jitc_path().empty() ? " (no executable gjx_jitc next to the library; GJX_JITC=path overrides)" : "");
      return false;
    }
    static bool warned = false;
    if (!warned) {
      warned = true;
      std::fprintf(stderr, "[gjx] gjx_jitc could not be started: compiling in-process (GJX_JIT_INPROC_FALLBACK=1)\n");
    }
  }
  hiprtcProgram prog;
  const char* hn[] = {"gjx_device.hpp"};
  const char* hs[] = {kDeviceHeader};
  if (hiprtcCreateProgram(&prog, src.c_str(), "gjx_plan.hip", 1, hs, hn) != HIPRTC_SUCCESS) return false;
  const std::vector<std::string> extra = compile_options();
  std::vector<const char*> opts;
  for (const std::string& t : extra) opts.push_back(t.c_str());
  const hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
  if (r != HIPRTC_SUCCESS) {
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    std::string log(ls, 0);
    if (ls) hiprtcGetProgramLog(prog, &log[0]);
    // always say THAT it failed; the full compiler log on request
    std::fprintf(stderr, "[gjx] hiprtc: plan specialisation failed to compile (%s)%s\n", hiprtcGetErrorString(r),
                 std::getenv("GJX_PLAN_JIT_VERBOSE") ? ":" : "; GJX_PLAN_JIT_VERBOSE=1 prints the compiler log");
    if (std::getenv("GJX_PLAN_JIT_VERBOSE")) std::fprintf(stderr, "%s\n", log.c_str());
    hiprtcDestroyProgram(&prog);
    return false;
  }
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  code->assign(cs, 0);
  hiprtcGetCode(prog, &(*code)[0]);
  hiprtcDestroyProgram(&prog);
  routes().inproc++;
  return true;
}
inline bool compile_only(const std::string& src) {
  std::string code;
  return compile_to_code(src, &code) && !code.empty();
}

// The module cache.  Identical sources (same model structure) share one loaded module process-wide, so re-creating a
// plan — or running the same model on another dataset, whose values are launch parameters, not source — does not
// recompile.  Entries are reference-counted by the plans that use them; unreferenced modules stay cached for reuse up to
// GJX_JIT_CACHE_MAX entries (default 64) and are then unloaded least-recently-used first, so a long-lived process that
// keeps generating NEW model structures holds a bounded number of code objects.
struct ModuleCache {
  struct Entry {
    hipModule_t mod = nullptr;
    int refs = 0;
    uint64_t tick = 0;
  };
  std::mutex mu;
  std::unordered_map<std::string, Entry> map;
  uint64_t clock = 0, compiles = 0, evictions = 0;
  size_t cap = [] {
    const char* e = std::getenv("GJX_JIT_CACHE_MAX");
    const long v = e ? atol(e) : 64;
    return (size_t)(v < 1 ? 1 : v);
  }();
  static ModuleCache& get() {
    static ModuleCache c;
    return c;
  }
  // -> the loaded module of `src` with one more reference, or nullptr (compile / load failure, logged)
  hipModule_t acquire(const std::string& src) {
    std::lock_guard<std::mutex> lock(mu);
    auto it = map.find(src);
    if (it == map.end()) {
      std::string code;
      if (!compile_to_code(src, &code)) return nullptr;
      hipModule_t mod = nullptr;
      const hipError_t le = hipModuleLoadData(&mod, code.data());
      if (le != hipSuccess) {
        (void)hipGetLastError();
        std::fprintf(stderr, "[gjx] hipModuleLoadData failed for a specialised plan kernel: %s\n", hipGetErrorString(le));
        return nullptr;
      }
      ++compiles;
      it = map.emplace(src, Entry{mod, 1, ++clock}).first;  // referenced before anything is evicted
      evict_locked();
      return mod;
    }
    it->second.refs++;
    it->second.tick = ++clock;
    return it->second.mod;
  }
  void release(const std::string& src) {
    if (src.empty()) return;
    std::lock_guard<std::mutex> lock(mu);
    auto it = map.find(src);
    if (it != map.end() && it->second.refs > 0) it->second.refs--;
    evict_locked();
  }
  void evict_locked() {
    while (map.size() > cap) {
      auto victim = map.end();
      for (auto it = map.begin(); it != map.end(); ++it)
        if (it->second.refs == 0 && (victim == map.end() || it->second.tick < victim->second.tick)) victim = it;
      if (victim == map.end()) return;  // everything is in use
      // no plan references the module, but launches of its kernels may still be in flight on some stream: an eviction
      // is rare (a process that keeps generating new model structures), so wait for the device before unloading
      (void)hipDeviceSynchronize();
      (void)hipModuleUnload(victim->second.mod);
      map.erase(victim);
      ++evictions;
    }
  }
};
inline void release(Compiled* c) {
  ModuleCache::get().release(c->key);
  c->key.clear();
  c->fn = nullptr;
  c->mod = nullptr;
}
inline bool compile(const std::string& src, int impl, Compiled* out, const char* kernel = nullptr) {
  release(out);  // (a slot that is being rebuilt, e.g. without the occupancy hint)
  hipModule_t mod = ModuleCache::get().acquire(src);
  if (!mod) return false;
  out->key = src;
  hipFunction_t fn = nullptr;
  const hipError_t ge =
      hipModuleGetFunction(&fn, mod, kernel ? kernel : (impl == 0 ? "gjx_plan_kernel_threefry" : "gjx_plan_kernel_philox"));
  if (ge != hipSuccess) {
    (void)hipGetLastError();
    std::fprintf(stderr, "[gjx] hipModuleGetFunction failed for a specialised plan kernel: %s\n", hipGetErrorString(ge));
    release(out);
    return false;
  }
  out->mod = nullptr;  // owned by the cache
  out->fn = fn;
  return true;
}
struct CompiledSmc {
  hipFunction_t step = nullptr, step_adaptive = nullptr, init = nullptr;
  int state = 0;  // 0 untried, 1 ready, -1 failed
  std::string key;
  gjx::PlanTables tabs;  // the device tables of THIS plan (kernel argument of both kernels)
};
inline void release_smc(CompiledSmc* c) {
  ModuleCache::get().release(c->key);
  c->key.clear();
  c->step = c->step_adaptive = c->init = nullptr;
}
inline bool compile_smc(const std::string& src, CompiledSmc* out) {
  hipModule_t mod = ModuleCache::get().acquire(src);
  if (!mod) return false;
  out->key = src;
  if (hipModuleGetFunction(&out->step, mod, "gjx_smc_step_kernel") != hipSuccess ||
      hipModuleGetFunction(&out->step_adaptive, mod, "gjx_smc_step_kernel_adaptive") != hipSuccess ||
      hipModuleGetFunction(&out->init, mod, "gjx_smc_init_kernel") != hipSuccess) {
    (void)hipGetLastError();
    std::fprintf(stderr, "[gjx] hipModuleGetFunction failed for a generated SMC kernel\n");
    release_smc(out);
    return false;
  }
  return true;
}

}  // namespace gjx_jit
