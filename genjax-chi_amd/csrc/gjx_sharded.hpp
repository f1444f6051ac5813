// gjx_sharded.hpp — the native multi-rank driver of the sharded bootstrap filter (include/gjx.h: gjx_comm,
// gjx_smc_sharded_run_*, gjx_comm_lse_combine).
//
// Orchestration only: every arithmetic step is one of the public per-step entry points (gjx_smc_*_step_a, gjx_smc_step_b,
// gjx_smc_source_ranges, gjx_smc_finish, gjx_lse_combine) and every exchange goes through the abstract Transport below, so
// the same driver serves libgjx_hip.so (RCCL ranks; virtual ranks on one device) and the CPU oracle build (virtual ranks),
// which is how the protocol is tested without a multi-GPU box.  A build provides `Mem` — copy / max on ITS memory.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/gjx.h"

namespace gjx_sharded {

struct Seg {  // elements [a, b) of a global column travel to / from `peer`
  int peer;
  uint64_t a, b;
};

struct Transport {
  int rank = 0, world = 1;
  virtual ~Transport() {}
  // element-wise max over the ranks' buffers, result in every rank's buffer
  virtual int allreduce_max_f32(float* buf, size_t n, gjx_stream s) = 0;
  // in place: rank r's block is full + r * bytes_per_rank
  virtual int allgather(void* full, size_t bytes_per_rank, gjx_stream s) = 0;
  // slices keep their global position on both sides: cols[c] + a * elem .. cols[c] + b * elem
  virtual int exchange(void* const* cols, int n_cols, size_t elem, const Seg* sends, int ns, const Seg* recvs, int nr,
                       gjx_stream s) = 0;
  virtual int stream_sync(gjx_stream s) = 0;
};

// ---- virtual ranks: threads of one process ------------------------------------------------------------------------
// Ranks share one device and ONE stream, so the enqueue order on the stream is the execution order: a host barrier
// between "everyone has enqueued its writes" and "everyone enqueues its reads" is all the ordering the copies need.
struct Group {
  int world;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  uint64_t generation = 0;
  bool broken = false;
  std::vector<const void*> slot;  // one posted pointer per rank
  explicit Group(int w) : world(w), slot((size_t)w, nullptr) {}
  // -> false if a rank did not arrive within the timeout (another rank failed): the group is broken for good
  bool barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (broken) return false;
    const uint64_t gen = generation;
    if (++arrived == world) {
      arrived = 0;
      ++generation;
      cv.notify_all();
      return true;
    }
    if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return generation != gen || broken; })) {
      broken = true;
      cv.notify_all();
      return false;
    }
    return !broken;
  }
};

// Mem: int copy(void* dst, const void* src, size_t bytes, gjx_stream); int max_f32(float* dst, const float* const* srcs,
// int world, size_t n, gjx_stream) (dst may be one of srcs? no: dst is scratch); void* scratch(size_t bytes); int sync(gjx_stream)
template <class Mem>
struct LocalTransport : Transport {
  Group* g;
  Mem mem;
  struct ExPost {
    void* const* cols;
    const Seg* sends;
    int ns;
  };
  ExPost post;
  LocalTransport(Group* grp, int r) : g(grp) {
    rank = r;
    world = grp->world;
  }
  int allreduce_max_f32(float* buf, size_t n, gjx_stream s) override {
    if (world == 1) return GJX_OK;
    g->slot[(size_t)rank] = buf;
    if (!g->barrier()) return GJX_ERR_LAUNCH;  // every rank's values are enqueued
    float* tmp = (float*)mem.scratch(n * sizeof(float));
    if (!tmp) return GJX_ERR_WORKSPACE;
    std::vector<const float*> srcs((size_t)world);
    for (int r = 0; r < world; ++r) srcs[(size_t)r] = (const float*)g->slot[(size_t)r];
    int rc = mem.max_f32(tmp, srcs.data(), world, n, s);
    if (!g->barrier()) return GJX_ERR_LAUNCH;  // every rank has read every buffer
    if (rc == GJX_OK) rc = mem.copy(buf, tmp, n * sizeof(float), s);
    return rc;
  }
  int allgather(void* full, size_t bytes, gjx_stream s) override {
    if (world == 1) return GJX_OK;
    g->slot[(size_t)rank] = full;
    if (!g->barrier()) return GJX_ERR_LAUNCH;
    int rc = GJX_OK;
    for (int r = 0; r < world && rc == GJX_OK; ++r)
      if (r != rank) rc = mem.copy((char*)full + (size_t)r * bytes, (const char*)g->slot[(size_t)r] + (size_t)r * bytes, bytes, s);
    if (!g->barrier()) return GJX_ERR_LAUNCH;
    return rc;
  }
  int exchange(void* const* cols, int n_cols, size_t elem, const Seg* sends, int ns, const Seg* recvs, int nr,
               gjx_stream s) override {
    if (world == 1) return GJX_OK;
    post = ExPost{cols, sends, ns};
    g->slot[(size_t)rank] = &post;
    if (!g->barrier()) return GJX_ERR_LAUNCH;
    int rc = GJX_OK;
    for (int i = 0; i < nr && rc == GJX_OK; ++i) {  // pull every received slice from its owner's buffers
      const ExPost* peer = (const ExPost*)g->slot[(size_t)recvs[i].peer];
      bool matched = false;  // the owner must have listed the same slice for us (both sides derive the same ranges)
      for (int k = 0; k < peer->ns; ++k)
        matched = matched || (peer->sends[k].peer == rank && peer->sends[k].a == recvs[i].a && peer->sends[k].b == recvs[i].b);
      if (!matched) rc = GJX_ERR_INVALID;
      for (int c = 0; c < n_cols && rc == GJX_OK; ++c)
        rc = mem.copy((char*)cols[c] + recvs[i].a * elem, (const char*)peer->cols[c] + recvs[i].a * elem,
                      (size_t)(recvs[i].b - recvs[i].a) * elem, s);
    }
    if (!g->barrier()) return GJX_ERR_LAUNCH;
    return rc;
  }
  int stream_sync(gjx_stream s) override { return mem.sync(s); }
};

// Tickets of the range kernel (gjx_smc_source_ranges): process-wide and never reused, so a ticket left in a caller's
// `ranges` buffer by an earlier run — of any model — cannot be mistaken for the current one.
inline std::atomic<int64_t> g_ticket_source{0};

// ---- the sharded filter -----------------------------------------------------------------------------------------------
// StepA: int(int t, int cur, int prv) -> status: step A of step t for the rank's own slots (buffers by parity).
template <class StepA>
int sharded_run(Transport& T, const gjx_smc_config* cfg, int n_state, size_t state_elem, const gjx_sharded_io* io,
                StepA step_a, gjx_stream s) {
  const int world = T.world, rank = T.rank;
  const uint64_t tile = gjx_smc_tile(), N = cfg->n_total, nl = cfg->n_local, lo = cfg->first_slot;
  if (!io || !io->tile_sums || !io->max_partials || !io->out_max || !io->out_q || !io->logw[0] || !io->logw[1] ||
      cfg->n_filters > 1 || N % ((uint64_t)world * tile) != 0 || nl != N / (uint64_t)world || lo != (uint64_t)rank * nl ||
      (io->shuffle == 0 && world > 1 && !io->ranges) || world > 64 || state_elem != 4 /* every state column is 4-byte */)
    return GJX_ERR_INVALID;
  const uint64_t nt = gjx_num_tiles(N), tiles_local = nl / tile;
  const bool adaptive = cfg->ess_threshold > 0.0f && cfg->ess_threshold < 1.0f;
  if (adaptive && !cfg->tile_ess) return GJX_ERR_INVALID;
  uint64_t received = 0;
  int64_t ticket = 0;
  int rc = GJX_OK;
  for (int t = 0; t < cfg->n_steps && rc == GJX_OK; ++t) {
    const int cur = t & 1, prv = cur ^ 1;
    rc = step_a(t, cur, prv);
    if (rc) break;
    if ((rc = T.allreduce_max_f32(io->max_partials, (size_t)nt, s))) break;
    if ((rc = gjx_smc_step_b(cfg, io->logw[cur] + lo, io->max_partials, io->out_max + t, io->tile_sums, s))) break;
    if ((rc = T.allgather(io->tile_sums, (size_t)tiles_local * sizeof(uint64_t), s))) break;
    if (adaptive && (rc = T.allgather(cfg->tile_ess, (size_t)tiles_local * 2 * sizeof(uint64_t), s))) break;
    if (t + 1 >= cfg->n_steps || world == 1) continue;
    // ---- the ancestor shuffle: make the source ranges of the next resampling present on every rank
    void* cols[GJX_SMC_MAX_STATE + 1];
    for (int k = 0; k < n_state; ++k) cols[k] = io->state[cur][k];
    cols[n_state] = io->logw[cur];
    if (io->shuffle == 1) {
      for (int c = 0; c <= n_state && rc == GJX_OK; ++c) rc = T.allgather(cols[c], (size_t)nl * 4, s);
      received += N - nl;
      continue;
    }
    ticket = ++g_ticket_source;
    if ((rc = gjx_smc_source_ranges(cfg, io->tile_sums, world, ticket, io->ranges, s))) break;
    // the range kernel stores its words straight into pinned host memory, the ticket last with a system-scope release:
    // poll the ticket instead of synchronising the stream; if it does not show up soon, wait for the stream
    volatile int64_t* rh = io->ranges;
    const auto t0 = std::chrono::steady_clock::now();
    bool synced = false;
    while (rh[2 * world] != ticket) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(synced ? 5000 : 50)) {
        if (synced) return GJX_ERR_LAUNCH;
        if ((rc = T.stream_sync(s))) return rc;
        synced = true;
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    Seg sends[64], recvs[64];
    int ns = 0, nr = 0;
    const uint64_t hi = lo + nl;
    for (int j = 0; j < world; ++j) {
      if (j == rank) continue;
      const uint64_t ja = (uint64_t)rh[2 * j] * tile, jb = (uint64_t)rh[2 * j + 1] * tile;  // what rank j needs
      uint64_t a = ja > lo ? ja : lo, b = jb < hi ? jb : hi;                                // ... of my block
      if (a < b) sends[ns++] = Seg{j, a, b};
      const uint64_t ma = (uint64_t)rh[2 * rank] * tile, mb = (uint64_t)rh[2 * rank + 1] * tile;  // what I need
      const uint64_t jl = (uint64_t)j * nl;
      a = ma > jl ? ma : jl;
      b = mb < jl + nl ? mb : jl + nl;
      if (a < b) {
        recvs[nr++] = Seg{j, a, b};
        received += b - a;
      }
    }
    rc = T.exchange(cols, n_state + 1, 4, sends, ns, recvs, nr, s);
  }
  if (rc == GJX_OK) rc = gjx_smc_finish(cfg, io->tile_sums, io->out_q + (cfg->n_steps - 1), s);
  if (io->received) *io->received = received;
  return rc;
}

inline int run_lgssm(Transport& T, const gjx_smc_config* cfg, const gjx_lgssm* model, const float* y, const gjx_sharded_io* io,
                     gjx_stream s) {
  if (!cfg || !model || !y || !io || !io->state[0][0] || !io->state[1][0]) return GJX_ERR_INVALID;
  const uint64_t lo = cfg->first_slot, nl = cfg->n_local;
  auto step_a = [&](int t, int cur, int prv) {
    return gjx_smc_lgssm_step_a(cfg, model, t, y[t], t ? (const float*)io->state[prv][0] : nullptr, t ? io->logw[prv] : nullptr,
                                t ? io->out_max + (t - 1) : nullptr, t ? io->tile_sums : nullptr, t ? io->out_q + (t - 1) : nullptr,
                                (float*)io->state[cur][0] + lo, io->logw[cur] + lo, io->max_partials,
                                io->ancestors ? io->ancestors + (size_t)t * nl : nullptr, s);
  };
  return sharded_run(T, cfg, 1, sizeof(float), io, step_a, s);
}
inline int run_hmm(Transport& T, const gjx_smc_config* cfg, const gjx_hmm* model, const int32_t* y, const uint32_t* trans_alias,
                   const float* obs_logp, const gjx_sharded_io* io, gjx_stream s) {
  if (!cfg || !model || !y || !io || !trans_alias || !obs_logp || !io->state[0][0] || !io->state[1][0]) return GJX_ERR_INVALID;
  const uint64_t lo = cfg->first_slot, nl = cfg->n_local;
  auto step_a = [&](int t, int cur, int prv) {
    return gjx_smc_hmm_step_a(cfg, model, t, y[t], t ? (const int32_t*)io->state[prv][0] : nullptr, t ? io->logw[prv] : nullptr,
                              t ? io->out_max + (t - 1) : nullptr, t ? io->tile_sums : nullptr, t ? io->out_q + (t - 1) : nullptr,
                              trans_alias, obs_logp, (int32_t*)io->state[cur][0] + lo, io->logw[cur] + lo, io->max_partials,
                              io->ancestors ? io->ancestors + (size_t)t * nl : nullptr, s);
  };
  return sharded_run(T, cfg, 1, sizeof(int32_t), io, step_a, s);
}
inline int run_plan(Transport& T, const gjx_smc_config* cfg, gjx_smc_plan* plan, int n_state, int n_obs, const float* obs,
                    const gjx_sharded_io* io, gjx_stream s) {
  if (!cfg || !plan || !io || (n_obs > 0 && !obs) || n_state < 1 || n_state > GJX_SMC_MAX_STATE) return GJX_ERR_INVALID;
  for (int k = 0; k < n_state; ++k)
    if (!io->state[0][k] || !io->state[1][k]) return GJX_ERR_INVALID;
  const uint64_t lo = cfg->first_slot, nl = cfg->n_local;
  auto step_a = [&](int t, int cur, int prv) {
    const float* prev[GJX_SMC_MAX_STATE];
    float* own[GJX_SMC_MAX_STATE];
    for (int k = 0; k < n_state; ++k) {
      prev[k] = (const float*)io->state[prv][k];
      own[k] = (float*)io->state[cur][k] + lo;
    }
    return gjx_smc_plan_step_a(cfg, plan, t, n_obs ? obs + (size_t)t * (size_t)n_obs : nullptr, t ? prev : nullptr,
                               t ? io->logw[prv] : nullptr, t ? io->out_max + (t - 1) : nullptr, t ? io->tile_sums : nullptr,
                               t ? io->out_q + (t - 1) : nullptr, own, io->logw[cur] + lo, io->max_partials,
                               io->ancestors ? io->ancestors + (size_t)t * nl : nullptr, s);
  };
  return sharded_run(T, cfg, n_state, sizeof(float), io, step_a, s);
}

// log-marginal of sharded importance passes: all-gather of the shards' records, then the exact merge
inline int lse_combine(Transport& T, const uint64_t* records, int32_t n_batch, uint64_t* gathered, int32_t* out_e,
                       uint64_t* out_q, float* out_lse, gjx_stream s, int (*copy)(void*, const void*, size_t, gjx_stream)) {
  if (!records || !gathered || n_batch < 1) return GJX_ERR_INVALID;
  const size_t bytes = (size_t)n_batch * GJX_LSE_RECORD_WORDS * sizeof(uint64_t);
  int rc = copy((char*)gathered + (size_t)T.rank * bytes, records, bytes, s);
  if (rc) return rc;
  if ((rc = T.allgather(gathered, bytes, s))) return rc;
  return gjx_lse_combine(gathered, T.world, (uint64_t)n_batch * GJX_LSE_RECORD_WORDS, n_batch, GJX_LSE_RECORD_WORDS, out_e, out_q,
                         out_lse, nullptr, s);
}

}  // namespace gjx_sharded

// The C objects (shared by both builds).
struct gjx_comm_group {
  gjx_sharded::Group g;
  explicit gjx_comm_group(int w) : g(w) {}
};
struct gjx_comm {
  gjx_sharded::Transport* t = nullptr;
  ~gjx_comm() { delete t; }
};
