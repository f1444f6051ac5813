// gjx_sharded.hpp — the native multi-rank driver of the sharded bootstrap filter (include/gjx.h: gjx_comm,
// gjx_smc_sharded_run_*, gjx_comm_lse_combine).
//
// Orchestration only: every arithmetic step is one of the public per-step entry points (gjx_smc_*_step,
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
#include <new>
#include <thread>
#include <vector>

#include "../../include/gjx.h"

namespace gjx_sharded {

using Seg = gjx_seg;  // elements [a, b) of a global column travel to / from `peer`
static_assert(sizeof(size_t) == sizeof(uint64_t), "gjx_exchange_fn passes element sizes as uint64_t");

struct Transport {
  int rank = 0, world = 1;
  virtual ~Transport() {}
  // in place: rank r's block is full + r * bytes_per_rank
  virtual int allgather(void* full, size_t bytes_per_rank, gjx_stream s) = 0;
  // slices keep their global position on both sides.  Segments are ranges of PARTICLES (multiples of the tile); column c
  // holds one element of elems[c] bytes per units[c] particles (1: a per-particle column; the tile size: a per-tile array
  // such as the sub-prefixes), so its slice is cols[c] + (a / units[c]) * elems[c] .. cols[c] + (b / units[c]) * elems[c]
  virtual int exchange(void* const* cols, const size_t* elems, const size_t* units, int n_cols, const Seg* sends, int ns,
                       const Seg* recvs, int nr, gjx_stream s) = 0;
  virtual int stream_sync(gjx_stream s) = 0;
  // this rank has failed and will not take part in further collectives: release the peers (best effort)
  virtual void abort() {}
  // r04, the peer transport (gjx.h: gjx_comm_init_peers): non-null = no collective exists, the steps read their peers' arenas
  virtual const gjx_smc_peers* peers() const { return nullptr; }
  // ... virtual ranks sharing one stream: every rank's signal is ENQUEUED before any rank enqueues the launch that waits for it
  virtual bool enqueue_barrier() { return true; }
  int wait_launch = 0;  // a one-workgroup wait launch in front of every step (ranks that share a device as processes)
  bool serial_stream = false;  // the ranks' launches execute in ENQUEUE order on one stream (virtual ranks on a device): a
                               // launch that waits for a peer must not also carry this rank's signal
  uint64_t epoch = 0;   // every rank's arrival words are >= epoch once its previous run is complete (0: freshly allocated)
};
inline size_t seg_off(uint64_t a, size_t elem, size_t unit) { return (size_t)(a / unit) * elem; }

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

// Mem: int copy(void* dst, const void* src, size_t bytes, gjx_stream); int sync(gjx_stream)
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
  int exchange(void* const* cols, const size_t* elems, const size_t* units, int n_cols, const Seg* sends, int ns, const Seg* recvs,
               int nr, gjx_stream s) override {
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
      for (int c = 0; c < n_cols && rc == GJX_OK; ++c) {
        const size_t a = seg_off(recvs[i].a, elems[c], units[c]), b = seg_off(recvs[i].b, elems[c], units[c]);
        rc = mem.copy((char*)cols[c] + a, (const char*)peer->cols[c] + a, b - a, s);
      }
    }
    if (!g->barrier()) return GJX_ERR_LAUNCH;
    return rc;
  }
  int stream_sync(gjx_stream s) override { return mem.sync(s); }
  void abort() override {
    std::lock_guard<std::mutex> lk(g->mu);
    g->broken = true;
    g->cv.notify_all();
  }
};

// ---- the caller's collectives (gjx_comm_init_callbacks) ---------------------------------------------------------------
struct CallbackTransport : Transport {
  gjx_allgather_fn ag;
  gjx_exchange_fn ex;
  gjx_stream_sync_fn sy;
  void* user;
  int allgather(void* full, size_t bytes, gjx_stream s) override { return world == 1 ? GJX_OK : ag(user, full, (uint64_t)bytes, s); }
  int exchange(void* const* cols, const size_t* elems, const size_t* units, int n_cols, const Seg* sends, int ns, const Seg* recvs,
               int nr, gjx_stream s) override {
    if (world == 1 || (!ns && !nr)) return GJX_OK;
    return ex(user, cols, reinterpret_cast<const uint64_t*>(elems), reinterpret_cast<const uint64_t*>(units), n_cols, sends, ns,
              recvs, nr, s);
  }
  int stream_sync(gjx_stream s) override { return sy ? sy(user, s) : GJX_OK; }
};
inline int comm_init_callbacks(int rank, int world, gjx_allgather_fn ag, gjx_exchange_fn ex, gjx_stream_sync_fn sy, void* user,
                               Transport** out) {
  if (!out || world < 1 || world > 64 || rank < 0 || rank >= world || (world > 1 && (!ag || !ex))) return GJX_ERR_INVALID;
  CallbackTransport* t = new (std::nothrow) CallbackTransport;
  if (!t) return GJX_ERR_LAUNCH;
  t->rank = rank; t->world = world; t->ag = ag; t->ex = ex; t->sy = sy; t->user = user;
  *out = t;
  return GJX_OK;
}

// ---- the peer transport (gjx_comm_init_peers) -----------------------------------------------------------------------------
struct PeerTransport : Transport {
  gjx_smc_peers p;
  Group* g = nullptr;  // virtual ranks of one process sharing a stream (tests): host barriers order their enqueues
  int allgather(void*, size_t, gjx_stream) override { return GJX_ERR_UNSUPPORTED; }
  int exchange(void* const*, const size_t*, const size_t*, int, const Seg*, int, const Seg*, int, gjx_stream) override { return GJX_ERR_UNSUPPORTED; }
  int stream_sync(gjx_stream) override { return GJX_OK; }
  const gjx_smc_peers* peers() const override { return &p; }
  bool enqueue_barrier() override { return g ? g->barrier() : true; }
  void abort() override {
    if (!g) return;
    std::lock_guard<std::mutex> lk(g->mu);
    g->broken = true;
    g->cv.notify_all();
  }
};
inline int comm_init_peers(const gjx_smc_peers* peers, Group* g, int wait_launch, bool serial_stream, Transport** out) {
  if (!out || !peers || peers->world < 2 || peers->world > GJX_MAX_PEERS || peers->rank < 0 || peers->rank >= peers->world ||
      !peers->flags || !peers->error || peers->delta[peers->rank] != 0 || (g && g->world != peers->world))
    return GJX_ERR_INVALID;
  PeerTransport* t = new (std::nothrow) PeerTransport;
  if (!t) return GJX_ERR_LAUNCH;
  t->rank = peers->rank; t->world = peers->world; t->p = *peers; t->g = g; t->wait_launch = wait_launch;
  t->serial_stream = serial_stream;
  t->p.signal_value = 0;
  *out = t;
  return GJX_OK;
}

// Tickets of the range kernel (gjx_smc_source_ranges): process-wide and never reused, so a ticket left in a caller's
// `ranges` buffer by an earlier run — of any model — cannot be mistaken for the current one.
inline std::atomic<int64_t> g_ticket_source{0};

// ---- the sharded filter -----------------------------------------------------------------------------------------------
// Step: int(const gjx_smc_config* cfg, int t, const gjx_smc_pop* prev, const gjx_smc_pop* out, int32_t* prev_e, uint64_t*
// prev_q, int32_t* anc) -> status: ONE launch for the rank's own slots.
//
// Collective transports (RCCL, the caller's callbacks, virtual ranks by copies), r04: per step that launch, then ONE
// all-gather — the 16-byte tile records; for ESS-adaptive filters records and ESS sums packed into one message (pack /
// all-gather / unpack: two tiny launches instead of a second collective) — then the ancestor shuffle, which moves, for the
// range a rank's slots draw from, the state columns, the fixed-point weights AND the tiles' sub-prefixes (128 bytes per
// tile: only the window scan of the tiles actually read needs them; r03 all-gathered them to everybody).
template <class Step>
int sharded_steps(Transport& T, const gjx_smc_config* cfg, int n_state, const gjx_sharded_io* io, Step step, gjx_stream s,
                  uint64_t* received_out) {
  const int world = T.world, rank = T.rank;
  const uint64_t tile = gjx_smc_tile(), N = cfg->n_total, nl = cfg->n_local, lo = cfg->first_slot;
  const uint64_t tiles_local = nl / tile;
  const bool adaptive = cfg->ess_threshold > 0.0f && cfg->ess_threshold < 1.0f;
  const int n_steps = cfg->n_steps;
  uint64_t& received = *received_out;
  int rc = GJX_OK;
  for (int t = 0; t < n_steps && rc == GJX_OK; ++t) {
    const int cur = t & 1, prv = cur ^ 1;
    const gjx_smc_pop& full = io->pop[cur];
    gjx_smc_pop out = full;  // the rank's own block of the local arrays; records / ESS sums stay global
    for (int k = 0; k < n_state; ++k) out.state[k] = (char*)full.state[k] + lo * 4;
    out.qw = full.qw + lo;
    out.logw = (adaptive || t == n_steps - 1) && full.logw ? full.logw + lo : nullptr;
    if ((rc = step(cfg, t, &io->pop[prv], &out, t ? io->out_e + (t - 1) : nullptr, t ? io->out_q + (t - 1) : nullptr,
                   io->ancestors ? io->ancestors + (size_t)t * nl : nullptr)))
      break;
    // ---- ONE all-gather per step
    if (world > 1 && adaptive) {
      if ((rc = gjx_smc_records_pack(cfg, world, 0, full.recs, full.ess, io->stage, s))) break;
      if ((rc = T.allgather(io->stage, (size_t)tiles_local * (sizeof(gjx_tile_rec) + sizeof(gjx_tile_ess)), s))) break;
      if ((rc = gjx_smc_records_pack(cfg, world, 1, full.recs, full.ess, io->stage, s))) break;
    } else if ((rc = T.allgather(full.recs, (size_t)tiles_local * sizeof(gjx_tile_rec), s))) {
      break;
    }
    if (t + 1 >= n_steps || world == 1) continue;
    // ---- the ancestor shuffle: make the source ranges of the next resampling present on every rank
    void* cols[GJX_SMC_MAX_STATE + 3];
    size_t elems[GJX_SMC_MAX_STATE + 3], units[GJX_SMC_MAX_STATE + 3];
    int nc = 0;
    for (int k = 0; k < n_state; ++k) { cols[nc] = full.state[k]; elems[nc] = 4; units[nc++] = 1; }
    cols[nc] = full.qw; elems[nc] = 4; units[nc++] = 1;
    if (adaptive) { cols[nc] = full.logw; elems[nc] = 4; units[nc++] = 1; }
    cols[nc] = full.subs; elems[nc] = sizeof(gjx_tile_sub); units[nc++] = (size_t)tile;
    if (io->shuffle == 1) {
      for (int c = 0; c < nc && rc == GJX_OK; ++c) rc = T.allgather(cols[c], (size_t)(nl / units[c]) * elems[c], s);
      received += N - nl;
      continue;
    }
    const int64_t ticket = ++g_ticket_source;
    if ((rc = gjx_smc_source_ranges(cfg, full.recs, full.ess, world, ticket, io->ranges, s))) break;
    // the range kernel stores its words straight into pinned host memory, the ticket last with a system-scope release:
    // poll the ticket instead of synchronising the stream; if it does not show up soon, wait for the stream
    volatile int64_t* rh = io->ranges;
    const auto t0 = std::chrono::steady_clock::now();
    bool synced = false;
    while (rh[2 * world] != ticket && rc == GJX_OK) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(synced ? 5000 : 50)) {
        if (synced) rc = GJX_ERR_LAUNCH;
        else if ((rc = T.stream_sync(s)) == GJX_OK) synced = true;
      }
    }
    if (rc) break;
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
    rc = T.exchange(cols, elems, units, nc, sends, ns, recvs, nr, s);
  }
  if (rc == GJX_OK)
    rc = gjx_smc_finish(cfg, io->pop[(n_steps - 1) & 1].recs, io->out_e + (n_steps - 1), io->out_q + (n_steps - 1), s);
  return rc;
}

// The peer transport (r04; gjx.h: gjx_smc_peers, gjx_comm_init_peers): no collective and nothing decided on the host.
// io->pop[2] are this rank's arrays INSIDE its arena (global-size, like the collective transports' — only the rank's own block
// of the per-particle arrays and the whole record arrays are ever written).  Per step t:
//   [wait launch for base + t — only for t = 0 (the init kernels do not wait themselves) or when T.wait_launch]
//   ONE step launch: its workgroups wait (bounded) until every rank's arrival word is >= base + t, then resample reading
//                    remote source windows where they live, propagate, weight, store the rank's own block;
//   ONE signal launch: the rank's tile records (+ ESS sums) into every peer's arena, system-scope release, arrival word
//                    base + t + 1 into every rank's flags.
// Closing: wait for base + T, merge the last records (all of them are in this rank's arena), arrival word base + T + 1 —
// what step 0 of the NEXT run on this communicator waits for (a peer may deposit next-run records only after this rank
// has merged the last ones).  The arrival words only grow; `epoch` carries the count from run to run.
template <class Step>
int sharded_steps_peers(Transport& T, const gjx_smc_config* cfg0, int n_state, const gjx_sharded_io* io, Step step, gjx_stream s) {
  gjx_smc_peers P = *T.peers();
  gjx_smc_config cfg = *cfg0;
  cfg.peers = &P;
  const uint64_t tile = gjx_smc_tile(), nl = cfg.n_local, lo = cfg.first_slot;
  const uint64_t tiles_local = nl / tile, tile0 = lo / tile;
  const bool adaptive = cfg.ess_threshold > 0.0f && cfg.ess_threshold < 1.0f;
  const int n_steps = cfg.n_steps;
  const uint64_t base = T.epoch;
  T.epoch = base + (uint64_t)n_steps + 1;  // (whatever happens below: the next run must not wait for words of this one)
  // a step's first launch can carry the previous step's signal (large populations: the group-record launch waits anyway) —
  // unless a wait launch goes in front of it, or all ranks' launches share one stream (a launch that both signals and waits
  // would wait for signals queued behind it)
  const bool defer = gjx_smc_peer_signal_fused(&cfg) != 0 && !T.wait_launch && !T.serial_stream;
  int rc = GJX_OK;
  for (int t = 0; t < n_steps && rc == GJX_OK; ++t) {
    const int cur = t & 1, prv = cur ^ 1;
    const gjx_smc_pop& full = io->pop[cur];
    gjx_smc_pop out = full;
    for (int k = 0; k < n_state; ++k) out.state[k] = (char*)full.state[k] + lo * 4;
    out.qw = full.qw + lo;
    out.logw = (adaptive || t == n_steps - 1) && full.logw ? full.logw + lo : nullptr;
    P.wait_value = base + (uint64_t)t;
    if (!T.enqueue_barrier()) return GJX_ERR_LAUNCH;
    if ((t == 0 || T.wait_launch) && (rc = gjx_smc_peer_wait(&P, P.wait_value, s))) break;
    rc = step(&cfg, t, &io->pop[prv], &out, t ? io->out_e + (t - 1) : nullptr, t ? io->out_q + (t - 1) : nullptr,
              io->ancestors ? io->ancestors + (size_t)t * nl : nullptr);
    P.signal_value = 0;  // (a deferred signal has gone out with that step's first launch)
    if (rc) break;
    if (defer && t >= 0 && t + 1 < n_steps) {  // step t + 1's first launch deposits step t's records and raises the word
      P.signal_recs = full.recs;
      P.signal_ess = adaptive ? full.ess : nullptr;
      P.signal_first_tile = tile0;
      P.signal_n_tiles = tiles_local;
      P.signal_value = base + (uint64_t)t + 1;
    } else {
      rc = gjx_smc_peer_signal(&P, full.recs, adaptive ? full.ess : nullptr, tile0, tiles_local, base + (uint64_t)t + 1, s);
    }
  }
  if (rc) return rc;
  if (!T.enqueue_barrier()) return GJX_ERR_LAUNCH;
  if ((rc = gjx_smc_peer_wait(&P, base + (uint64_t)n_steps, s))) return rc;
  if ((rc = gjx_smc_finish(&cfg, io->pop[(n_steps - 1) & 1].recs, io->out_e + (n_steps - 1), io->out_q + (n_steps - 1), s))) return rc;
  return gjx_smc_peer_signal(&P, nullptr, nullptr, 0, 0, base + (uint64_t)n_steps + 1, s);
}
template <class Step>
int sharded_run(Transport& T, const gjx_smc_config* cfg, int n_state, const gjx_sharded_io* io, Step step, gjx_stream s) {
  if (!io) return GJX_ERR_INVALID;
  uint64_t received = 0;
  int rc = GJX_ERR_INVALID;
  const int world = T.world, rank = T.rank;
  const uint64_t tile = gjx_smc_tile();
  const bool adaptive = cfg && cfg->ess_threshold > 0.0f && cfg->ess_threshold < 1.0f;
  const bool peers = T.peers() != nullptr;
  bool ok = cfg && io->out_e && io->out_q && cfg->n_filters <= 1 && cfg->n_steps > 0 && world >= 1 && world <= 64 &&
            cfg->n_total % ((uint64_t)world * tile) == 0 && cfg->n_local == cfg->n_total / (uint64_t)world &&
            cfg->first_slot == (uint64_t)rank * cfg->n_local && !(io->shuffle == 0 && world > 1 && !io->ranges && !peers) &&
            !(adaptive && world > 1 && !peers && !io->stage);
  for (int i = 0; i < 2 && ok; ++i) {
    ok = io->pop[i].qw && io->pop[i].recs && io->pop[i].subs && (!adaptive || (io->pop[i].logw && io->pop[i].ess));
    for (int k = 0; k < n_state && ok; ++k) ok = io->pop[i].state[k] != nullptr;
  }
  if (ok) {
    rc = peers ? sharded_steps_peers(T, cfg, n_state, io, step, s) : sharded_steps(T, cfg, n_state, io, step, s, &received);
    // a rank that leaves the loop early would leave its peers waiting in the next collective: tell the transport
    if (rc != GJX_OK) T.abort();
  }
  if (io->received) *io->received = received;
  return rc;
}

inline int run_lgssm(Transport& T, const gjx_smc_config* cfg, const gjx_lgssm* model, const float* y, const gjx_sharded_io* io,
                     gjx_stream s) {
  if (!cfg || !model || !y || !io) return GJX_ERR_INVALID;
  auto step = [&](const gjx_smc_config* c, int t, const gjx_smc_pop* prev, const gjx_smc_pop* out, int32_t* pe, uint64_t* pq, int32_t* anc) {
    return gjx_smc_lgssm_step(c, model, t, y[t], prev, out, pe, pq, anc, s);
  };
  return sharded_run(T, cfg, 1, io, step, s);
}
inline int run_hmm(Transport& T, const gjx_smc_config* cfg, const gjx_hmm* model, const int32_t* y, const uint32_t* trans_alias,
                   const float* obs_logp, const gjx_sharded_io* io, gjx_stream s) {
  if (!cfg || !model || !y || !io || !trans_alias || !obs_logp) return GJX_ERR_INVALID;
  auto step = [&](const gjx_smc_config* c, int t, const gjx_smc_pop* prev, const gjx_smc_pop* out, int32_t* pe, uint64_t* pq, int32_t* anc) {
    return gjx_smc_hmm_step(c, model, t, y[t], prev, out, pe, pq, trans_alias, obs_logp, anc, s);
  };
  return sharded_run(T, cfg, 1, io, step, s);
}
inline int run_plan(Transport& T, const gjx_smc_config* cfg, gjx_smc_plan* plan, int n_state, int n_obs, const float* obs,
                    const gjx_sharded_io* io, gjx_stream s) {
  if (!cfg || !plan || !io || (n_obs > 0 && !obs) || n_state < 1 || n_state > GJX_SMC_MAX_STATE) return GJX_ERR_INVALID;
  auto step = [&](const gjx_smc_config* c, int t, const gjx_smc_pop* prev, const gjx_smc_pop* out, int32_t* pe, uint64_t* pq, int32_t* anc) {
    return gjx_smc_plan_step(c, plan, t, n_obs ? obs + (size_t)t * (size_t)n_obs : nullptr, prev, out, pe, pq, anc, s);
  };
  return sharded_run(T, cfg, n_state, io, step, s);
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
