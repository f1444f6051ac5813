// ORACLE build of the multi-rank entry points (gjx.h "multi-GPU") — test infrastructure, like the rest of oracle/.
// The driver (genjax-chi_amd/csrc/gjx_sharded.hpp) is orchestration only: it calls the public per-step entry points
// of whichever library it is compiled into, so here it runs the CPU restatement under `world` virtual ranks (threads
// of one process, host memory).  There is no RCCL on this side: gjx_comm_init_rccl reports GJX_ERR_UNSUPPORTED.
#include <cstdlib>
#include <new>

#include "../genjax-chi_amd/csrc/gjx_sharded.hpp"

extern "C" int gjo_smc_plan_dims(const gjx_smc_plan* p, int* n_state, int* n_obs);

namespace {
struct HostMem {
  int copy(void* dst, const void* src, size_t bytes, gjx_stream) {
    if (bytes) memmove(dst, src, bytes);
    return GJX_OK;
  }
  int sync(gjx_stream) { return GJX_OK; }
};
int host_copy(void* dst, const void* src, size_t bytes, gjx_stream) {
  if (bytes) memmove(dst, src, bytes);
  return GJX_OK;
}
}  // namespace

extern "C" {
int gjx_comm_unique_id(void* id_out) { (void)id_out; return GJX_ERR_UNSUPPORTED; }
int gjx_comm_init_rccl(const void* id, int rank, int world, gjx_comm** out) {
  (void)id; (void)rank; (void)world; (void)out;
  return GJX_ERR_UNSUPPORTED;
}
int gjx_comm_group_create(int world, gjx_comm_group** out) {
  if (!out || world < 1 || world > 16) return GJX_ERR_INVALID;
  *out = new (std::nothrow) gjx_comm_group(world);
  return *out ? GJX_OK : GJX_ERR_LAUNCH;
}
int gjx_comm_group_destroy(gjx_comm_group* g) { delete g; return GJX_OK; }
int gjx_comm_init_local(gjx_comm_group* g, int rank, gjx_comm** out) {
  if (!g || !out || rank < 0 || rank >= g->g.world) return GJX_ERR_INVALID;
  gjx_comm* c = new (std::nothrow) gjx_comm;
  if (!c) return GJX_ERR_LAUNCH;
  c->t = new (std::nothrow) gjx_sharded::LocalTransport<HostMem>(&g->g, rank);
  if (!c->t) { delete c; return GJX_ERR_LAUNCH; }
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
  // (the oracle's virtual ranks are threads whose steps run synchronously and concurrently: nothing queues behind anything)
  const int rc = gjx_sharded::comm_init_peers(peers, group ? &group->g : nullptr, wait_launch, false, &c->t);
  if (rc) {
    delete c;
    return rc;
  }
  *out = c;
  return GJX_OK;
}
int gjx_comm_destroy(gjx_comm* c) { delete c; return GJX_OK; }
int gjx_comm_rank(const gjx_comm* c) { return c && c->t ? c->t->rank : -1; }
int gjx_comm_world(const gjx_comm* c) { return c && c->t ? c->t->world : -1; }
int gjx_comm_lse_combine(gjx_comm* c, const uint64_t* records, int32_t n_batch, uint64_t* gathered, int32_t* out_e,
                         uint64_t* out_q, float* out_lse, gjx_stream s) {
  if (!c || !c->t) return GJX_ERR_INVALID;
  return gjx_sharded::lse_combine(*c->t, records, n_batch, gathered, out_e, out_q, out_lse, s, host_copy);
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
  int n_state = 0, n_obs = 0;
  if (!c || !c->t || !plan || gjo_smc_plan_dims(plan, &n_state, &n_obs)) return GJX_ERR_INVALID;
  return gjx_sharded::run_plan(*c->t, cfg, plan, n_state, n_obs, obs_host, io, s);
}
}
