// transport_local.hip — collectives that need no collective library: the in-process loopback group (`world` swarms of ONE
// process, one host thread each: virtual shards on one device — what the tests use on the one-GPU box — or several devices of a
// node without RCCL), a caller-supplied all-gather, and the measurement stand-in (ONE rank of `world` alone on a device).
#include "host_internal.h"

// ---- in-process collective: `world` swarms of one process, one host thread each -------------------------------------------------
struct mrs_loopback_group {
  int                      world = 0;
  std::atomic<int>         arrived{0};
  std::atomic<unsigned>    generation{0};
  std::vector<const void*> send;
  std::vector<hipEvent_t>  ev_ready, ev_copied;
  std::vector<int>         device;
  std::atomic<int>         failed{0};
  // rendezvous mode (mrs_loopback_group_set_rendezvous): no barrier — a rank stages its block, publishes the index of the collective
  // and only waits until every peer has published the same index (it cannot enqueue copies of data a peer has not enqueued yet);
  // staging buffers and events alternate between two sets, so a fast rank never waits for a slow one to have COPIED
  int                                     rendezvous = 0;
  std::unique_ptr<std::atomic<unsigned>[]> seq;           // seq[q] = collectives rank q has published
  std::vector<void*>                      stage[2];
  std::vector<size_t>                     stage_cap[2];
  std::vector<hipEvent_t>                 ev_ready2[2], ev_copied2[2];
  void barrier() {  // sense-reversing spin barrier (at most a handful of threads, all inside the same library call)
    const unsigned gen = generation.load(std::memory_order_acquire);
    if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == world) {
      arrived.store(0, std::memory_order_relaxed);
      generation.store(gen + 1, std::memory_order_release);
    } else {
      long spins = 0;
      const auto t0 = std::chrono::steady_clock::now();
      while (generation.load(std::memory_order_acquire) == gen) {
        if (++spins > 2000) std::this_thread::yield();
        // a peer that failed (or whose driver thread died) never arrives: give up instead of hanging the process
        if ((spins & 0xFFFF) == 0 && (failed.load() || std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))) {
          failed.store(1);
          return;
        }
      }
    }
  }
};

namespace mrs_host {
// all-gather among the swarms of a loopback group: every rank copies every rank's send buffer into its own receive buffer, device
// to device on its own stream; events order the copies behind the producers and the producers' next writes behind the copies
static int loopback_allgather_rendezvous(mrs_loopback_group* g, int rank, const void* send, void* recv, size_t bytes, hipStream_t st) {
  const size_t   r = (size_t)rank;
  const unsigned k = g->seq[r].load(std::memory_order_relaxed);  // index of this collective (only this thread writes seq[rank])
  const int      par = (int)(k & 1u);
  hipError_t     e = hipSuccess;
  // the staging buffer of this parity was last read by the peers in collective k - 2: their "copied" events were recorded before they
  // published k - 1, which this rank waited for in collective k - 1
  for (int q = 0; q < g->world && e == hipSuccess && k >= 2u; q++) e = hipStreamWaitEvent(st, g->ev_copied2[par][(size_t)q], 0);
  if (e == hipSuccess && bytes > g->stage_cap[par][r]) {
    e = hipStreamSynchronize(st);
    if (g->stage[par][r]) (void)hipFree(g->stage[par][r]);
    g->stage[par][r] = nullptr;
    if (e == hipSuccess) e = hipMalloc(&g->stage[par][r], bytes);
    g->stage_cap[par][r] = e == hipSuccess ? bytes : 0;
  }
  if (e == hipSuccess && bytes) e = hipMemcpyAsync(g->stage[par][r], send, bytes, hipMemcpyDeviceToDevice, st);
  if (e == hipSuccess) e = hipEventRecord(g->ev_ready2[par][r], st);
  if (e != hipSuccess) g->failed.store(1);
  g->seq[r].store(k + 1u, std::memory_order_release);
  const auto t0 = std::chrono::steady_clock::now();
  for (int q = 0; q < g->world; q++) {
    unsigned long spins = 0;
    while (g->seq[(size_t)q].load(std::memory_order_acquire) < k + 1u) {
      if (++spins > 2000) std::this_thread::yield();
      if ((spins & 0xFFFF) == 0 && (g->failed.load() || std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))) {
        g->failed.store(1);
        return fail(MRS_ERR_HIP, "loopback all-gather (rendezvous): a peer never arrived at collective " + std::to_string(k));
      }
    }
  }
  for (int q = 0; q < g->world && e == hipSuccess; q++) {
    e = hipStreamWaitEvent(st, g->ev_ready2[par][(size_t)q], 0);
    if (e == hipSuccess && bytes) e = hipMemcpyAsync((char*)recv + (size_t)q * bytes, g->stage[par][(size_t)q], bytes, hipMemcpyDeviceToDevice, st);
  }
  if (e == hipSuccess) e = hipEventRecord(g->ev_copied2[par][r], st);
  if (e != hipSuccess || g->failed.load()) {
    g->failed.store(1);
    return fail(MRS_ERR_HIP, std::string("loopback all-gather (rendezvous): ") + hipGetErrorString(e));
  }
  return MRS_OK;
}

int loopback_allgather(mrs_loopback_group* g, int rank, const void* send, void* recv, size_t bytes, hipStream_t st) {
  if (g->rendezvous) return loopback_allgather_rendezvous(g, rank, send, recv, bytes, st);
  const char* what = "record";
  hipError_t  e = hipGetLastError();  // (an error left behind by an earlier asynchronous launch belongs to that launch, not to this collective)
  if (e != hipSuccess) what = "an earlier launch on this thread";
  if (e == hipSuccess) e = hipEventRecord(g->ev_ready[(size_t)rank], st);
  g->send[(size_t)rank] = send;
  if (e != hipSuccess) g->failed.store(1);
  g->barrier();  // every rank has published its buffer and recorded "my send data is complete"
  for (int q = 0; q < g->world && e == hipSuccess; q++) {
    what = "wait for a peer's data";
    e = hipStreamWaitEvent(st, g->ev_ready[(size_t)q], 0);
    if (e == hipSuccess && bytes) {
      what = "copy";
      e = hipMemcpyAsync((char*)recv + (size_t)q * bytes, g->send[(size_t)q], bytes, hipMemcpyDeviceToDevice, st);
    }
  }
  if (e == hipSuccess) e = hipEventRecord(g->ev_copied[(size_t)rank], st);
  if (e != hipSuccess) g->failed.store(1);
  g->barrier();  // every rank has enqueued its copies
  for (int q = 0; q < g->world && e == hipSuccess; q++) e = hipStreamWaitEvent(st, g->ev_copied[(size_t)q], 0);  // nobody overwrites a buffer a peer still reads
  if (e != hipSuccess) return fail(MRS_ERR_HIP, std::string("loopback all-gather (") + what + ", " + std::to_string(bytes) + " bytes per rank): " + hipGetErrorString(e));
  if (g->failed.load()) return fail(MRS_ERR_HIP, "loopback all-gather: a peer of the group has failed");
  return MRS_OK;
}

// The collective of the measurement stand-in: ONE kernel that takes `standin_delay_us` of stream time (a collective's latency) and
// leaves the rank's own block in its own place and in the places of its two neighbours in the slab order — records (recognised by
// their size) moved one slab width to either side, slot maps and export blocks as they are: the neighbours are periodic images of
// this rank, so the export sets mirror each other as they do between real neighbours (positions of foreign partners are the
// images' only at a search; between searches they read as far away — fine for a time measurement, meaningless as a simulation).
int standin_allgather(mrs_swarm* s, const void* send, void* recv, size_t bytes) {
  // what travels is told by where it comes from: position records, halo entries of a search, slot maps (with the search box at the tail)
  const int       kind = send == (const void*)s->comm_send ? 1 : (s->cwork && send == mrs_collide_halo_send(s->cwork) ? 2 : (send == (const void*)s->x_map_send ? 3 : 0));
  const long long aux  = kind == 3 ? (long long)(map_boxw(s) / 4) : 0;
  if (bytes % 16 != 0) {  // (a block that is no whole number of 16-byte units — none of the library's own collectives any more) plain copies
    HIPCHK(hipMemsetAsync(recv, 0, bytes * (size_t)s->comm_world, s->cstream));
    for (int d = -1; d <= 1; d++)
      if (s->comm_rank + d >= 0 && s->comm_rank + d < s->comm_world)
        HIPCHK(hipMemcpyAsync((char*)recv + (size_t)(s->comm_rank + d) * bytes, send, bytes, hipMemcpyDeviceToDevice, s->cstream));
    return MRS_OK;
  }
  // MRS_STANDIN_GBPS (default 0: off): the collective also takes the time its bytes need on the links — (world - 1) blocks received
  // at that many GB/s — on top of the fixed latency; without it the 42 MB of a search tick's record gather cost as much as 54 KB
  const double gbps    = s->standin_gbps;  // (read when the stand-in communicator is bound)
  const double wire_us = gbps > 0.0 ? (double)bytes * (double)(s->comm_world - 1) / (gbps * 1e3) : 0.0;
  HIPCHK(mrs_launch_standin_gather(send, recv, bytes, s->comm_rank, s->comm_world, s->standin_delay_us + wire_us, kind, aux, s->standin_width, s->cstream));
  return MRS_OK;
}
}  // namespace mrs_host

extern "C" {

int mrs_swarm_comm_init_custom(mrs_swarm_t* s, int32_t world, int32_t rank, int64_t n_total, mrs_allgather_fn fn, void* user) {
  MRS_ENTER(s);
  if (!s || !fn) return fail(MRS_ERR_ARG, "null argument");
  int rc = comm_setup(s, world, rank, n_total);
  if (rc) return rc;
  HIPCHK(hipSetDevice(s->device));
  s->comm_fn   = fn;
  s->comm_user = user;
  return comm_buffers(s, world, rank, n_total);
}

int mrs_swarm_comm_init_standin(mrs_swarm_t* s, int32_t world, int32_t rank, int64_t n_total, double collective_latency_us, double slab_width) {
  MRS_ENTER(s);
  if (!s || !(collective_latency_us >= 0) || !(slab_width > 0)) return fail(MRS_ERR_ARG, "bad stand-in arguments");
  int rc = comm_setup(s, world, rank, n_total);
  if (rc) return rc;
  HIPCHK(hipSetDevice(s->device));
  s->comm_standin     = true;
  s->standin_delay_us = collective_latency_us;
  s->standin_width    = slab_width;
  s->standin_gbps     = getenv("MRS_STANDIN_GBPS") ? atof(getenv("MRS_STANDIN_GBPS")) : 0.0;
  return comm_buffers(s, world, rank, n_total);
}

int mrs_loopback_group_create(int32_t world, mrs_loopback_group_t** out) {
  if (!out || world < 1 || world > 64) return fail(MRS_ERR_ARG, "bad loopback group size");
  mrs_loopback_group* g = new mrs_loopback_group();
  g->world = world;
  g->send.assign((size_t)world, nullptr);
  g->ev_ready.assign((size_t)world, nullptr);
  g->ev_copied.assign((size_t)world, nullptr);
  g->device.assign((size_t)world, -1);
  g->seq.reset(new std::atomic<unsigned>[(size_t)world]);
  for (int q = 0; q < world; q++) g->seq[(size_t)q].store(0u);
  for (int par = 0; par < 2; par++) {
    g->stage[par].assign((size_t)world, nullptr);
    g->stage_cap[par].assign((size_t)world, 0);
    g->ev_ready2[par].assign((size_t)world, nullptr);
    g->ev_copied2[par].assign((size_t)world, nullptr);
  }
  *out = g;
  return MRS_OK;
}

int mrs_loopback_group_set_rendezvous(mrs_loopback_group_t* g, int32_t on) {
  if (!g) return fail(MRS_ERR_ARG, "null group");
  for (int q = 0; q < g->world; q++)
    if (g->seq[(size_t)q].load() != 0u) return fail(MRS_ERR_ARG, "the mode of a loopback group is chosen before its first collective");
  g->rendezvous = on ? 1 : 0;
  return MRS_OK;
}

int mrs_loopback_group_destroy(mrs_loopback_group_t* g) {
  if (!g) return MRS_OK;
  for (int q = 0; q < g->world; q++) {
    if (g->device[(size_t)q] >= 0) (void)hipSetDevice(g->device[(size_t)q]);
    if (g->ev_ready[(size_t)q]) (void)hipEventDestroy(g->ev_ready[(size_t)q]);
    if (g->ev_copied[(size_t)q]) (void)hipEventDestroy(g->ev_copied[(size_t)q]);
    for (int par = 0; par < 2; par++) {
      if (g->ev_ready2[par][(size_t)q]) (void)hipEventDestroy(g->ev_ready2[par][(size_t)q]);
      if (g->ev_copied2[par][(size_t)q]) (void)hipEventDestroy(g->ev_copied2[par][(size_t)q]);
      if (g->stage[par][(size_t)q]) (void)hipFree(g->stage[par][(size_t)q]);
    }
  }
  delete g;
  return MRS_OK;
}

int mrs_swarm_comm_init_loopback(mrs_swarm_t* s, mrs_loopback_group_t* g, int32_t rank, int64_t n_total) {
  MRS_ENTER(s);
  if (!s || !g) return fail(MRS_ERR_ARG, "null argument");
  int rc = comm_setup(s, g->world, rank, n_total);
  if (rc) return rc;
  if (g->ev_ready[(size_t)rank]) return fail(MRS_ERR_ARG, "this rank of the loopback group is taken");
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipEventCreateWithFlags(&g->ev_ready[(size_t)rank], hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&g->ev_copied[(size_t)rank], hipEventDisableTiming));
  for (int par = 0; par < 2; par++) {
    HIPCHK(hipEventCreateWithFlags(&g->ev_ready2[par][(size_t)rank], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&g->ev_copied2[par][(size_t)rank], hipEventDisableTiming));
    HIPCHK(hipEventRecord(g->ev_copied2[par][(size_t)rank], s->stream));  // (an event that was never recorded must not be waited for)
  }
  g->device[(size_t)rank] = s->device;
  s->comm_group           = g;
  return comm_buffers(s, g->world, rank, n_total);
}

int mrs_debug_stream_delay(void* stream, double microseconds) {
  if (!(microseconds >= 0) || microseconds > 1e6) return fail(MRS_ERR_ARG, "bad delay");
  HIPCHK(mrs_launch_stream_delay((hipStream_t)stream, microseconds));
  return MRS_OK;
}

}  // extern "C"
