// tick_sharded.hip — timerMain on every rank of a sharded swarm (mrs_swarm_tick_sharded_n): makeStep, then handleCollisions over
// ALL UAVs of all ranks (src/multirotor_simulator.cpp:211-217, 295-359).  Communicator bookkeeping, the search path of the
// export-set exchange, and its ticks in the serial and the split form.  The protocol and its proof obligations: DESIGN §5; its
// host-side decisions as pure functions: sharded_protocol.h.
#include "host_internal.h"
#include "sharded_protocol.h"

namespace mrs_host {
int comm_allgather(mrs_swarm* s, const void* send, void* recv, size_t bytes) {
  if (s->comm_peer) return peer_allgather(s, send, recv, bytes);
  if (s->comm_standin) return standin_allgather(s, send, recv, bytes);
  if (s->rccl_comm) return rccl_check(g_rccl.AllGather(send, recv, bytes, /*ncclInt8*/ 0, s->rccl_comm, s->cstream), "ncclAllGather");
  if (s->comm_group) return loopback_allgather(s->comm_group, s->comm_rank, send, recv, bytes, s->cstream);
  if (s->comm_fn) {
    const int rc = s->comm_fn(s->comm_user, send, recv, (uint64_t)bytes, (void*)s->cstream);
    return rc == 0 ? MRS_OK : fail(MRS_ERR_HIP, "the caller's all-gather failed with code " + std::to_string(rc));
  }
  return fail(MRS_ERR_ARG, "no communicator");
}

int comm_setup(mrs_swarm* s, int world, int rank, int64_t n_total) {
  if (world < 1 || rank < 0 || rank >= world || n_total < s->n) return fail(MRS_ERR_ARG, "bad communicator shape");
  if (s->comm_world > 0) return fail(MRS_ERR_ARG, "communicator already initialised");
  const int64_t base = n_total / world, rem = n_total % world;
  const int64_t mine = base + (rank < rem ? 1 : 0);  // equal-count shards, sizes differ by at most one
  if (mine != s->n) return fail(MRS_ERR_ARG, "this swarm does not hold the shard of its rank (n_total / world UAVs, the first n_total % world ranks one more)");
  // (the position records of a shard are addressed through a buffer descriptor with 32-bit byte offsets: step_device.inc store_pos_sc1)
  if ((long long)s->n >= (1ll << 27)) return fail(MRS_ERR_ARG, "a shard of a sharded swarm holds at most 2^27 - 1 UAVs (use more ranks)");
  return MRS_OK;
}

int comm_buffers(mrs_swarm* s, int world, int rank, int64_t n_total) {
  s->comm_world   = world;
  s->comm_rank    = rank;
  s->comm_n_total = n_total;
  s->comm_n_max   = (n_total + world - 1) / world;
  if (s->comm_n_max < 1) s->comm_n_max = 1;
  s->x_ok         = false;
  s->halo_ok      = false;
  s->halo_cap     = 0;
  s->halo_backoff = 0;
  s->halo_backoff_len = 16;
  if (const char* e = getenv("MRS_SEARCH_HALO")) s->halo_enabled = atoi(e) != 0;
  s->halo_trace = getenv("MRS_HALO_TRACE") && atoi(getenv("MRS_HALO_TRACE")) != 0;
  s->x_last_overflow.assign((size_t)world, 0u);
  if (const char* e = getenv("MRS_EXCHANGE")) s->exchange = atoi(e) == 1 ? MRS_EXCHANGE_FULL_GATHER : MRS_EXCHANGE_EXPORT_SETS;
  HIPCHK(hipMalloc(&s->comm_send, sizeof(PosRecord) * (size_t)s->comm_n_max));
  HIPCHK(hipMalloc(&s->comm_recv, sizeof(PosRecord) * (size_t)s->comm_n_max * (size_t)world));
  HIPCHK(hipMalloc(&s->x_map_send, sizeof(uint32_t) * (size_t)map_stride(s)));
  HIPCHK(hipMalloc(&s->x_map_recv, sizeof(uint32_t) * (size_t)map_stride(s) * (size_t)world));
  HIPCHK(hipMemsetAsync(s->x_map_recv, 0, sizeof(uint32_t) * (size_t)map_stride(s) * (size_t)world, s->stream));  // (the stand-in collective leaves absent ranks' maps alone)
  HIPCHK(hipMemsetAsync(s->comm_send, 0xFF, sizeof(PosRecord) * (size_t)s->comm_n_max, s->stream));  // NaN padding records never collide
  HIPCHK(hipMemsetAsync(s->x_map_send, 0xFF, sizeof(uint32_t) * (size_t)map_stride(s), s->stream));      // (the box at its tail: NaN bounds until a search writes it — for good on a rank without UAVs)
  mrs_collide_set_box_out(&s->cwork, reinterpret_cast<double*>(s->x_map_send + map_boxw(s)));
  return MRS_OK;
}
}  // namespace mrs_host

extern "C" {

int mrs_swarm_get_split_stats(mrs_swarm_t* s, int64_t* split_ticks, int64_t* boundary_blocks) {
  MRS_LOCK(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (split_ticks) *split_ticks = s->x_split_ticks;
  if (boundary_blocks) *boundary_blocks = (int64_t)s->x_nbnd;
  return MRS_OK;
}

int mrs_swarm_get_search_stats(mrs_swarm_t* s, int64_t* searches, int64_t* halo_searches, int64_t* halo_repeats, int64_t* halo_capacity) {
  MRS_LOCK(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (searches) *searches = s->x_searches;
  if (halo_searches) *halo_searches = s->x_halo_searches;
  if (halo_repeats) *halo_repeats = s->x_halo_repeats;
  if (halo_capacity) *halo_capacity = halo_next(s) ? s->halo_cap : 0;
  return MRS_OK;
}

int mrs_swarm_debug_chaos(mrs_swarm_t* s, int32_t max_sleep_us, uint64_t seed) {
  MRS_LOCK(s);
  if (!s || max_sleep_us < 0) return fail(MRS_ERR_ARG, "bad chaos arguments");
  s->chaos_max_us = max_sleep_us;
  s->chaos_state  = seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
  if (s->chaos_state == 0) s->chaos_state = 1;
  return MRS_OK;
}

int mrs_swarm_set_exchange(mrs_swarm_t* s, int32_t exchange) {
  MRS_ENTER(s);
  if (!s || (exchange != MRS_EXCHANGE_FULL_GATHER && exchange != MRS_EXCHANGE_EXPORT_SETS)) return fail(MRS_ERR_ARG, "bad exchange");
  if (exchange != s->exchange) {
    s->x_ok = false;
    s->halo_ok = false;
    mrs_collide_invalidate_gathered(s->cwork);  // the two exchanges keep their neighbour lists in different forms
  }
  s->exchange = exchange;
  return MRS_OK;
}

int mrs_swarm_comm_destroy(mrs_swarm_t* s) {
  MRS_ENTER(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (s->comm_world == 0) {
    if (s->peer_window) peer_release(s);  // a window that never became a communicator
    return MRS_OK;
  }
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipStreamSynchronize(s->stream));
  int rc = MRS_OK;
  if (s->rccl_comm) rc = rccl_check(g_rccl.CommDestroy(s->rccl_comm), "ncclCommDestroy");
  s->rccl_comm  = nullptr;
  s->comm_fn    = nullptr;
  s->comm_user  = nullptr;
  s->comm_group = nullptr;
  s->comm_standin = false;
  peer_release(s);
  s->comm_world = 0;
  s->x_ok       = false;
  s->halo_ok    = false;
  mrs_collide_invalidate_gathered(s->cwork);
  if (s->comm_send) (void)hipFree(s->comm_send);
  if (s->comm_recv) (void)hipFree(s->comm_recv);
  if (s->cwork) mrs_collide_set_box_out(&s->cwork, nullptr);
  if (s->x_map_send) (void)hipFree(s->x_map_send);
  if (s->x_map_recv) (void)hipFree(s->x_map_recv);
  s->comm_send = s->comm_recv = nullptr;
  s->x_map_send = s->x_map_recv = nullptr;
  return rc;
}

int mrs_swarm_comm_info(mrs_swarm_t* s, mrs_comm_info_t* out) {
  MRS_ENTER(s);
  if (!s || !out) return fail(MRS_ERR_ARG, "null argument");
  memset(out, 0, sizeof *out);
  if (s->comm_world == 0) return MRS_OK;
  out->world    = s->comm_world;
  out->rank     = s->comm_rank;
  out->n_total  = s->comm_n_total;
  out->exchange = s->exchange;
  const int64_t full = (int64_t)sizeof(PosRecord) * s->comm_n_max;
  if (s->exchange == MRS_EXCHANGE_EXPORT_SETS) {
    out->export_capacity   = mrs_collide_export_capacity(s->cwork);
    out->export_count      = s->x_export_count;
    out->bytes_per_tick    = (int64_t)sizeof(Pos4) * (1 + out->export_capacity);
    // (a search on a halo exchange sends the entries its neighbours can list instead of all records)
    out->bytes_per_rebuild = (halo_next(s) ? (int64_t)sizeof(HaloEntry) * (1 + s->halo_cap) : full) + (int64_t)sizeof(uint32_t) * map_stride(s);
  } else {
    out->bytes_per_tick = out->bytes_per_rebuild = full;
  }
  out->ticks      = s->x_ticks;
  out->searches   = s->x_searches;
  out->noop_ticks = s->x_noop_ticks;
  if (s->rccl_comm && g_rccl.CommCount) {
    int c = 0;
    int rc = rccl_check(g_rccl.CommCount(s->rccl_comm, &c), "ncclCommCount");
    if (rc) return rc;
    out->rccl_ranks = c;
  }
  return MRS_OK;
}

// UAVs sorted by x (ties: by public index), cut into equal-count slabs by the caller
int mrs_slab_partition(const double* pos_xyz, int64_t n_total, int32_t world, int64_t* order) {
  if (!pos_xyz || !order || n_total < 0 || world < 1) return fail(MRS_ERR_ARG, "bad partition arguments");
  for (int64_t k = 0; k < n_total; k++) order[k] = k;
  std::stable_sort(order, order + n_total, [&](int64_t a, int64_t b) {
    const double xa = pos_xyz[3 * a], xb = pos_xyz[3 * b];
    if (xa != xa || xb != xb) return (xa == xa) && (xb != xb);  // NaN positions last
    return xa < xb;
  });
  return MRS_OK;
}

// Morton order of the neighbour-list cells (non-finite positions last, ties by index)
int mrs_cell_order(const double* pos_xyz, int64_t n_total, double cell, int64_t* order) {
  if (!pos_xyz || !order || n_total < 0) return fail(MRS_ERR_ARG, "bad order arguments");
  if (!(cell > 0)) cell = 2.25;
  double lo[3] = {0, 0, 0};
  bool   any   = false;
  for (int64_t k = 0; k < n_total; k++) {
    const double* p = pos_xyz + 3 * k;
    if (!(std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2]))) continue;
    for (int a = 0; a < 3; a++) lo[a] = any ? std::min(lo[a], p[a]) : p[a];
    any = true;
  }
  std::vector<uint64_t> key((size_t)n_total);
  for (int64_t k = 0; k < n_total; k++) {
    const double* p = pos_xyz + 3 * k;
    uint64_t      m = ~0ull;
    if (std::isfinite(p[0]) && std::isfinite(p[1]) && std::isfinite(p[2])) {
      m = 0;
      for (int a = 0; a < 3; a++) {
        double c = std::floor((p[a] - lo[a]) / cell);
        if (c > 2097151.0) c = 2097151.0;  // 21 bits per axis
        const uint64_t ci = (uint64_t)c;
        for (int b = 0; b < 21; b++) m |= ((ci >> b) & 1ull) << (3 * b + a);
      }
    }
    key[(size_t)k] = m;
  }
  for (int64_t k = 0; k < n_total; k++) order[k] = k;
  std::stable_sort(order, order + n_total, [&](int64_t a, int64_t b) { return key[(size_t)a] < key[(size_t)b]; });
  return MRS_OK;
}

}  // extern "C"

// ---- sharded ticks ----
namespace mrs_host {

// every tick gathers all records (MRS_EXCHANGE_FULL_GATHER, and the fallback of the export-set exchange when some UAV has more
// neighbours than its list holds): step, pack, ONE all-gather of the 48-B records, collision pass against the gathered records
int full_gather_ticks(mrs_swarm* s, double dt, int n_ticks, const mrs_swarm::Collide& c) {
  const int64_t n_rec = s->comm_n_max * s->comm_world;
  int           rc;
  s->x_ok    = false;
  s->halo_ok = false;  // (the record table is about to hold other ticks' records; the next search of the export-set exchange is a full one)
  s->p_valid = false;
  s->fk_ok   = false;
  for (int k = 0; k < n_ticks; k++) {
    if (s->n > 0 && (rc = launch_step(s, dt, 1))) return rc;
    if (s->n > 0) HIPCHK(mrs_launch_pack_positions(s->view(), s->comm_send, s->stream));
    if ((rc = comm_allgather(s, s->comm_send, s->comm_recv, sizeof(PosRecord) * (size_t)s->comm_n_max))) return rc;
    s->x_ticks++;
    if (s->n == 0) continue;
    s->fext_active = true;
    s->collision_ticks++;
    s->nbr_dirty = true;
    if (s->use_lists)
      HIPCHK(mrs_collide_run_lists_gathered(s->view(), &s->cwork, s->comm_recv, n_rec, (int64_t)s->comm_rank * s->comm_n_max, c.crash, c.rebounce, 0, s->stream));
    else
      HIPCHK(mrs_collide_run(s->view(), &s->cwork, s->comm_recv, n_rec, (int64_t)s->comm_rank * s->comm_n_max, c.crash, c.rebounce, 0, s->stream));
  }
  return MRS_OK;
}

// The tick after the most recent step on the SEARCH path of the export-set exchange: gather all records, search (which evaluates
// this tick's handleCollisions), then derive the export sets and rewrite the lists.  Collective.  Returns 1 when the lists came out
// incomplete (a UAV with more neighbours than its list holds, on any rank): the caller stays on the full exchange for a while.
// first half: everything up to the point where the host needs numbers of its own — ENQUEUED only (no host wait), so that a search
// that is certain can follow the last launches of a segment in stream order and the segment's one synchronisation serves both
// dt: the step of the ticks that follow (the search also runs the displacement bound over MRS_PRED_HORIZON of them, split protocol only)
// The exchange of the search is a HALO exchange (collide.hip mrs_collide_halo_*) when the last search left every rank's box in the slot
// maps: each rank sends the records some other rank can list, not all of them.  Whether that was enough is known with the head words
// (every rank reads the same halo headers): export_search_finish repeats the search on the full exchange if not (allow_halo = false).
int export_search_enqueue(mrs_swarm* s, const mrs_swarm::Collide& c, double dt, bool allow_halo) {
  const int     world = s->comm_world, rank = s->comm_rank;
  const int64_t n_max = s->comm_n_max, n_rec = n_max * world, stride = map_stride(s);
  int           rc;
  s->x_ok = false;
  s->x_searches++;
  s->halo_pass = allow_halo && halo_next(s);
  s->fext_active = true;
  s->nbr_dirty   = true;  // (a later single-GPU tick starts from a search of its own)
  if (s->halo_pass) {
    s->x_halo_searches++;
    HIPCHK(mrs_collide_halo_prepare(&s->cwork, world, s->halo_cap, s->stream));
    // (a rank whose tables are not what a search of this exchange left — ticks of another kind in between, on this rank only — says so in its header)
    const int ready = s->n == 0 || mrs_collide_halo_ready(s->cwork, n_rec);
    // (one pass over the own UAVs: records into the table, halo entries, partial boxes, header)
    HIPCHK(mrs_collide_halo_select(s->view(), s->cwork, s->comm_recv, n_max, rank, world, s->x_map_recv, stride, (int)map_boxw(s), ready ? 0 : 1, s->stream));
    if ((rc = comm_allgather(s, mrs_collide_halo_send(s->cwork), mrs_collide_halo_recv(s->cwork), sizeof(HaloEntry) * (size_t)(s->halo_cap + 1)))) return rc;
    if (s->n > 0 && ready) HIPCHK(mrs_collide_run_lists_halo(s->view(), &s->cwork, s->comm_recv, n_rec, n_max, rank, world, c.crash, c.rebounce, s->stream));
  } else {
    if (s->n > 0) HIPCHK(mrs_launch_pack_positions(s->view(), s->comm_send, s->stream));
    if ((rc = comm_allgather(s, s->comm_send, s->comm_recv, sizeof(PosRecord) * (size_t)n_max))) return rc;
    if (s->n > 0)
      HIPCHK(mrs_collide_run_lists_gathered(s->view(), &s->cwork, s->comm_recv, n_rec, (int64_t)rank * n_max, c.crash, c.rebounce, /*force=*/1, s->stream));
  }
  const long long cap = mrs_collide_export_capacity(s->cwork);
  HIPCHK(mrs_collide_export_prepare(s->view(), &s->cwork, world, cap > 0 ? cap : 64, /*zero=*/0, s->stream));  // (zeroed by the marking launches)
  // (the map's reset stops short of its tail: the search has left this rank's box there — what the next search's halos are chosen by)
  HIPCHK(mrs_collide_export_mark(s->view(), s->cwork, n_max, map_boxw(s), rank, s->x_map_send, s->shard_split ? (double)MRS_PRED_HORIZON * dt : -1.0, c.rebounce,
                                 s->stream));
  if ((rc = comm_allgather(s, s->x_map_send, s->x_map_recv, sizeof(uint32_t) * (size_t)stride))) return rc;
  // the heads of all ranks' maps: export count, lanes over the list capacity so far — the same numbers on every rank
  const uint32_t* heads = nullptr;  // (pinned host words, written by one small launch; valid once the stream has been synchronised)
  HIPCHK(mrs_collide_heads_to_host(s->cwork, s->x_map_recv, stride, world, s->halo_pass ? 1 : 0, &heads, s->stream));
  return MRS_OK;
}

// second half, behind a synchronisation of the stream: capacities, then the lists go into export form
// may_leave: some UAV of some rank may leave its skin within MRS_PRED_HORIZON steps of the state the search ran on (or the bound was
// not evaluated): the ticks that follow take the serial form until the launches' own announcements cover the horizon
int export_search_finish(mrs_swarm* s, const mrs_swarm::Collide& c, double dt, int* incomplete, bool* may_leave) {
  const int       world = s->comm_world, rank = s->comm_rank;
  const int64_t   n_max = s->comm_n_max;
  const uint32_t* heads = mrs_collide_host_heads(s->cwork);
  *incomplete = 0;
  if (!heads) return fail(MRS_ERR_HIP, "export-set search: no head words");
  if (s->halo_pass) {
    // the halo headers, the same on every rank: did every rank's halo hold what it had to (nobody outside the margin, nothing cut off)?
    const uint32_t wanted = heads[2 * world + 2], flags = heads[2 * world + 3];
    s->halo_pass = false;
    const int64_t cap_max = halo_cap_max(s);
    int64_t       next    = (((int64_t)wanted + (int64_t)wanted / 4 + 64 + 63) / 64) * 64;  // headroom: the sets drift from search to search
    if (next > cap_max) next = cap_max;
    if ((int64_t)wanted > cap_max) {  // most records would travel anyway: the next searches gather them all — 16 of them, then twice as many each time it happens again
      s->halo_backoff     = s->halo_backoff_len;
      s->halo_backoff_len = s->halo_backoff_len < 1024 ? 2 * s->halo_backoff_len : 1024;
    } else if (!flags) {
      s->halo_backoff_len = 16;
    }
    if (s->halo_trace)
      fprintf(stderr, "[mrs halo] rank %d search %lld: most entries wanted by a rank %u, block capacity %lld (next %lld, at most %lld), flags %u%s\n", rank,
              (long long)s->x_searches, wanted, (long long)s->halo_cap, (long long)next, (long long)cap_max, flags, flags ? " -> repeated on all records" : "");
    s->halo_cap = next;
    if (flags) {  // the lists of this pass may miss partners: the same search again, on all records (forces are set, not added; crashes only grow)
      s->x_halo_repeats++;
      int rc = export_search_enqueue(s, c, dt, /*allow_halo=*/false);
      if (rc) return rc;
      HIPCHK(hipStreamSynchronize(s->stream));
    }
  }
  mrs_collide_host_words_reset(s->cwork);  // launch indices restart at 1: the host mirrors of the old segment's words are void
  const long long cap = mrs_collide_export_capacity(s->cwork);
  s->x_nbnd = s->n > 0 ? heads[2 * world] : 0u;
  s->x_nl1  = s->n > 0 ? heads[2 * world + 1] : 0u;
  *may_leave = !s->shard_split;
  long long need = 0;
  for (int q = 0; q < world; q++) {
    if (heads[(size_t)q * 2] & 0x80000000u) *may_leave = true;
    if ((long long)(heads[(size_t)q * 2] & 0x7FFFFFFFu) > need) need = heads[(size_t)q * 2] & 0x7FFFFFFFu;
    if (heads[(size_t)q * 2 + 1] != s->x_last_overflow[(size_t)q]) *incomplete = 1;
    s->x_last_overflow[(size_t)q] = heads[(size_t)q * 2 + 1];
  }
  s->x_export_count = heads[(size_t)rank * 2] & 0x7FFFFFFFu;
  if (*incomplete) {
    mrs_collide_invalidate_gathered(s->cwork);
    return MRS_OK;
  }
  if (need > cap || cap == 0) {  // grow with headroom: the sets change from search to search
    long long ncap = ((need + need / 2 + 64 + 63) / 64) * 64;
    HIPCHK(mrs_collide_export_prepare(s->view(), &s->cwork, world, ncap, /*zero=*/1, s->stream));
  }
  HIPCHK(mrs_collide_export_translate(s->view(), s->cwork, n_max, map_stride(s), rank, s->x_map_recv, s->comm_recv, s->stream));
  // (the lists are in export form now; collide.hip remembers that, and the full exchange would start with a search of its own)
  s->x_ok = true;
  s->halo_ok = true;  // (every rank's box of this search sits in x_map_recv)
  // the first halo search sends blocks as large as a full gather's (no more bytes than the search it replaces, and no overflow short
  // of "most records wanted", which is the back-off's case); it reports what is needed and the capacity follows
  if (s->halo_cap < 1) s->halo_cap = halo_cap_max(s);
  if (s->halo_backoff > 0) s->halo_backoff--;
  s->tau  = 0;
  return MRS_OK;
}

int export_search(mrs_swarm* s, const mrs_swarm::Collide& c, double dt, int* incomplete, bool* may_leave) {
  int rc = export_search_enqueue(s, c, dt, true);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(s->stream));  // (a stamp in pinned memory polled by the host instead: measured, 35.1-35.5 against 35.2 us per tick — no gain, removed)
  return export_search_finish(s, c, dt, incomplete, may_leave);
}

int launch_fused_export(mrs_swarm* s, double dt, const mrs_swarm::Collide& eval) {
  if (s->n > 0) {
    CollDev  cd;
    SwarmDev v = s->view();
    HIPCHK(mrs_collide_export_dev(&v, s->cwork, (int64_t)s->comm_rank * s->comm_n_max, s->tau + 1, eval.on ? 1 : 0, eval.crash, eval.rebounce, &cd));
    mrs_collide_export_part(&cd, MRS_PART_FULL, 0u, dt, s->shard_split ? 1 : 0);  // (MRS_SHARD_SPLIT=0: round 2's protocol, nothing announced)
    s->region_launches++;
    const int variant = s->n_cascade > 0 ? 0 : 1;
    if (s->arith == MRS_ARITH_FAST)
      HIPCHK(mrs_launch_step_coll_fast(v, cd, dt, variant, 0, s->stream));
    else
      HIPCHK(mrs_launch_step_coll_literal(v, cd, dt, variant, 0, s->stream));
    mrs_collide_fused_advance(s->cwork);
  } else {
    HIPCHK(mrs_collide_export_fold_stall(s->cwork, s->tau + 1, s->stream));  // a rank without UAVs still watches the headers
  }
  s->tau++;
  static const int exp_skip = getenv("MRS_EXP_SPLIT_SKIP") ? atoi(getenv("MRS_EXP_SPLIT_SKIP")) : 0;  // (measurement only, see launch_split_export)
  if (exp_skip & 2) return MRS_OK;
  const size_t bytes = sizeof(Pos4) * (size_t)(mrs_collide_export_capacity(s->cwork) + 1);
  return comm_allgather(s, mrs_collide_export_send(s->cwork), mrs_collide_export_recv(s->cwork), bytes);
}

// One tick of the export-set exchange in the SPLIT form: the boundary launch (the blocks that hold a UAV with a foreign partner; it
// evaluates, steps, writes this rank's export block) and the collective on `stream`, the interior launch on `stream2`.  The two
// chains meet inside the kernels (per-block epoch words: step_device.inc), never on the host and never through an event, so the
// interior launch of tick t+1 runs beside the collective of tick t.  `split_base`: launch index behind which this run of split
// ticks started (mrs_collide_handoff_init).
int launch_split_export(mrs_swarm* s, double dt, const mrs_swarm::Collide& eval, unsigned split_base) {
  CollDev  cd, part;
  SwarmDev v = s->view();
  HIPCHK(mrs_collide_export_dev(&v, s->cwork, (int64_t)s->comm_rank * s->comm_n_max, s->tau + 1, eval.on ? 1 : 0, eval.crash, eval.rebounce, &cd));
  s->region_launches++;
  const int      variant = s->n_cascade > 0 ? 0 : 1, bound_ok = 1;
  const unsigned grid_b  = s->x_nbnd > 0 ? s->x_nbnd : 1u;
  auto launch = [&](const CollDev& c, int grid, hipStream_t st) {
    return s->arith == MRS_ARITH_FAST ? mrs_launch_step_coll_fast(v, c, dt, variant, grid, st) : mrs_launch_step_coll_literal(v, c, dt, variant, grid, st);
  };
  // measurement only (wrong results; tools/sharded_interior_alone.py with a -DMRS_WAIT_TICKS=0 build): one of the three parts of a
  // split tick left out, to see what each costs beside the others
  static const int exp_skip = getenv("MRS_EXP_SPLIT_SKIP") ? atoi(getenv("MRS_EXP_SPLIT_SKIP")) : 0;  // bit 0 boundary launch, 1 collective, 2 interior launch
  part = cd;
  mrs_collide_export_part(&part, MRS_PART_BOUNDARY, s->x_nbnd, dt, bound_ok);
  if (!(exp_skip & 1)) HIPCHK(launch(part, (int)grid_b, s->cstream));
  const size_t bytes = sizeof(Pos4) * (size_t)(mrs_collide_export_capacity(s->cwork) + 1);
  int rc = (exp_skip & 2) ? MRS_OK : comm_allgather(s, mrs_collide_export_send(s->cwork), mrs_collide_export_recv(s->cwork), bytes);
  if (rc) return rc;
  part = cd;
  mrs_collide_export_part(&part, MRS_PART_INTERIOR, s->x_nbnd, dt, bound_ok);
  // (tuning: MRS_INTERIOR_L1_FIRST=1 — the interior launch's first blocks are its layer-1 blocks, the ones the next boundary launch waits for)
  static const bool l1_first = getenv("MRS_INTERIOR_L1_FIRST") && atoi(getenv("MRS_INTERIOR_L1_FIRST")) != 0;
  part.n_l1 = l1_first ? s->x_nl1 : 0u;
  if (!(exp_skip & 4)) HIPCHK(launch(part, part.n_l1 ? (int)(part.n_l1 + (unsigned)((s->n + 63) / 64)) : 0, s->stream_i ? s->stream_i : s->stream2));
  mrs_collide_fused_advance(s->cwork);
  s->tau++;
  s->x_split_ticks++;
  return MRS_OK;
}

// Ticks of the export-set exchange.  All ranks must issue the same launches and collectives in the same order, yet nobody may wait
// for anybody on the host.  What keeps them in step: every decision is taken from words that reach all ranks with the positions
// themselves (headers of the export collective, folded by each fused launch into pinned host words), at a launch index that is a
// function of those words alone:
//   * warning word W (tick in which some UAV of some rank had used 75 % of its skin): the search is done before launch W + D;
//   * stall word T (some UAV left its skin during step T; launches > T are no-ops everywhere): the segment ends with launch T + L + 1;
//   * L = launches a host may run ahead of its device (progress word), D = L + 3: a host deciding on launch W + D, or on T + L + 1,
//     has provably seen W, or T (the launches that report them have completed on its device by then).
// A segment ends with fold + synchronise (the only host wait), then the search where one is due.
int export_ticks(mrs_swarm* s, double dt, int n_ticks, const mrs_swarm::Collide& c) {
  int  rc, done = 0;
  bool pending = false;  // the collision tick after the most recent step has not been evaluated yet
  s->p_valid = false;    // (single-GPU lazies do not mix with this path)
  s->fk_ok   = false;
  // Split ticks (see launch_split_export; the protocol and its proof obligations: DESIGN §5).  In the split form a launch does not
  // see the reports the collective of the previous tick carries, so a stall index T must be ANNOUNCED MRS_PRED_HORIZON launches
  // ahead (displacement bound, step_device.inc) — and the ticks the bound cannot vouch for run in the serial form, whose launches
  // test exactly and hear of each other's reports through the collective in stream order: the first MRS_PRED_HORIZON ticks of
  // every call (the host may have written positions, velocities or airframe constants since the last one) and after every search.
  const bool protocol_split = s->shard_split;  // (the same on every rank: it sets how long a report takes to reach everybody)
  const unsigned lead = (unsigned)(s->fused_lead > 0 ? s->fused_lead : 1), search_ahead = mrs_protocol::search_ahead(lead, protocol_split);
  int serial_left = (int)MRS_PRED_HORIZON;
  s->cstream = s->stream;  // (a call that failed inside a split segment may have left it elsewhere)
  if (dt != s->x_dt) s->x_ok = false;  // the announcements of the last call's final launches assumed its dt: start from a search
  s->x_dt = dt;
  const int nb = (s->n + 63) / 64;
  const volatile unsigned* hw = nullptr;  // pinned host mirror of the control words (exists once a search has run)
  auto split_ok = [&]() {
    // (a communicator of one rank has no boundary and announces nothing: its exact reports need the serial form)
    if (!(protocol_split && s->comm_world > 1 && s->n > 0 && nb >= s->split_min_blocks && (double)s->x_nbnd <= s->split_max_fraction * nb && s->mixed_blocks.empty() && s->stream2 != nullptr))
      return false;
    // Residency: the waves that SPIN inside a split tick — block 0 and the layer-1 blocks of an interior launch waiting for
    // the boundary launch of the previous tick, the boundary blocks waiting for an interior launch — hold their wave slots while they
    // wait.  "Producers are enqueued before consumers" covers the hardware queues, not SIMD and register slots: if spinning interior
    // waves could fill the device, a boundary launch queued behind a late collective would find no slot and the tick would end in the
    // 10-s give-up.  So a rank stays in the serial form unless the spinners leave at least half of the wave slots (at the interior
    // kernel's two waves per SIMD) to everybody else — unless the boundary chain owns compute units of its own (MRS_SPLIT_CU_RESERVE).
    // The count comes from the search (CTL_NL1, read with its head words).
    return mrs_protocol::split_residency_ok(s->x_nl1, s->x_nbnd, s->resident_waves, s->cu_reserve);
  };
  while (done < n_ticks) {
    if (s->x_fallback_left > 0) {
      const int k = n_ticks - done < s->x_fallback_left ? n_ticks - done : s->x_fallback_left;
      if ((rc = full_gather_ticks(s, dt, k, c))) return rc;
      s->x_fallback_left -= k;
      done += k;
      continue;
    }
    if (!s->x_ok) {  // no usable export lists (first tick, lists gone stale): this tick on the search path
      if (s->n > 0 && (rc = launch_step(s, dt, 1))) return rc;
      int  incomplete = 0;
      bool may_leave  = true;
      if ((rc = export_search(s, c, dt, &incomplete, &may_leave))) return rc;
      s->collision_ticks++;
      s->x_ticks++;
      done++;
      pending = false;
      // (the very first ticks of a call stay serial whatever the search says: the host may have written state since the last call)
      if (may_leave || serial_left > 0) serial_left = (int)MRS_PRED_HORIZON;
      if (incomplete) s->x_fallback_left = 64;
      continue;
    }
    // ---- a segment of fused ticks ----
    hw = mrs_collide_host_words(s->cwork);
    if (!hw) return fail(MRS_ERR_HIP, "export-set exchange: the control words of the fused launches do not exist");
    const unsigned first = s->tau + 1;                                 // launch indices run on from the last search
    unsigned       last  = s->tau + (unsigned)(n_ticks - done);       // ... to the end of the call, unless a word says otherwise
    mrs_swarm::Collide off;
    bool     in_split   = false;
    unsigned split_base = 0;
    s->chaos_T = s->chaos_W = 0u;
    while (s->tau < last) {
      const unsigned next = s->tau + 1;
      // (a device that makes no progress is an ERROR here, never a reason to launch anyway: lock-step of the ranks rests on every host
      //  having seen the words of launch next - lead - 1 before it issues launch `next`)
      if ((rc = wait_for_progress(s, hw, next, (int)lead))) return rc;
      unsigned T = stall_word(hw), W = warn_word(hw);
      if (s->chaos_max_us > 0) {
        s->chaos_state ^= s->chaos_state << 13; s->chaos_state ^= s->chaos_state >> 7; s->chaos_state ^= s->chaos_state << 17;
        const unsigned Tf = T, Wf = W;
        if (s->chaos_state & 0x100u) { T = s->chaos_T; W = s->chaos_W; }
        s->chaos_T = Tf; s->chaos_W = Wf;
        usleep((useconds_t)((s->chaos_state >> 16) % (uint64_t)(s->chaos_max_us + 1)));
      }
      last = mrs_protocol::segment_last(last, T, W, lead, search_ahead);
      if (next > last) break;
      if (serial_left == 0 && !in_split && split_ok() && last - s->tau >= 4u) {
        // from the serial form to the split one: what the last collective carried is folded into the control words (the first
        // interior launch reads nothing else), every block counts as finished by launch tau, and the second stream starts behind all that
        HIPCHK(mrs_collide_export_fold_stall(s->cwork, 0u, s->stream));
        HIPCHK(mrs_collide_handoff_init(s->cwork, s->n, s->tau, s->stream));
        HIPCHK(hipEventRecord(s->ev_fork, s->stream));
        HIPCHK(hipStreamWaitEvent(s->stream_i ? s->stream_i : s->stream2, s->ev_fork, 0));
        if (s->stream_b) {
          HIPCHK(hipStreamWaitEvent(s->stream_b, s->ev_fork, 0));
          s->cstream = s->stream_b;
        }
        in_split   = true;
        split_base = s->tau;
      }
      if (in_split) {
        if ((rc = launch_split_export(s, dt, pending ? c : off, split_base))) return rc;
      } else {
        if ((rc = launch_fused_export(s, dt, pending ? c : off))) return rc;
        if (serial_left > 0) serial_left--;
      }
      pending = true;
    }
    if (in_split) {  // back to one stream: everything that follows (fold, search, the caller's work) comes behind the interior launches too
      HIPCHK(hipEventRecord(s->ev_join, s->stream_i ? s->stream_i : s->stream2));
      HIPCHK(hipStreamWaitEvent(s->stream, s->ev_join, 0));
      if (s->stream_b) {
        HIPCHK(hipEventRecord(s->ev_join_b, s->stream_b));
        HIPCHK(hipStreamWaitEvent(s->stream, s->ev_join_b, 0));
        s->cstream = s->stream;
      }
    }
    if (protocol_split) {
      // What a rank's interior launches reported in the last ticks of the segment sits in the header of its export block but has
      // not travelled yet: one more exchange of the export blocks, so that every rank ends the segment with the same words.
      // (All ranks do this, whichever form their own ticks took.)
      const size_t bytes = sizeof(Pos4) * (size_t)(mrs_collide_export_capacity(s->cwork) + 1);
      if ((rc = comm_allgather(s, mrs_collide_export_send(s->cwork), mrs_collide_export_recv(s->cwork), bytes))) return rc;
    }
    HIPCHK(mrs_collide_export_fold_stall(s->cwork, 0u, s->stream));
    // A search that is CERTAIN — this host has seen a stall index, or a warning ended the segment before the call's last tick (every
    // rank will find the same after the fold: words only ever appear) — follows in stream order, and the one synchronisation below
    // serves the segment and the search.  It runs on whatever state the launches left: the state after step T when launches after T
    // turned into no-ops, which is the state the search belongs on either way.
    const unsigned launched = s->tau + 1 - first;
    const bool     early_search = s->early_search && mrs_protocol::search_due(stall_word(hw), warn_word(hw), done + (int)launched < n_ticks);
    if (early_search && (rc = export_search_enqueue(s, c, dt, true))) return rc;
    HIPCHK(hipStreamSynchronize(s->stream));
    const unsigned T = stall_word(hw), W = warn_word(hw);  // identical on every rank
    const unsigned ran = mrs_protocol::ticks_ran(T, first, launched);
    done += (int)ran;
    s->x_ticks += ran;
    s->collision_ticks += ran;
    s->x_noop_ticks += launched - ran;
    if (early_search && !mrs_protocol::search_due(T, W, done < n_ticks)) return fail(MRS_ERR_HIP, "export-set exchange: a search was queued that the folded words do not ask for");
    if (mrs_protocol::search_due(T, W, done < n_ticks)) {
      // the lists are stale after step T / about to be: all ranks search on the state they have now, which also evaluates the
      // collision tick that followed the last step that ran
      int  incomplete = 0;
      bool may_leave  = true;
      if (early_search) {
        if ((rc = export_search_finish(s, c, dt, &incomplete, &may_leave))) return rc;
      } else if ((rc = export_search(s, c, dt, &incomplete, &may_leave))) {
        return rc;
      }
      pending = false;
      // the search ran the displacement bound on the very state the next launches start from: nobody can leave its skin within the
      // horizon -> no stall index <= MRS_PRED_HORIZON can exist, the split form may start with the first tick after the search
      serial_left = may_leave ? (int)MRS_PRED_HORIZON : 0;
      if (incomplete) s->x_fallback_left = 64;
    }
  }
  if (pending && s->n > 0) {  // the last tick's handleCollisions: the export buffer holds the positions after the last step
    CollDev  cd;
    SwarmDev v = s->view();
    HIPCHK(mrs_collide_export_dev(&v, s->cwork, (int64_t)s->comm_rank * s->comm_n_max, 1u, 1, c.crash, c.rebounce, &cd));
    cd.p_out = nullptr;
    HIPCHK(mrs_collide_export_eval(v, cd, s->stream));
  }
  unsigned w[CTL_WORDS];
  HIPCHK(mrs_collide_fused_words(s->cwork, s->stream, w));
  if (w[CTL_BADSLOT]) return fail(MRS_ERR_HIP, "export-set exchange: a listed foreign UAV is not in its owner's export set (" + std::to_string(w[CTL_BADSLOT]) + " entries)");
  if (s->peer_err && *s->peer_err) return peer_failed(s);
  if (w[CTL_ERROR] & 1u) return fail(MRS_ERR_HIP, "split sharded tick: a launch waited in vain for the launch on the other stream (the results of this call are not valid; the stream and the communicator are dead: use a fresh process)");
  if (w[CTL_ERROR] & 2u) return fail(MRS_ERR_HIP, "split sharded tick: a UAV left its skin without the displacement bound announcing it (DESIGN §5) — the results of this call are not valid; run with MRS_SHARD_SPLIT=0 on every rank and report the case");
  if (w[CTL_ERROR] & 0x300u)  // (any rank's error invalidates every rank's results: the ranks that only HEARD of it would otherwise return MRS_OK with a wrong state)
    return fail(MRS_ERR_HIP, std::string("sharded tick: another rank of the swarm reported ") + ((w[CTL_ERROR] & 0x200u) ? "an unannounced skin exit (displacement bound violated)" : "a wait that ran out") +
                                 " — the results of this call are not valid on ANY rank");
  return MRS_OK;
}
}  // namespace mrs_host

extern "C" {

// timerMain on every rank of a sharded swarm (no host synchronisation inside a batch of ticks, everything on the swarm's stream)
int mrs_swarm_tick_sharded_n(mrs_swarm_t* s, double dt, int32_t n_ticks, int32_t enabled, int32_t crash, double rebounce) {
  MRS_ENTER(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (s->comm_world == 0) return fail(MRS_ERR_ARG, "mrs_swarm_comm_init has not been called");
  if (!(dt > 0) || n_ticks < 0) return fail(MRS_ERR_ARG, "bad tick arguments");
  if (n_ticks == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  s->cstream = s->stream;  // (a call that failed inside a split segment may have left it on the boundary chain's stream)
  int rc = upload_types(s, dt);
  if (rc) return rc;
  // Host writes since the last sharded tick (set_state, set_mass, ... possibly on this rank only) do not touch x_ok: all ranks must
  // take the same path.  The first fused launch notices them on the device — a UAV away from its recorded position or with other
  // airframe constants than its record raises the stall word, which travels to every rank in the collective's headers.
  if ((rc = begin_profile(s))) return rc;
  if (!(crash || enabled)) {  // src/multirotor_simulator.cpp:299-301: no collision pass, no exchange
    for (int k = 0; k < n_ticks; k++)
      if (s->n > 0 && (rc = launch_step(s, dt, 1))) return rc;
    return finish_profile(s);
  }
  const mrs_swarm::Collide c{true, enabled, crash, rebounce};
  if (s->exchange == MRS_EXCHANGE_EXPORT_SETS && s->use_lists && s->use_fused)
    rc = export_ticks(s, dt, n_ticks, c);
  else
    rc = full_gather_ticks(s, dt, n_ticks, c);
  // (a peer that never answered is the CAUSE of whatever else went wrong behind it — a search over blocks that never arrived, say)
  if (s->peer_err && *s->peer_err) return peer_failed(s);
  if (rc) return rc;
  s->nbr_dirty = false;
  return finish_profile(s);
}

}  // extern "C"
