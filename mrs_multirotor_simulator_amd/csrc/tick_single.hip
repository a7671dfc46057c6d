// tick_single.hip — the hot path of one GPU: `for (i) uavs_[i]->makeStep(dt)` (src/multirotor_simulator.cpp:211-213) as one launch
// (two half-swarm launches on two streams for runs of steps), and MultirotorSimulator::handleCollisions (:295-359) evaluated lazily
// by the NEXT makeStep launch from neighbour lists: fused launches, searches queued ahead of time, stall + replay.
#include "host_internal.h"
#include "sharded_protocol.h"

namespace mrs_host {
// ---- hot path ----
int launch_part(mrs_swarm* s, double dt, int substeps, int blk0, int nblk, int with_mixed, hipStream_t st) {
  const int variant = s->n_cascade > 0 ? 0 : 1;  // 0 all input modes | 1 model only
  if (s->arith == MRS_ARITH_FAST)
    HIPCHK(mrs_launch_step_fast(s->view(), dt, substeps, variant, blk0, nblk, with_mixed, st));
  else
    HIPCHK(mrs_launch_step_literal(s->view(), dt, substeps, variant, blk0, nblk, with_mixed, st));
  return MRS_OK;
}

int launch_step(mrs_swarm* s, double dt, int substeps) {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  s->region_launches++;
  if (s->profiling == 2) {
    while ((int)s->ev.size() < s->ev_used + 2) {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      s->ev.push_back(e);
    }
    e0 = s->ev[(size_t)s->ev_used];
    e1 = s->ev[(size_t)s->ev_used + 1];
    s->ev_used += 2;
    HIPCHK(hipEventRecord(e0, s->stream));
  }
  int rc = launch_part(s, dt, substeps, 0, (s->n + 63) / 64, 1, s->stream);
  if (rc) return rc;
  if (s->profiling == 2) HIPCHK(hipEventRecord(e1, s->stream));
  return MRS_OK;
}

// one step as two independent half-swarm launches, one per stream (between fork_streams and join_streams)
int launch_step_split(mrs_swarm* s, double dt, int substeps) {
  s->region_launches++;
  const int nb = (s->n + 63) / 64, half = nb / 2;
  int rc = launch_part(s, dt, substeps, 0, half, 1, s->stream);
  if (rc) return rc;
  return launch_part(s, dt, substeps, half, nb - half, 0, s->stream2);
}
int fork_streams(mrs_swarm* s) {
  HIPCHK(hipEventRecord(s->ev_fork, s->stream));
  HIPCHK(hipStreamWaitEvent(s->stream2, s->ev_fork, 0));
  return MRS_OK;
}
int join_streams(mrs_swarm* s) {
  HIPCHK(hipEventRecord(s->ev_join, s->stream2));
  HIPCHK(hipStreamWaitEvent(s->stream, s->ev_join, 0));
  return MRS_OK;
}

int begin_profile(mrs_swarm* s) {
  s->ev_used          = 0;
  s->region_launches  = 0;
  s->prof_split       = false;
  if (s->profiling == 1) {
    while (s->ev.size() < 2) {
      hipEvent_t e;
      HIPCHK(hipEventCreate(&e));
      s->ev.push_back(e);
    }
    HIPCHK(hipEventRecord(s->ev[0], s->stream));
  }
  return MRS_OK;
}

int finish_profile(mrs_swarm* s) {
  if (s->profiling) {  // the timed region ends when every tick of it has really run (replays and the last collision tick included)
    int rc = settle(s);
    if (rc) return rc;
  }
  if (s->profiling == 1) {
    // a split run has recorded its own end events, one per stream, before joining the streams: the region ends when the later of
    // the two halves has finished its last step (the join is stream bookkeeping, not part of the steps)
    if (!s->prof_split) HIPCHK(hipEventRecord(s->ev[1], s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, s->ev[0], s->ev[1]));
    if (s->prof_split) {
      float ms2 = 0;
      HIPCHK(hipEventElapsedTime(&ms2, s->ev[0], s->ev_end2));
      if (ms2 > ms) ms = ms2;
      s->prof_split = false;
    }
    s->last_launches = s->region_launches;
    s->last_ms       = s->region_launches ? (double)ms / s->region_launches : 0.0;
    return MRS_OK;
  }
  if (!s->profiling || s->ev_used == 0) return MRS_OK;
  HIPCHK(hipStreamSynchronize(s->stream));
  double total = 0;
  for (int k = 0; k < s->ev_used; k += 2) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, s->ev[(size_t)k], s->ev[(size_t)k + 1]));
    total += ms;
  }
  s->last_launches = s->ev_used / 2;
  s->last_ms       = total / s->last_launches;
  s->ev_used       = 0;
  return MRS_OK;
}

// ------------------------------------------------------------------------------------------------
// lazily evaluated collision ticks (see the `pend` / `log` members of mrs_swarm)
// ------------------------------------------------------------------------------------------------
// handleCollisions launched on its own: pack + insert / list evaluation, then the query (collide.hip) — the device decides
// whether the search has to be repeated, unless `force` says so
int collide_now(mrs_swarm* s, const mrs_swarm::Collide& c, bool force) {
  int rc = upload_types(s, s->table_dt > 0 ? s->table_dt : 0.001);
  if (rc) return rc;
  HIPCHK(mrs_collide_run_lists(s->view(), &s->cwork, c.crash, c.rebounce, (force || s->nbr_dirty) ? 1 : 0, 0u, s->stream));
  s->nbr_dirty = false;
  s->f_lazy.on = false;  // the pass latched this tick's force
  s->p_valid = true;  // the pass refreshed the position records
  if (!s->use_fused) return MRS_OK;  // (every tick on its own: nobody needs to know, the call stays asynchronous)
  // the host must know whether the lists are complete before a step kernel may evaluate a tick from them
  unsigned w[8];
  HIPCHK(mrs_collide_debug_words(s->cwork, s->stream, w));  // (synchronises the stream)
  s->fk_ok         = w[6] == s->last_overflow;  // no UAV over the list capacity in this pass
  s->last_overflow = w[6];
  return MRS_OK;
}

bool fused_usable(const mrs_swarm* s) {
  return s->use_lists && s->use_fused && s->fk_ok && s->p_valid && !s->nbr_dirty && !s->blocks_dirty && !s->types_dirty && s->cwork != nullptr;
}

// The host runs at most `lead` launches ahead of its device: it waits (spinning on the pinned progress word, no synchronisation)
// until launch `index - lead` has started or some launch has reported stale lists.  A device that makes no progress for
// MRS_PROGRESS_TIMEOUT_S seconds (default 30; a wedged kernel, a collective whose peer is gone) is an error, returned with the words
// the host last saw — after it the swarm's stream, and for a sharded swarm its communicator, must be considered dead: destroy the
// swarm from a fresh process (never re-exec a process that has touched the GPU).

int wait_for_progress(mrs_swarm* s, const volatile unsigned* hw, unsigned index, int lead) {
  if (!hw) return MRS_OK;
  // (launches after a stall index T are no-ops and report no progress: once launch T has started nothing more will come.  A sharded
  //  swarm ANNOUNCES stall indices ahead of time, so a known T does not end the waiting by itself — an earlier one may still turn up,
  //  and the host must not outrun what it has seen)
  auto behind = [&]() { return mrs_protocol::host_is_behind(index, hw[CTL_PROGRESS], stall_word(hw), lead); };
  if (!behind()) return MRS_OK;
  static const double limit_s = getenv("MRS_PROGRESS_TIMEOUT_S") ? atof(getenv("MRS_PROGRESS_TIMEOUT_S")) : 30.0;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned long spins = 1; behind(); spins++) {
    __builtin_ia32_pause();
    if ((spins & 0xFFFFul) != 0) continue;
    // a launch that failed asynchronously never writes its words — on whichever stream of a split tick it ran
    for (hipStream_t st : {s->stream, s->stream2, s->stream_i, s->stream_b}) {
      if (!st) continue;
      const hipError_t q = hipStreamQuery(st);
      if (q != hipSuccess && q != hipErrorNotReady) return fail(MRS_ERR_HIP, std::string("fused launches: ") + hipGetErrorString(q));
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s)
      return fail(MRS_ERR_HIP, "the device made no progress for " + std::to_string((int)limit_s) + " s: waiting for launch " + std::to_string(index - (unsigned)lead) +
                                   ", progress word " + std::to_string(hw[CTL_PROGRESS]) + ", stall word " + std::to_string(stall_word(hw)) + ", warning word " +
                                   std::to_string(warn_word(hw)) + " (the stream" + (s->comm_world > 1 ? " and the communicator are" : " is") + " dead: use a fresh process)");
  }
  return MRS_OK;
}

// one fused launch: evaluate collision tick `e.eval` (if any) from the lists, then makeStep(e.dt)
int launch_fused(mrs_swarm* s, const mrs_swarm::TickRec& e) {
  const volatile unsigned* hw = mrs_collide_host_words(s->cwork);
  // do not run further ahead of the device than a few launches: when the lists go stale at tick T everything queued behind T is wasted
  if (int rcw = wait_for_progress(s, hw, s->tau, s->fused_lead)) return rcw;
  if (e.dt != s->table_dt) {  // (a replayed tick of another dt: the motor-filter constants of the type table follow)
    int rc = upload_types(s, e.dt);
    if (rc) return rc;
  }
  CollDev cd;
  SwarmDev v = s->view();
  HIPCHK(mrs_collide_fused_dev(&v, s->cwork, s->tau + 1, (e.eval.on && !e.searched) ? 1 : 0, e.eval.crash, e.eval.rebounce, &cd));
  const int variant = s->n_cascade > 0 ? 0 : 1;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  s->region_launches++;
  if (s->profiling == 2) {
    while ((int)s->ev.size() < s->ev_used + 2) {
      hipEvent_t ev;
      HIPCHK(hipEventCreate(&ev));
      s->ev.push_back(ev);
    }
    e0 = s->ev[(size_t)s->ev_used];
    e1 = s->ev[(size_t)s->ev_used + 1];
    s->ev_used += 2;
    HIPCHK(hipEventRecord(e0, s->stream));
  }
  if (s->arith == MRS_ARITH_FAST)
    HIPCHK(mrs_launch_step_coll_fast(v, cd, e.dt, variant, 0, s->stream));
  else
    HIPCHK(mrs_launch_step_coll_literal(v, cd, e.dt, variant, 0, s->stream));
  if (s->profiling == 2) HIPCHK(hipEventRecord(e1, s->stream));
  mrs_swarm::TickRec rec = e;
  rec.pin = mrs_collide_fused_pin(s->cwork);
  if (e.eval.on) {
    s->f_lazy.on = !e.searched;  // (a search queued right before the launch latched the force itself)
    if (!e.searched) {
      s->f_lazy     = e.eval;
      s->f_lazy_pin = rec.pin;
    }
  }
  mrs_collide_fused_advance(s->cwork);
  s->tau++;
  s->n_fused++;
  s->log.push_back(rec);
  if (e.eval.on) s->fext_active = true;
  return MRS_OK;
}

// Wait for the device and make good for launches that turned into no-ops: if the lists went stale during step T (a UAV left its
// skin), repeat the search on the state after step T — which also evaluates the collision tick that followed step T — and issue
// the ticks after T again.  Returns with an empty log.
int drain(mrs_swarm* s) {
  if (s->log.empty()) return MRS_OK;
  const volatile unsigned* hw = mrs_collide_host_words(s->cwork);
  for (;;) {
    HIPCHK(hipStreamSynchronize(s->stream));
    const unsigned T = hw ? hw[CTL_STALL] : 0u;
    if (T == 0u || T > s->log.size()) {
      s->log.clear();
      s->tau = 0;
      s->search_mark = 0;
      if (T) return fail(MRS_ERR_HIP, "collision lists: stall index beyond the launch log");
      HIPCHK(mrs_collide_fused_reset(s->cwork, s->stream));  // (progress word back to 0 with tau)
      return MRS_OK;
    }
    s->n_stalls++;
    s->n_noop_launches += (int64_t)s->log.size() - T;
    {  // launch T was the last one that ran.  If it evaluated a collision tick, that force is the latched one (not written: see
       // f_lazy); a launch without evaluation only ever follows a pass that wrote the columns
      const mrs_swarm::TickRec& last = s->log[T - 1];
      s->f_lazy.on = last.eval.on && !last.searched;
      if (s->f_lazy.on) {
        s->f_lazy     = last.eval;
        s->f_lazy_pin = last.pin;
      }
    }
    std::vector<mrs_swarm::TickRec> tail(s->log.begin() + T, s->log.end());
    for (auto& e : tail) e.searched = false;  // (a search queued ahead of time behind the stalled launch did nothing)
    s->log.clear();
    s->tau = 0;
    s->search_mark = 0;
    HIPCHK(mrs_collide_fused_reset(s->cwork, s->stream));
    // the collision tick that followed step T: the first replayed launch was going to evaluate it, or it is the pending one
    mrs_swarm::Collide& c = tail.empty() ? s->pend : tail[0].eval;
    if (c.on) {
      int rc = collide_now(s, c, /*force=*/true);
      if (rc) return rc;
      c.on = false;
    } else {
      s->fk_ok = false;  // nobody needs the lists right now: the next collision tick starts with a search
      s->nbr_dirty = true;
    }
    for (const auto& e : tail) {
      int rc;
      if (fused_usable(s)) {
        if ((rc = launch_fused(s, e))) return rc;
      } else {  // (lists incomplete: dense neighbourhoods) every tick on its own
        if (e.eval.on && (rc = collide_now(s, e.eval, false))) return rc;
        s->p_valid = false;
        s->region_launches++;
        if ((rc = launch_part(s, e.dt, 1, 0, (s->n + 63) / 64, 1, s->stream))) return rc;
      }
      // a pipelined output download packed behind the no-op took the state of an earlier tick: once more, behind the real launch
      if (e.out_ticket >= 0 && s->oslot[e.out_ticket & 1].ticket == e.out_ticket && (rc = issue_outputs(s, e.out_ticket & 1))) return rc;
    }
    if (s->log.empty()) return MRS_OK;
  }
}

// everything the caller asked for so far has happened on the device (asynchronously at most the plain launches)
int settle(mrs_swarm* s) {
  if (s->log.empty() && !s->pend.on && !s->f_lazy.on) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  int rc = drain(s);
  if (rc) return rc;
  if (s->pend.on) {
    const mrs_swarm::Collide c = s->pend;
    s->pend.on = false;
    if ((rc = collide_now(s, c, false))) return rc;
  } else if (s->f_lazy.on) {  // nothing newer overwrites the force the last fused launch evaluated: write it out now
    if ((rc = upload_types(s, s->table_dt > 0 ? s->table_dt : 0.001))) return rc;
    HIPCHK(mrs_collide_latch_force(s->view(), s->cwork, s->f_lazy_pin, s->f_lazy.crash, s->f_lazy.rebounce, s->stream));
    s->f_lazy.on = false;
  }
  return MRS_OK;
}

// one makeStep of every UAV; the collision tick requested since the previous step (if any) is evaluated by the same launch
int step_one(mrs_swarm* s, double dt) {
  int rc;
  if (s->collide_since_step && s->use_lists && s->use_fused) {
    const volatile unsigned* hw = mrs_collide_host_words(s->cwork);
    if (hw && hw[CTL_STALL] != 0u && (rc = drain(s))) return rc;  // seen without synchronising: stop feeding no-ops
    // (a caller that never looks at the swarm — ticks and pipelined downloads only — would let the log grow for ever)
    if (s->log.size() >= 65536u && (rc = drain(s))) return rc;
    if (s->pend.on && !fused_usable(s) && (rc = settle(s))) return rc;  // first tick / after host writes: the pass on its own
    if (fused_usable(s)) {
      mrs_swarm::TickRec e{dt, s->pend, false};
      if (s->pend.on && hw && hw[CTL_WARN] > s->search_mark) {
        // some UAV has used up most of its skin: repeat the search NOW, in stream order — it evaluates the pending collision tick
        // itself — instead of running into the stall a few ticks on (no synchronisation, nothing to replay)
        if ((rc = upload_types(s, dt))) return rc;
        HIPCHK(mrs_collide_run_lists(s->view(), &s->cwork, s->pend.crash, s->pend.rebounce, 1, s->tau, s->stream));  // (tau >= 1: the warning came from a launch of this log)
        e.searched     = true;
        s->search_mark = s->tau;
        s->n_ahead_searches++;
      }
      s->pend.on            = false;
      s->collide_since_step = false;
      return launch_fused(s, e);
    }
  }
  if ((rc = settle(s))) return rc;
  s->collide_since_step = false;
  s->p_valid            = false;  // a plain step kernel does not refresh the position records
  return launch_step(s, dt, 1);
}

}  // namespace mrs_host

extern "C" {

int mrs_swarm_step_n(mrs_swarm_t* s, double dt, int32_t n_steps, int32_t substeps_per_launch) {
  MRS_LOCK(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (!(dt > 0) || n_steps < 0 || substeps_per_launch < 1) return fail(MRS_ERR_ARG, "bad step arguments");
  if (s->n == 0 || n_steps == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  int rc = upload_types(s, dt);
  if (rc) return rc;
  // (a caller holding the handle of mrs_swarm_stream() may have enqueued work of its own without an ABI call: then the stream is
  //  asked — BEFORE the profile's start event goes onto it.  Only then: the query itself puts a marker on the stream, which costs
  //  the run that follows ~12 us — 0.6 us per step of a 20-step region, measured)
  const bool idle_at_entry = s->quiet_seq + 1 == s->op_seq && (!s->stream_exported || hipStreamQuery(s->stream) == hipSuccess);
  if ((rc = begin_profile(s))) return rc;
  bool enqueued = false;  // something of this call is already on the stream
  if (s->collide_since_step && substeps_per_launch == 1) {  // the first step of the run may carry a collision tick
    if ((rc = step_one(s, dt))) return rc;
    n_steps--;
    enqueued = true;
  }
  if (n_steps > 0) {
    enqueued = enqueued || !s->log.empty() || s->pend.on || s->f_lazy.on;  // (what settle is about to issue)
    if ((rc = settle(s))) return rc;
    s->p_valid = false;
    // enough launches to overlap, enough blocks for two useful halves, and no per-launch events to keep in order
    static const int split_min_blocks = getenv("MRS_SPLIT_MIN_BLOCKS") ? atoi(getenv("MRS_SPLIT_MIN_BLOCKS")) : 1024;  // tuning aid
    const bool split = s->split_steps && s->profiling != 2 && (s->n + 63) / 64 >= split_min_blocks && (n_steps + substeps_per_launch - 1) / substeps_per_launch >= 4;
    // (the call right before this one was mrs_swarm_synchronize and nothing has been enqueued since — not even by the lines above:
    //  both streams are idle (upload_types synchronises when it copies), the second one needs no event to wait for)
    const bool quiet = idle_at_entry && !enqueued;
    if (split && !quiet && (rc = fork_streams(s))) return rc;
    int left = n_steps;
    while (left > 0 && rc == MRS_OK) {
      const int sub = left < substeps_per_launch ? left : substeps_per_launch;
      rc = split ? launch_step_split(s, dt, sub) : launch_step(s, dt, sub);
      left -= sub;
    }
    if (split) {  // also on a failed launch: nothing else may touch the state before the second stream has been joined
      if (rc == MRS_OK && s->profiling == 1 && hipEventRecord(s->ev[1], s->stream) == hipSuccess && hipEventRecord(s->ev_end2, s->stream2) == hipSuccess)
        s->prof_split = true;
      const int rcj = join_streams(s);
      if (rc == MRS_OK) rc = rcj;
    }
    if (rc) return rc;
  }
  return finish_profile(s);
}

// makeStep for the UAVs [first, first + count) ONLY — the reference's per-UAV call `uavs_[i]->makeStep(dt)` (src/multirotor_simulator.cpp:212)
// when it is not part of a whole-swarm round.  Per-lane airframe constants (the mixed-block kernels) over a view of the state shifted
// to `first`: any range, any mix of airframes and input modes; results identical to a whole-swarm step of the same UAVs.
int mrs_swarm_step_range(mrs_swarm_t* s, int32_t first, int32_t count, double dt) {
  MRS_ENTER(s);  // (a collision tick still pending is evaluated first: its forces act on this step)
  int rc = check_range(s, first, count);
  if (rc) return rc;
  if (!(dt > 0)) return fail(MRS_ERR_ARG, "bad step arguments");
  if (count == 0) return MRS_OK;
  if (first == 0 && count == s->n) return mrs_swarm_step_n(s, dt, 1, 1);
  HIPCHK(hipSetDevice(s->device));
  if ((rc = upload_types(s, dt))) return rc;
  const int nb = (count + 63) / 64;
  if (nb > s->iota_cap) {
    HIPCHK(hipStreamSynchronize(s->stream));
    if (s->dIota) HIPCHK(hipFree(s->dIota));
    s->dIota = nullptr;
    int cap = 16;
    while (cap < nb) cap *= 2;
    std::vector<int32_t> iota((size_t)cap);
    for (int b = 0; b < cap; b++) iota[(size_t)b] = b;
    HIPCHK(hipMalloc(&s->dIota, sizeof(int32_t) * (size_t)cap));
    HIPCHK(hipMemcpy(s->dIota, iota.data(), sizeof(int32_t) * (size_t)cap, hipMemcpyHostToDevice));
    s->iota_cap = cap;
  }
  SwarmDev v = s->view();
  v.S += first;  // column f of UAV first + i sits at S[f * npad + first + i]: the same stride, shifted base
  v.F += first;
  v.n       = count;
  v.MB      = s->dIota;
  v.n_mixed = nb;
  v.BT      = nullptr;  // (mixed-block kernels read the airframe type per lane, from the flag word)
  v.vl_rec  = nullptr;  // no skin test in a partial step: the next collision tick searches (nbr_dirty below)
  v.vl_flag = nullptr;
  if (s->arith == MRS_ARITH_FAST)
    HIPCHK(mrs_launch_step_fast(v, dt, 1, 0, 0, 0, 1, s->stream));
  else
    HIPCHK(mrs_launch_step_literal(v, dt, 1, 0, 0, 0, 1, s->stream));
  s->p_valid   = false;
  s->nbr_dirty = true;
  return MRS_OK;
}

int mrs_swarm_step(mrs_swarm_t* s, double dt) {
  MRS_LOCK(s); return mrs_swarm_step_n(s, dt, 1, 1); }

int mrs_swarm_pack_positions(mrs_swarm_t* s, void** dev_ptr, int64_t* n_bytes) {
  MRS_ENTER(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  HIPCHK(hipSetDevice(s->device));
  int rc = upload_types(s, s->table_dt > 0 ? s->table_dt : 0.001);
  if (rc) return rc;
  if (!s->dRec) HIPCHK(hipMalloc(&s->dRec, sizeof(PosRecord) * (size_t)s->npad));
  HIPCHK(mrs_launch_pack_positions(s->view(), s->dRec, s->stream));
  if (dev_ptr) *dev_ptr = s->dRec;
  if (n_bytes) *n_bytes = (int64_t)sizeof(PosRecord) * s->n;
  return MRS_OK;
}

int mrs_swarm_pack_positions_to(mrs_swarm_t* s, void* dev_dst) {
  MRS_ENTER(s);
  if (!s || !dev_dst) return fail(MRS_ERR_ARG, "null argument");
  HIPCHK(hipSetDevice(s->device));
  int rc = upload_types(s, s->table_dt > 0 ? s->table_dt : 0.001);
  if (rc) return rc;
  HIPCHK(mrs_launch_pack_positions(s->view(), (PosRecord*)dev_dst, s->stream));
  return MRS_OK;
}

int mrs_swarm_handle_collisions_gathered(mrs_swarm_t* s, const void* dev_records, int64_t n_total, int64_t my_offset, int32_t enabled,
                                         int32_t crash, double rebounce) {
  MRS_ENTER(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (!(crash || enabled)) return MRS_OK;  // src/multirotor_simulator.cpp:299-301
  if (!dev_records || n_total < s->n || my_offset < 0 || my_offset + s->n > n_total) return fail(MRS_ERR_ARG, "bad gathered-record arguments");
  if (s->n == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  s->fext_active = true;
  s->collision_ticks++;
  s->nbr_dirty = true;  // a later single-GPU tick starts from a rebuild
  if (s->use_lists)
    HIPCHK(mrs_collide_run_lists_gathered(s->view(), &s->cwork, (const PosRecord*)dev_records, n_total, my_offset, crash, rebounce, 0, s->stream));
  else
    HIPCHK(mrs_collide_run(s->view(), &s->cwork, (const PosRecord*)dev_records, n_total, my_offset, crash, rebounce, 0, s->stream));
  return MRS_OK;
}

int mrs_swarm_handle_collisions(mrs_swarm_t* s, int32_t enabled, int32_t crash, double rebounce) {
  MRS_LOCK(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (!(crash || enabled)) return MRS_OK;  // src/multirotor_simulator.cpp:299-301
  if (s->comm_world > 1) return fail(MRS_ERR_ARG, "this swarm is one shard of a sharded swarm: use mrs_swarm_tick_sharded_n (the collision pass is collective)");
  if (s->n == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  int rc;
  s->collision_ticks++;
  if (s->use_lists) {
    // two collision ticks without a step in between: the earlier one is evaluated now (its crash flags stay, its forces are overwritten)
    if (s->pend.on && (rc = settle(s))) return rc;
    s->fext_active        = true;
    s->pend               = mrs_swarm::Collide{true, enabled, crash, rebounce};
    s->collide_since_step = true;
    if (!s->use_fused) return settle(s);
    return MRS_OK;  // evaluated by the next step launch, or by settle() when the host looks at the swarm first
  }
  if ((rc = settle(s))) return rc;
  if ((rc = upload_types(s, s->table_dt > 0 ? s->table_dt : 0.001))) return rc;
  s->fext_active = true;
  if (!s->dRec) HIPCHK(hipMalloc(&s->dRec, sizeof(PosRecord) * (size_t)s->npad));
  HIPCHK(mrs_collide_run(s->view(), &s->cwork, s->dRec, s->n, 0, crash, rebounce, /*rec_is_local_scratch=*/1, s->stream));
  return MRS_OK;
}


int mrs_swarm_tick_n(mrs_swarm_t* s, double dt, int32_t n_ticks, int32_t enabled, int32_t crash, double rebounce) {
  MRS_LOCK(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (!(dt > 0) || n_ticks < 0) return fail(MRS_ERR_ARG, "bad tick arguments");
  if (s->n == 0 || n_ticks == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  int rc = upload_types(s, dt);
  if (rc) return rc;
  if ((rc = begin_profile(s))) return rc;
  for (int k = 0; k < n_ticks; k++) {
    if ((rc = step_one(s, dt))) return rc;
    if ((rc = mrs_swarm_handle_collisions(s, enabled, crash, rebounce))) return rc;
  }
  return finish_profile(s);
}

// measurement hook (bench.py roofline_collision): `reps` neighbour searches of the single-GPU collision pass back to back on the
// swarm's stream — pack + insert, then the list-building query, exactly what a tick that repeats the search launches — between
// two hipEvents.  The forces / crash flags latched are those of handleCollisions(enabled, crash, rebounce) on the current positions.
int mrs_swarm_debug_neighbour_lists(mrs_swarm_t* s, int32_t crash, double rebounce, uint32_t* count, uint32_t* nbr, int32_t list_cap_in, int32_t* list_cap,
                                    double* list_radius) {
  MRS_ENTER(s);
  if (!s || !count || !nbr || list_cap_in < 1 || !list_cap || !list_radius) return fail(MRS_ERR_ARG, "bad neighbour-list arguments");
  if (s->comm_world > 1 || !s->use_lists) return fail(MRS_ERR_ARG, "neighbour lists: single-GPU swarms with neighbour lists only");
  mrs_collide_list_geometry(list_cap, list_radius);
  if (s->n == 0) return MRS_OK;
  HIPCHK(hipSetDevice(s->device));
  int rc = settle(s);
  if (rc) return rc;
  if ((rc = upload_types(s, s->table_dt > 0 ? s->table_dt : 0.001))) return rc;
  const mrs_swarm::Collide c{true, 1, crash, rebounce};
  if ((rc = collide_now(s, c, /*force=*/true))) return rc;
  s->fext_active = true;
  HIPCHK(mrs_collide_copy_lists(s->cwork, s->n, count, nbr, list_cap_in, s->stream));
  return MRS_OK;
}

int mrs_swarm_debug_search_ms(mrs_swarm_t* s, int32_t reps, int32_t crash, double rebounce, double* avg_ms) {
  MRS_ENTER(s);
  if (!s || reps < 1 || !avg_ms) return fail(MRS_ERR_ARG, "bad search-timing arguments");
  if (s->comm_world > 1 || !s->use_lists) return fail(MRS_ERR_ARG, "search timing: single-GPU swarms with neighbour lists only");
  if (s->n == 0) { *avg_ms = 0.0; return MRS_OK; }
  HIPCHK(hipSetDevice(s->device));
  int rc = upload_types(s, s->table_dt > 0 ? s->table_dt : 0.001);
  if (rc) return rc;
  const mrs_swarm::Collide c{true, 1, crash, rebounce};
  if ((rc = collide_now(s, c, /*force=*/true))) return rc;  // buffers exist, tables are in their steady state
  while (s->ev.size() < 2) {
    hipEvent_t e;
    HIPCHK(hipEventCreate(&e));
    s->ev.push_back(e);
  }
  HIPCHK(hipEventRecord(s->ev[0], s->stream));
  for (int k = 0; k < reps; k++) HIPCHK(mrs_collide_run_lists(s->view(), &s->cwork, crash, rebounce, 1, 0u, s->stream));
  HIPCHK(hipEventRecord(s->ev[1], s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, s->ev[0], s->ev[1]));
  *avg_ms = (double)ms / reps;
  s->fext_active = true;
  return collide_now(s, c, /*force=*/true);  // host bookkeeping (list completeness, lazies) as after any stand-alone pass
}

int mrs_swarm_set_profiling(mrs_swarm_t* s, int32_t enabled) {
  MRS_LOCK(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  s->profiling = enabled < 0 ? 0 : (enabled > 2 ? 2 : enabled);
  return MRS_OK;
}

int mrs_swarm_last_step_kernel_ms(mrs_swarm_t* s, double* avg_ms, int32_t* n_launches) {
  MRS_LOCK(s);
  if (!s) return fail(MRS_ERR_ARG, "null swarm");
  if (avg_ms) *avg_ms = s->last_ms;
  if (n_launches) *n_launches = s->last_launches;
  return MRS_OK;
}

}  // extern "C"
