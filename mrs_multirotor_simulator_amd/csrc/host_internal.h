// host_internal.h — what the host-side units of libmrs_swarm.so share: the swarm object, the launchers the kernel units export,
// and the helpers one unit defines for the others.  Units (each citing the reference calls it replaces in include/mrs_swarm.h):
//   host_api.hip          C ABI: parameters, lifetime, construction, commands, state access, publisher payloads, probes
//   tick_single.hip       the hot path of one GPU: makeStep launches, lazily evaluated collision ticks (fused launches, stall + replay)
//   tick_sharded.hip      the sharded tick: communicator bookkeeping, search path, serial and split segments of the export-set exchange
//   transport_rccl.hip    RCCL bound at run time (dlopen)
//   transport_local.hip   in-process loopback group, caller-supplied all-gather, measurement stand-in
//   transport_peer.hip    peer-window exchange (direct writes into the peers' device memory)
// sharded_protocol.h holds the pure decision functions of the sharded protocol (tested without a GPU: tests/cpp/sharded_protocol_test.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <dlfcn.h>

#include "../../include/mrs_swarm.h"
#include "swarm_layout.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

// ---- launchers exported by the kernel units (step_kernel_*.hip, collide.hip, outputs.hip) ----
extern "C" hipError_t mrs_launch_step_literal(SwarmDev sw, double dt, int substeps, int cascade, int blk0, int nblk, int with_mixed, hipStream_t st);
extern "C" hipError_t mrs_launch_step_fast(SwarmDev sw, double dt, int substeps, int cascade, int blk0, int nblk, int with_mixed, hipStream_t st);
extern "C" hipError_t mrs_launch_pid_probe_literal(const double*, const double*, const double*, const double*, const double*, double*, int, int, hipStream_t);
extern "C" hipError_t mrs_launch_pid_probe_fast(const double*, const double*, const double*, const double*, const double*, double*, int, int, hipStream_t);
extern "C" hipError_t mrs_launch_pid_update_probe_literal(const double*, double*, const double*, const double*, double*, int, hipStream_t);
extern "C" hipError_t mrs_launch_pid_update_probe_fast(const double*, double*, const double*, const double*, double*, int, hipStream_t);
extern "C" hipError_t mrs_launch_component_probe_literal(SwarmDev, int, int, int, const double*, int, double*, int, double, hipStream_t);
extern "C" hipError_t mrs_launch_component_probe_fast(SwarmDev, int, int, int, const double*, int, double*, int, double, hipStream_t);
// collide.hip
extern "C" hipError_t mrs_launch_flags_update(uint32_t* F, int first, int count, uint32_t and_mask, uint32_t or_mask, hipStream_t st);
extern "C" hipError_t mrs_launch_pack_positions(SwarmDev sw, PosRecord* out, hipStream_t st);
struct CollideWork;
extern "C" hipError_t mrs_collide_run(SwarmDev sw, CollideWork** work, const PosRecord* rec, long long n_total, long long my_offset,
                                      int crash, double rebounce, int rec_is_local_scratch, hipStream_t st);
extern "C" hipError_t mrs_collide_run_lists(SwarmDev sw, CollideWork** work, int crash, double rebounce, int force_rebuild, unsigned guard_tau,
                                            hipStream_t st);
extern "C" hipError_t mrs_collide_run_lists_gathered(SwarmDev sw, CollideWork** work, const PosRecord* rec, long long n_total, long long my_offset,
                                                     int crash, double rebounce, int force_rebuild, hipStream_t st);
extern "C" hipError_t mrs_collide_export_prepare(SwarmDev sw, CollideWork** work, int world, long long cap, int zero, hipStream_t st);
extern "C" long long  mrs_collide_export_capacity(const CollideWork* w);
extern "C" const uint32_t* mrs_collide_host_heads(const CollideWork* w);
extern "C" void       mrs_collide_host_words_reset(CollideWork* w);
extern "C" void*      mrs_collide_export_send(const CollideWork* w);
extern "C" void*      mrs_collide_export_recv(const CollideWork* w);
extern "C" hipError_t mrs_collide_export_mark(SwarmDev sw, CollideWork* w, long long n_max, long long map_words, int rank, uint32_t* map_send, double pred_hdt,
                                              double rebounce, hipStream_t st);
extern "C" hipError_t mrs_collide_export_translate(SwarmDev sw, CollideWork* w, long long n_max, long long map_stride, int rank, const uint32_t* maps,
                                                   const PosRecord* rec_all, hipStream_t st);
extern "C" hipError_t mrs_collide_export_dev(const SwarmDev* sw, CollideWork* w, long long my_offset, unsigned tau, int eval, int crash, double rebounce,
                                             CollDev* cd);
extern "C" hipError_t mrs_collide_export_eval(SwarmDev sw, CollDev cd, hipStream_t st);
extern "C" hipError_t mrs_collide_export_fold_stall(CollideWork* w, unsigned progress_tau, hipStream_t st);
extern "C" hipError_t mrs_collide_fused_words(const CollideWork* w, hipStream_t st, unsigned* out8);
extern "C" void       mrs_collide_invalidate_gathered(CollideWork* w);
extern "C" void mrs_collide_step_hook(const CollideWork* w, const PosRecord** rec, uint32_t** flag, double* lim2);
extern "C" void       mrs_collide_list_geometry(int* list_cap, double* list_radius);
extern "C" hipError_t mrs_collide_copy_lists(const CollideWork* w, long long n, uint32_t* count, uint32_t* nbr, int rows, hipStream_t st);
extern "C" hipError_t mrs_collide_rebuilds(const CollideWork* w, hipStream_t st, unsigned* out);
extern "C" hipError_t mrs_collide_debug_words(const CollideWork* w, hipStream_t st, unsigned* out8);
extern "C" void mrs_collide_free(CollideWork* w);
extern "C" hipError_t mrs_launch_step_coll_literal(SwarmDev sw, CollDev cd, double dt, int variant, int grid_blocks, hipStream_t st);
extern "C" hipError_t mrs_launch_step_coll_fast(SwarmDev sw, CollDev cd, double dt, int variant, int grid_blocks, hipStream_t st);
extern "C" void       mrs_collide_export_part(CollDev* cd, int part, unsigned n_bnd, double dt, int announce);
extern "C" hipError_t mrs_collide_handoff_init(CollideWork* w, int n, unsigned tau, hipStream_t st);
extern "C" const uint32_t* mrs_collide_ctl_words(const CollideWork* w);
extern "C" hipError_t mrs_collide_heads_to_host(CollideWork* w, const uint32_t* maps, long long stride, int world, int halo, const uint32_t** out, hipStream_t st);
// halo exchange of a search tick (collide.hip)
extern "C" hipError_t mrs_collide_halo_prepare(CollideWork** work, int world, long long cap, hipStream_t st);
extern "C" long long  mrs_collide_halo_capacity(const CollideWork* w);
extern "C" void*      mrs_collide_halo_send(const CollideWork* w);
extern "C" void*      mrs_collide_halo_recv(const CollideWork* w);
extern "C" int        mrs_collide_halo_ready(const CollideWork* w, long long n_total);
extern "C" hipError_t mrs_collide_halo_select(SwarmDev sw, CollideWork* w, PosRecord* table, long long n_max, int rank, int world, const uint32_t* maps,
                                              long long stride, int boxw, int not_ready, hipStream_t st);
extern "C" void       mrs_collide_set_box_out(CollideWork** work, double* box_out);
extern "C" hipError_t mrs_collide_run_lists_halo(SwarmDev sw, CollideWork** work, PosRecord* table, long long n_total, long long n_max, int rank, int world,
                                                 int crash, double rebounce, hipStream_t st);
extern "C" hipError_t mrs_collide_fused_dev(const SwarmDev* sw, CollideWork* w, unsigned tau, int eval, int crash, double rebounce, CollDev* cd);
extern "C" void mrs_collide_fused_advance(CollideWork* w);
extern "C" const volatile unsigned* mrs_collide_host_words(const CollideWork* w);
extern "C" hipError_t mrs_collide_fused_reset(CollideWork* w, hipStream_t st);
extern "C" hipError_t mrs_collide_latch_force(SwarmDev sw, CollideWork* w, int pin, int crash, double rebounce, hipStream_t st);
extern "C" int        mrs_collide_fused_pin(const CollideWork* w);
// outputs.hip
extern "C" hipError_t mrs_launch_timeout_input(SwarmDev sw, int first, int count, hipStream_t st);
extern "C" hipError_t mrs_launch_unpack_rows(SwarmDev sw, const double* rows, int stride, int width, int base, int first, int count, hipStream_t st);
extern "C" hipError_t mrs_launch_pack_outputs(SwarmDev sw, int first, int count, mrs_uav_output_t* dev_out, hipStream_t st);
extern "C" hipError_t mrs_launch_pack_states(SwarmDev sw, int first, int count, mrs_uav_state_t* dev_out, hipStream_t st);
extern "C" hipError_t mrs_launch_peer_allgather(const MrsPeerWindows* pw, const void* send, void* recv, size_t bytes, int rank, int world, unsigned seq,
                                                size_t slot_bytes, unsigned* tickets, unsigned ticket_total, unsigned* err_host, unsigned* bpp_out,
                                                hipStream_t st);
extern "C" hipError_t mrs_launch_standin_gather(const void* send, void* recv, size_t bytes, int rank, int world, double latency_us, int kind, long long aux, double width,
                                                hipStream_t st);
extern "C" hipError_t mrs_launch_stream_delay(hipStream_t st, double microseconds);

namespace mrs_host {
int fail(int code, const std::string& msg);  // remembers msg for mrs_last_error() (thread-local), returns code
}
using mrs_host::fail;
#define HIPCHK(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) return fail(MRS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)


struct TypeKey {  // everything that distinguishes two UavSystem parameterisations
  mrs_model_params_t    mp;  // takeoff_patch_enabled normalised to 0 (it is per-UAV mutable state)
  mrs_mixer_params_t    mixer;
  mrs_rate_params_t     rate;
  mrs_attitude_params_t att;
  mrs_velocity_params_t vel;
  mrs_position_params_t pos;
};

// RCCL entry points, bound at run time (transport_rccl.hip)
namespace mrs_host {
struct NcclId { char internal[128]; };  // ncclUniqueId: NCCL_UNIQUE_ID_BYTES = 128, passed BY VALUE to ncclCommInitRank
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclId*)                                           = nullptr;
  int (*CommInitRank)(void**, int, NcclId, int)                         = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*)                                             = nullptr;
  int (*CommCount)(void*, int*)                                         = nullptr;
  const char* (*GetErrorString)(int)                                    = nullptr;
};
extern RcclApi g_rccl;
}  // namespace mrs_host

struct mrs_swarm {
  // every C-ABI call on a swarm is serialised (the reference's subscriber callbacks run concurrently with timerMain and are
  // serialised by mutex_uav_system_, src/uav_system_ros.cpp:267,702): recursive because entry points call each other
  std::recursive_mutex mtx;
  int32_t  n = 0, npad = 0, device = 0;
  int32_t  arith = MRS_ARITH_LITERAL;
  hipStream_t stream = nullptr;
  // a run of steps without collisions in between is issued as two half-swarm launches per step on two streams: the halves are
  // independent, so the drain of one launch overlaps the ramp of the other (tools/two_streams.py: +11 % at 100 k, +19 % at 200 k)
  hipStream_t stream2 = nullptr;
  // Split sharded ticks with reserved compute units (MRS_SPLIT_CU_RESERVE = R > 0; tools/cu_mask_probe.hip): the boundary launch and the
  // collective run on `stream_b`, whose queue may use R CUs only (mask bits 0..R-1: bit i is CU i / 8 of XCD i % 8), the interior
  // launch on `stream_i`, whose queue uses all the others — the lone waves of the boundary chain no longer share SIMDs with the
  // streaming interior waves.  `cstream`: where collectives and boundary launches go right now (`stream` outside split segments).
  hipStream_t stream_b = nullptr, stream_i = nullptr, cstream = nullptr;
  hipEvent_t  ev_join_b = nullptr;
  hipEvent_t  ev_copy = nullptr;   // mrs_swarm_copy_uavs between two swarms: orders the copy against the other swarm's stream
  int         cu_reserve = 0;
  hipEvent_t  ev_fork = nullptr, ev_join = nullptr;
  hipEvent_t  ev_end2 = nullptr;   // profiling: end of the second stream's part of a split run (recorded before the join)
  bool        prof_split = false;  // the end events of the running profile region have been recorded by the split run itself
  // every C-ABI call bumps op_seq (MRS_LOCK); mrs_swarm_synchronize notes the value it leaves behind: a run of steps that is the very
  // next call finds both streams idle and starts its second stream without the fork event
  uint64_t    op_seq = 0, quiet_seq = ~0ull;
  mutable bool stream_exported = false;  // mrs_swarm_stream has handed the stream to the caller, who may enqueue work behind the library's back
  bool        split_steps = true;  // tuning: MRS_SPLIT_STREAMS=0
  // native multi-GPU collision exchange (mrs_swarm_comm_init): RCCL all-gather issued on `stream`
  void*      rccl_comm = nullptr;
  int        comm_world = 0, comm_rank = 0;   // comm_world > 0: a communicator of some kind is bound
  int64_t    comm_n_total = 0, comm_n_max = 0;
  PosRecord* comm_send = nullptr;
  PosRecord* comm_recv = nullptr;
  // collective backend other than RCCL: a caller-supplied all-gather (mrs_swarm_comm_init_custom) or an in-process group of
  // swarms driven by one host thread each (mrs_swarm_comm_init_loopback)
  mrs_allgather_fn      comm_fn = nullptr;
  void*                 comm_user = nullptr;
  mrs_loopback_group_t* comm_group = nullptr;
  // measurement stand-in (mrs_swarm_comm_init_standin): ONE rank of `world` alone on the device; its neighbours in the slab order
  // are images of itself one slab width away, and every collective costs a fixed latency
  bool   comm_standin = false;
  double standin_delay_us = 0.0, standin_width = 0.0, standin_gbps = 0.0;  // gbps > 0: bytes of a collective / that rate on top of the latency
  // peer-window exchange (mrs_swarm_peer_window_create / mrs_swarm_comm_init_peer; collide.hip k_peer_allgather): ranks write their
  // blocks straight into each other's device memory, one kernel per collective on the swarm's stream, no collective library
  bool               comm_peer = false;
  void*              peer_window = nullptr;       // this rank's window (fine-grained device memory)
  size_t             peer_slot_bytes = 0, peer_window_bytes = 0;
  int                peer_world = 0, peer_rank = 0;
  int64_t            peer_n_total = 0;
  MrsPeerWindows     peer_windows{};              // every rank's window as this process addresses it
  std::vector<void*> peer_opened;                 // the ones mapped here through an IPC handle (closed by comm_destroy)
  unsigned           peer_seq = 0, peer_tickets = 0;
  unsigned*          peer_ticket = nullptr;       // device words, one per peer: blocks of all exchange kernels so far that have pushed their share for it
  unsigned*          peer_err = nullptr;          // pinned host word: an exchange kernel waited in vain for a peer
  // export-set exchange: between two searches only boundary UAVs travel
  int       exchange = MRS_EXCHANGE_EXPORT_SETS;
  bool      x_ok = false;           // export lists are live: the next tick can be a fused launch + export-set all-gather
  int       x_fallback_left = 0;    // ticks to stay on the full exchange after an incomplete (overflowing) search
  uint32_t* x_map_send = nullptr;   // [2 + n_max] slot map of this rank
  uint32_t* x_map_recv = nullptr;   // [world][2 + n_max]
  int64_t   x_export_count = 0;
  std::vector<unsigned> x_last_overflow;
  int64_t   x_searches = 0, x_ticks = 0, x_noop_ticks = 0;
  // split sharded ticks: between two searches the blocks that hold a boundary UAV are stepped by a small launch on
  // `stream`, followed there by the collective, while the interior launch runs on `stream2` and never waits for a collective
  // halo exchange of a search (collide.hip mrs_collide_halo_*; MRS_SEARCH_HALO=0: every search gathers all records, as up to round 4)
  bool      halo_trace = false;  // MRS_HALO_TRACE=1: one line on stderr per halo search
  bool      halo_enabled = true, halo_ok = false, halo_pass = false;  // ok: the last search left every rank's box in the maps; pass: the queued search is a halo one
  int64_t   halo_cap = 0, halo_backoff = 0, halo_backoff_len = 16;                                            // entries per block of the next halo search — the same on every rank
  int64_t   x_halo_searches = 0, x_halo_repeats = 0;
  bool      early_search = true;    // tuning: MRS_EARLY_SEARCH=0 — a certain search waits for the segment's synchronisation (round 3)
  bool      shard_split = true;     // tuning: MRS_SHARD_SPLIT=0 keeps every tick in the serial form (fused launch, then the collective)
  int       split_min_blocks = 512; // tuning / tests: MRS_SHARD_SPLIT_MIN_BLOCKS
  double    split_max_fraction = 0.25;  // ... and MRS_SHARD_SPLIT_MAX_FRACTION: the boundary launch may cover at most this share of the blocks
  uint32_t  x_nbnd = 0;             // boundary blocks of this rank as of the last search
  uint32_t  x_nl1 = 0xFFFFFFFFu;    // ... and its interior blocks that list a UAV of a boundary block (0xFFFFFFFF: no search yet)
  double    x_dt = -1.0;            // dt of the previous call: the announcements of its last launches assumed it
  int       resident_waves = 2048;  // wave slots of the device at the interior kernel's occupancy (2 per SIMD): see split_ok()
  int64_t   x_split_ticks = 0;
  // test hook (mrs_swarm_debug_chaos): this rank's host sleeps a random time before every launch of a sharded tick and, half of
  // the time, decides on the stall / warning words as it read them one launch earlier (still within what the protocol guarantees)
  int       chaos_max_us = 0;
  uint64_t  chaos_state = 0x9E3779B97F4A7C15ull;
  unsigned  chaos_T = 0, chaos_W = 0;
  double*   dS = nullptr;
  uint32_t* dF = nullptr;
  TypeParams* dT = nullptr;
  int32_t   dT_cap = 0;
  unsigned long long* dDiag = nullptr;
  std::vector<TypeKey>    keys;
  std::vector<TypeParams> tparams;
  std::map<std::string, int> key_index;
  std::vector<uint16_t>   uav_type;
  std::vector<uint8_t>    uav_mode;   // host mirror of active_input_: picks the kernel variant
  int64_t n_cascade = 0;              // UAVs whose mode needs the controller cascade
  bool   types_dirty = true;
  double table_dt    = -1.0;
  // publisher payloads: device pack buffer + pinned host staging
  mrs_uav_output_t* dOut = nullptr;
  mrs_uav_output_t* hOut = nullptr;
  int32_t           out_cap = 0;
  mrs_uav_state_t*  dSt = nullptr;   // packed states (mrs_swarm_get_states): device buffer + pinned host staging
  mrs_uav_state_t*  hSt = nullptr;
  int32_t           st_cap = 0;
  // Pipelined publisher download (mrs_swarm_get_outputs_async / mrs_swarm_outputs_wait): two pack buffers with a pinned block each;
  // the pack kernel runs on the step stream behind everything queued so far, the device-to-host copy on `stream_io`, so the download
  // of tick t overlaps step t + 1.  `packed` orders copy behind pack, `done` is what the host waits for (and what the next pack into
  // the same buffer waits for on the device).
  struct OutSlot {
    mrs_uav_output_t *d = nullptr, *h = nullptr;
    int32_t           cap = 0, ticket = -1, first = 0, count = 0;
    hipEvent_t        packed = nullptr, done = nullptr;
  };
  OutSlot     oslot[2];
  // copy streams: download (stream_io) and command upload (stream_up) — one per direction, PCIe is full duplex.  (The download cut
  // in two halves on two streams — two DMA engines — was measured and dropped: 0.40 against 0.38 ms per 13.6-MB tick.)
  hipStream_t stream_io = nullptr, stream_up = nullptr;
  int32_t     out_tickets = 0;
  // staged command upload: two pinned row blocks + device copies, handed out in turn — the caller fills block k + 1 while the copy of
  // block k may still be in flight (`copied`: the host may refill the rows; `unpacked`: the device copy may be overwritten)
  struct InSlot {
    double *   h = nullptr, *d = nullptr;
    int64_t    cap = 0;  // doubles
    hipEvent_t copied = nullptr, unpacked = nullptr;
  };
  InSlot  islot[2];
  int     in_turn = 0;   // the block the last mrs_swarm_input_staging handed out
  // collision scratch
  PosRecord*   dRec = nullptr;
  CollideWork* cwork = nullptr;
  bool         use_lists = true;   // single-GPU collision ticks reuse neighbour lists between rebuilds (tuning: MRS_NEIGHBOUR_LISTS=0)
  bool         nbr_dirty = true;   // the host wrote positions or airframe constants since the last collision tick
  int64_t      collision_ticks = 0;
  // Lazily evaluated collision ticks (single-GPU neighbour lists).  handleCollisions is not launched when it is called: it is
  // evaluated by the NEXT makeStep launch, whose prologue forms the forces from the neighbour lists (step_device.inc *_coll), or
  // by settle() when the host looks at the swarm first.  `log` holds the launches the device has not confirmed yet: when a UAV
  // leaves its skin during step T the launches after T turn into no-ops, and the host repeats the search and replays them.
  struct Collide { bool on = false; int enabled = 0, crash = 0; double rebounce = 0.0; };
  // one fused launch: the collision tick it evaluates first (searched: a search queued right before it has done that), then makeStep(dt)
  // pin: which position buffer the launch read; out_ticket: a pipelined output download packed right behind this launch (re-issued
  // when the launch is replayed after a stall: what it packed then was the state of an earlier tick)
  struct TickRec { double dt; Collide eval; bool searched; int pin; int out_ticket = -1; };
  Collide              pend;                        // requested after the most recent step, not evaluated yet
  // A fused launch consumes the force it evaluates from registers and does not write the F_ext columns (24 B per UAV and tick).
  // While f_lazy.on those columns are stale: the latched force is "collision tick f_lazy on the position records f_lazy_pin",
  // re-derived by settle() (or overwritten by the next search) before anything reads the columns.
  Collide              f_lazy;
  int                  f_lazy_pin = 0;
  bool                 collide_since_step = false;  // ... or evaluated already: either way the next step keeps the fused form
  bool                 p_valid = false;             // the position records hold the positions after the most recent step
  bool                 fk_ok   = false;             // the lists are complete (no UAV over the list capacity) and in local mode
  unsigned             last_overflow = 0;
  std::vector<TickRec> log;
  uint32_t             tau = 0;                     // tick index of the last fused launch since the stream was last drained
  bool                 use_fused = true;            // tuning: MRS_FUSED_COLLISIONS=0 launches every collision tick on its own
  int                  fused_lead = 3;              // launches the host may run ahead of the device (MRS_FUSED_LEAD; 2-3 measured best, 8: stalls)
  uint32_t             search_mark = 0;             // tick index behind which the last ahead-of-time search was queued
  int64_t              n_ahead_searches = 0;
  int64_t              n_stalls = 0, n_noop_launches = 0, n_fused = 0;
  // profiling
  int  profiling = 0;  // 0 off, 1 one event pair around the whole step_n/tick_n region, 2 one pair per step launch
  std::vector<hipEvent_t> ev;
  int  ev_used = 0;
  int  region_launches = 0;
  double last_ms = 0.0;
  int    last_launches = 0;
  std::vector<double>   stage;  // host staging column
  std::vector<uint32_t> stage_u;

  // per-64-block airframe type (0xFFFF = mixed) and the list of mixed blocks
  std::vector<uint32_t> block_type;
  std::vector<int32_t>  mixed_blocks;
  uint32_t* dBT = nullptr;
  int32_t*  dMB = nullptr;
  bool      blocks_dirty = true;
  int32_t*  dIota = nullptr;  // 0, 1, 2, ...: block list of a partial step (mrs_swarm_step_range)
  int       iota_cap = 0;

  bool      fext_active = false;  // apply_force / collisions were used at least once

  SwarmDev view() const {
    SwarmDev v{dS, dF, dT, dDiag, dBT, dMB, n, npad, (int32_t)mixed_blocks.size(), fext_active ? 1u : 0u, nullptr, nullptr, 0.0, 0, arith == MRS_ARITH_FAST ? 1 : 0};
    mrs_collide_step_hook(cwork, &v.vl_rec, &v.vl_flag, &v.vl_lim2);
    return v;
  }
};

#define MRS_LOCK(s)                                                                      \
  std::unique_lock<std::recursive_mutex> _lk;                                            \
  if (s) {                                                                               \
    _lk = std::unique_lock<std::recursive_mutex>(const_cast<mrs_swarm*>(s)->mtx);        \
    const_cast<mrs_swarm*>(s)->op_seq++;                                                 \
  }

namespace mrs_host {
int settle(mrs_swarm* s);
int drain(mrs_swarm* s);
}
// entry of every call that reads or writes swarm state: collision ticks still pending on the device side are evaluated first
#define MRS_ENTER(s)                                          \
  MRS_LOCK(s);                                                \
  if (s) {                                                    \
    int _src = settle(const_cast<mrs_swarm*>(s));             \
    if (_src) return _src;                                    \
  }


// entry of a call that writes nothing but command / feed-forward columns and mode flags (setInput, setFeedforward, the input timeout):
// the launches queued so far must have RUN — a replay after a stall would otherwise step with the new commands — but a collision tick
// still pending stays pending: its evaluation reads positions only, and the next makeStep launch keeps the fused form.  (With
// MRS_ENTER every command between two ticks cost a search pass of its own and two synchronisations.)
#define MRS_ENTER_COMMANDS(s)                                                   \
  MRS_LOCK(s);                                                                  \
  if (s && !const_cast<mrs_swarm*>(s)->log.empty()) {                           \
    if (hipSetDevice(s->device) != hipSuccess) return fail(MRS_ERR_HIP, "hipSetDevice"); \
    int _src = mrs_host::drain(const_cast<mrs_swarm*>(s));                      \
    if (_src) return _src;                                                      \
  }

namespace mrs_host {
// ---- host_api.hip ----
int     issue_outputs(mrs_swarm* s, int slot);  // pack + copy of the pipelined download held by oslot[slot]
void    track_mode(mrs_swarm* s, int first, int count, int mode);
int     check_range(const mrs_swarm* s, int first, int count);
int     intern_type(mrs_swarm* s, const TypeKey& k, int* out);
TypeKey make_key(const mrs_model_params_t* p);
int     upload_blocks(mrs_swarm* s);
int     upload_types(mrs_swarm* s, double dt);
int     put_column(mrs_swarm* s, int f, int first, int count, const double* col);
int     fill_column(mrs_swarm* s, int f, int first, int count, double value);
int     put_strided(mrs_swarm* s, int f, int first, int count, const double* src, int width, int j);
int     get_strided(mrs_swarm* s, int f, int first, int count, double* dst, int width, int j);
int     flags_update(mrs_swarm* s, int first, int count, uint32_t and_mask, uint32_t or_mask);
// ---- tick_single.hip ----
int  launch_part(mrs_swarm* s, double dt, int substeps, int blk0, int nblk, int with_mixed, hipStream_t st);
int  launch_step(mrs_swarm* s, double dt, int substeps);
int  begin_profile(mrs_swarm* s);
int  finish_profile(mrs_swarm* s);
int  collide_now(mrs_swarm* s, const mrs_swarm::Collide& c, bool force);
int  wait_for_progress(mrs_swarm* s, const volatile unsigned* hw, unsigned index, int lead);
int  step_one(mrs_swarm* s, double dt);
int  drain(mrs_swarm* s);
inline unsigned min_nonzero(unsigned a, unsigned b) { return a == 0u ? b : (b == 0u ? a : (a < b ? a : b)); }
// the stall / warning index the host knows of: each chain of a split tick keeps mirrors of its own (one writer per word)
inline unsigned stall_word(const volatile unsigned* hw) { return min_nonzero(hw[CTL_STALL], hw[CTL_STALL2]); }
inline unsigned warn_word(const volatile unsigned* hw) { return min_nonzero(hw[CTL_WARN], hw[CTL_WARN2]); }
// ---- transports (transport_*.hip) and the communicator bookkeeping (tick_sharded.hip) ----
int  rccl_load(const char* path);
int  rccl_check(int rc, const char* what);
int  loopback_allgather(mrs_loopback_group_t* g, int rank, const void* send, void* recv, size_t bytes, hipStream_t st);
int  standin_allgather(mrs_swarm* s, const void* send, void* recv, size_t bytes);
int  peer_allgather(mrs_swarm* s, const void* send, void* recv, size_t bytes);
int  peer_failed(mrs_swarm* s);
void peer_release(mrs_swarm* s);
int  comm_allgather(mrs_swarm* s, const void* send, void* recv, size_t bytes);
int  comm_setup(mrs_swarm* s, int world, int rank, int64_t n_total);
int  comm_buffers(mrs_swarm* s, int world, int rank, int64_t n_total);
// words of one rank's slot map in the collective of a search: [export count, overflow count, n_max slots], padded to whole 16-byte units
// a rank's slot-map block: [count | overflow lanes | slot of UAV 0 .. n_max-1 | pad to 16 B | box of this search: 6 doubles]
inline int64_t map_boxw(const mrs_swarm* s) { return (s->comm_n_max + 2 + 3) & ~(int64_t)3; }
inline int64_t map_stride(const mrs_swarm* s) { return map_boxw(s) + 12; }
// a halo block never exceeds the block of a full gather (the peer windows are sized by that): 64 B x (1 + cap) <= 48 B x n_max
inline int64_t halo_cap_max(const mrs_swarm* s) { return (int64_t)(sizeof(PosRecord) * (size_t)s->comm_n_max / sizeof(HaloEntry)) - 1; }
// will the next search of the export-set exchange run on a halo exchange?  (the same answer on every rank: all of it follows from collective calls)
inline bool halo_next(const mrs_swarm* s) { return s->halo_enabled && s->halo_ok && s->comm_world > 1 && s->halo_backoff == 0 && s->halo_cap >= 1 && s->halo_cap <= halo_cap_max(s); }
}  // namespace mrs_host
using namespace mrs_host;
