// transport_peer.hip — the peer-window exchange: the collectives of the sharded tick as direct writes into the peers' device memory
// over xGMI (collide.hip k_peer_allgather), no collective library and no host in the tick.  DESIGN §5.
#include "host_internal.h"

namespace mrs_host {
// the peer-window exchange: every rank issues the same collectives in the same order, so the sequence number is the same everywhere
int peer_allgather(mrs_swarm* s, const void* send, void* recv, size_t bytes) {
  if (bytes > s->peer_slot_bytes) return fail(MRS_ERR_ARG, "peer-window exchange: a block of " + std::to_string(bytes) + " bytes does not fit the window's slots (" + std::to_string(s->peer_slot_bytes) + ")");
  unsigned taken = 0;
  HIPCHK(mrs_launch_peer_allgather(&s->peer_windows, send, recv, bytes, s->comm_rank, s->comm_world, ++s->peer_seq, s->peer_slot_bytes, s->peer_ticket,
                                   s->peer_tickets, s->peer_err, &taken, s->cstream));
  s->peer_tickets += taken;
  return MRS_OK;
}

int peer_failed(mrs_swarm* s) {
  volatile unsigned* e = s->peer_err;
  return fail(MRS_ERR_HIP, "peer-window exchange: rank " + std::to_string(s->comm_rank) + " waited in vain for the block of rank " + std::to_string(e[2]) +
                               " in collective " + std::to_string(e[1]) + " (that rank's flag says " + std::to_string(e[3]) + "; this rank has issued " +
                               std::to_string(s->peer_seq) + " collectives, " + std::to_string(s->x_ticks) + " ticks, " + std::to_string(s->x_searches) +
                               " searches) — the results of this call are not valid and the windows are dead: use fresh processes");
}

void peer_release(mrs_swarm* s) {
  for (void* p : s->peer_opened) (void)hipIpcCloseMemHandle(p);
  s->peer_opened.clear();
  if (s->peer_window) (void)hipFree(s->peer_window);
  if (s->peer_ticket) (void)hipFree(s->peer_ticket);
  if (s->peer_err) (void)hipHostFree(s->peer_err);
  s->peer_window = nullptr;
  s->peer_ticket = s->peer_err = nullptr;
  s->peer_world  = 0;
  s->comm_peer   = false;
}
}  // namespace mrs_host

extern "C" {

int mrs_swarm_peer_window_create(mrs_swarm_t* s, int32_t world, int32_t rank, int64_t n_total, void** window, uint8_t* ipc_handle64) {
  MRS_ENTER(s);
  if (!s || world > MRS_MAX_PEERS) return fail(MRS_ERR_ARG, "bad peer-window arguments (at most 64 ranks)");
  int rc = comm_setup(s, world, rank, n_total);
  if (rc) return rc;
  if (s->peer_window) return fail(MRS_ERR_ARG, "this swarm already has a peer window");
  HIPCHK(hipSetDevice(s->device));
  // the largest block any collective of the sharded tick sends: the full gather of a search (one record per UAV of the largest shard)
  const int64_t n_max = (n_total + world - 1) / world > 0 ? (n_total + world - 1) / world : 1;
  size_t slot = sizeof(PosRecord) * (size_t)n_max;
  if (slot < sizeof(uint32_t) * (size_t)(n_max + 20)) slot = sizeof(uint32_t) * (size_t)(n_max + 20);  // (the slot maps: n_max + 2 words padded to 16-byte units + the search box, map_stride)
  // (an export block is header + capacity records of 32 B, the capacity up to 1.5 x the largest export set + 127: export_search)
  if (slot < sizeof(Pos4) * ((size_t)n_max + (size_t)n_max / 2 + 129)) slot = sizeof(Pos4) * ((size_t)n_max + (size_t)n_max / 2 + 129);
  slot = (slot + 255) / 256 * 256;
  s->peer_slot_bytes   = slot;
  s->peer_window_bytes = 4096 + 2 * (size_t)world * slot;
  // Written by other devices WHILE kernels of this one poll and read it: uncached device memory ("extended-scope fine-grained" — on
  // this GPU family plain fine-grained memory is only guaranteed coherent across devices at kernel boundaries, and the flags are
  // polled inside a kernel; collective libraries allocate their flag and staging memory the same way).  MRS_PEER_WINDOW_MEMORY =
  // finegrained | coarse for runtimes that cannot export an uncached allocation (every access of the exchange kernel is
  // system-scope either way).
  struct Release {  // any failure below gives the window, the ticket words and the pinned error word back
    mrs_swarm* p;
    ~Release() {
      if (p) peer_release(p);
    }
  } release{s};
  const char* kind = getenv("MRS_PEER_WINDOW_MEMORY");
  if (kind && strcmp(kind, "coarse") == 0)
    HIPCHK(hipMalloc(&s->peer_window, s->peer_window_bytes));
  else
    HIPCHK(hipExtMallocWithFlags(&s->peer_window, s->peer_window_bytes, kind && strcmp(kind, "finegrained") == 0 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
  HIPCHK(hipMalloc((void**)&s->peer_ticket, sizeof(unsigned) * (MRS_MAX_PEERS + 1)));  // (+ the give-up mark the exchange kernels read)
  HIPCHK(hipHostMalloc((void**)&s->peer_err, 64, hipHostMallocMapped));
  *s->peer_err = 0u;
  HIPCHK(hipMemsetAsync(s->peer_window, 0, 4096, s->stream));  // flags: no collective has happened
  HIPCHK(hipMemsetAsync(s->peer_ticket, 0, sizeof(unsigned) * (MRS_MAX_PEERS + 1), s->stream));
  HIPCHK(hipStreamSynchronize(s->stream));  // ... before any peer can learn the address
  if (ipc_handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C ABI carries IPC handles as 64 bytes");
    hipIpcMemHandle_t h;
    const hipError_t  e = hipIpcGetMemHandle(&h, s->peer_window);
    if (e != hipSuccess) return fail(MRS_ERR_HIP, std::string("peer window: hipIpcGetMemHandle: ") + hipGetErrorString(e));
    memcpy(ipc_handle64, &h, 64);
  }
  release.p = nullptr;
  if (window) *window = s->peer_window;
  s->peer_world   = world;
  s->peer_rank    = rank;
  s->peer_n_total = n_total;
  return MRS_OK;
}

int mrs_swarm_comm_init_peer(mrs_swarm_t* s, void* const* windows, const uint8_t* ipc_handles) {
  MRS_ENTER(s);
  if (!s || s->peer_world == 0) return fail(MRS_ERR_ARG, "mrs_swarm_peer_window_create has not been called");
  if (!windows && !ipc_handles) return fail(MRS_ERR_ARG, "the peers' windows are needed as pointers or as IPC handles");
  if (s->comm_world > 0) return fail(MRS_ERR_ARG, "communicator already initialised");
  HIPCHK(hipSetDevice(s->device));
  for (int q = 0; q < s->peer_world; q++) {
    void* p = nullptr;
    if (q == s->peer_rank) {
      p = s->peer_window;
    } else if (windows && windows[q]) {
      p = windows[q];
    } else if (ipc_handles) {
      hipIpcMemHandle_t h;
      memcpy(&h, ipc_handles + 64 * (size_t)q, 64);
      const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) return fail(MRS_ERR_HIP, "peer window of rank " + std::to_string(q) + ": hipIpcOpenMemHandle: " + hipGetErrorString(e));
      s->peer_opened.push_back(p);
    } else {
      return fail(MRS_ERR_ARG, "no window given for rank " + std::to_string(q));
    }
    s->peer_windows.win[q] = p;
  }
  s->comm_peer = true;
  s->peer_seq = s->peer_tickets = 0u;
  const int rc = comm_buffers(s, s->peer_world, s->peer_rank, s->peer_n_total);
  if (rc) s->comm_peer = false;
  return rc;
}

}  // extern "C"
