// collide.hip — mutual-collision pass: GPU uniform-grid spatial hash replacing the reference's per-tick
// nanoflann kd-tree (src/multirotor_simulator.cpp:295-359 + include/nanoflann.hpp radius search).
//
// Semantics reproduced exactly:
//   neighbour set of i  = { j : d2(i,j) < 3.0 }  with d2 = ((0 + dx^2) + dy^2) + dz^2, dx = x_i - x_j
//                         (nanoflann L2_Adaptor::evalMetric, RadiusResultSet(3.0): SQUARED distance vs 3.0)
//   for j != i with d2 < crit(i,j) = ((arm_i + prop_i) + arm_j) + prop_j   (squared metres vs metres — kept)
//     crash mode : j is crashed                       (:348)
//     otherwise  : F_i += ((rebounce * normalized(x_i - x_j)) * m_i) * (m_j / (m_i + m_j))   (:350)
//   applyForce(F_i) for every i, zero included         (:356-358)
// Only the neighbour *set* of the kd-tree matters; forces are summed in ascending partner index (the kd-tree's
// traversal order is not reproducible by any other structure; with <= 1 partner, the usual case, the sum is exact).
//
// Pipeline per tick (all on the swarm's stream, HBM-bound integer/index work) — two kernels:
//   insert : SoA state + type table -> 48-B PosRecord per UAV (the multi-GPU all-gather payload; gathered records skip the
//            packing); cell = floor(pos / 1.75 m) (> sqrt(3), so partners sit in the 27 adjacent cells);
//            bucket = hash(cell) & (T-1); the UAV becomes the head of its bucket's chain with ONE 64-bit atomic exchange and
//            keeps the previous head as its `next` link.  No counting pass, no prefix sum, no scatter.
//   query  : one single-wave workgroup per 64 local UAVs fetches the 27 bucket heads per lane, follows the (rare) chains and
//            evaluates the literal predicate; partners are consumed in ascending index, which makes the result
//            independent of the atomic arrival order.  It also clears the OTHER head table for the next tick.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "swarm_layout.h"

namespace {

constexpr double INV_CELL  = 1.0 / 1.75; // cell edge 1.75 m > sqrt(3.0) = 1.7320508; // cells are floor(pos * INV_CELL): any consistent assignment with edge > sqrt(3) works,
                                         // and the multiply avoids three ~70-cycle IEEE divisions per cell_of
constexpr double POS_LIMIT = 1.0e9;      // |coordinate| beyond this (or non-finite) never collides here

struct Cell { int x, y, z; bool ok; };

__device__ __forceinline__ Cell cell_of(double x, double y, double z) {
  Cell c;
  c.ok = (fabs(x) < POS_LIMIT) && (fabs(y) < POS_LIMIT) && (fabs(z) < POS_LIMIT);  // false for NaN/inf
  c.x  = c.ok ? (int)floor(x * INV_CELL) : 0;
  c.y  = c.ok ? (int)floor(y * INV_CELL) : 0;
  c.z  = c.ok ? (int)floor(z * INV_CELL) : 0;
  return c;
}

// the column (cx, cy) is hashed, cz is added: the three z-neighbours of a cell sit in consecutive buckets, so a UAV's 27
// probes touch ~9 cache lines of the descriptor table instead of 27
__device__ __forceinline__ uint32_t bucket_of(int cx, int cy, int cz, uint32_t mask) {
  uint32_t h = ((uint32_t)cx * 73856093u) ^ ((uint32_t)cy * 19349663u);
  h ^= h >> 15;
  h *= 0x2c1b3c6du;
  h ^= h >> 12;
  return (h + (uint32_t)cz) & mask;
}

__device__ __forceinline__ uint32_t cell_tag(int cx, int cy, int cz) {
  uint32_t h = (uint32_t)cx * 0x9E3779B1u + (uint32_t)cy * 0x85EBCA77u + (uint32_t)cz * 0xC2B2AE3Du;
  h ^= h >> 16;
  return h * 0x27D4EB2Fu;
}

__global__ void k_flags_update(uint32_t* F, int first, int count, uint32_t and_mask, uint32_t or_mask) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < count) F[first + k] = (F[first + k] & and_mask) | or_mask;
}

__global__ void k_pack_positions(SwarmDev sw, PosRecord* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sw.n) return;
  const TypeParams& P = sw.T[sw.F[i] >> FLAG_TYPE_SHIFT];
  PosRecord r;
  r.x = sw.S[(size_t)(F_X + 0) * sw.npad + i];
  r.y = sw.S[(size_t)(F_X + 1) * sw.npad + i];
  r.z = sw.S[(size_t)(F_X + 2) * sw.npad + i];
  r.mass        = P.mass;
  r.arm_length  = P.arm_length;
  r.prop_radius = P.prop_radius;
  out[i] = r;
}

// Head / link word: x = UAV index + 1 (0 = empty / end of chain), y = cell tag (31 bits) | CHAIN bit.
// The tag is a second, independent hash of the exact cell: it lets the query discard members of OTHER cells that share a
// bucket (about 5 probes per UAV at load factor 0.25) without fetching their 48-B records.  CHAIN on a bucket head says
// "more than one member": a single-member bucket is accepted or dropped from its head word alone.
constexpr uint32_t CHAIN_BIT = 0x80000000u;

__device__ __forceinline__ void insert_uav(long long j, const Cell& c, uint32_t mask, uint2* head, uint2* next) {
  if (!c.ok) {  // never inserted
    next[j] = make_uint2(0u, 0u);
    return;
  }
  const uint32_t b = bucket_of(c.x, c.y, c.z, mask);
  const unsigned long long me = (unsigned long long)((uint32_t)j + 1u) | ((unsigned long long)(cell_tag(c.x, c.y, c.z) & ~CHAIN_BIT) << 32);
  unsigned long long* slot = reinterpret_cast<unsigned long long*>(head + b);
  const unsigned long long old = atomicExch(slot, me);
  next[j] = make_uint2((uint32_t)old, (uint32_t)(old >> 32) & ~CHAIN_BIT);
  // Whoever finds the bucket occupied marks it as a chain.  The mark is only ever set, and the thread whose exchange comes
  // last is not the first member, so it sets the bit after its own exchange: the final head carries it iff members > 1.
  if ((uint32_t)old != 0u) atomicOr(slot, (unsigned long long)CHAIN_BIT << 32);
}

__global__ void k_insert(const PosRecord* rec, long long n_total, uint32_t mask, uint2* head, uint2* next) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_total) return;
  insert_uav(j, cell_of(rec[j].x, rec[j].y, rec[j].z), mask, head, next);
}

// single-GPU tick: pack and insert in one pass over the state (the records are still written: the query reads them)
__global__ void k_pack_insert(SwarmDev sw, PosRecord* rec, uint32_t mask, uint2* head, uint2* next) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sw.n) return;
  const TypeParams& P = sw.T[sw.F[i] >> FLAG_TYPE_SHIFT];
  PosRecord r;
  r.x = sw.S[(size_t)(F_X + 0) * sw.npad + i];
  r.y = sw.S[(size_t)(F_X + 1) * sw.npad + i];
  r.z = sw.S[(size_t)(F_X + 2) * sw.npad + i];
  r.mass        = P.mass;
  r.arm_length  = P.arm_length;
  r.prop_radius = P.prop_radius;
  rec[i] = r;
  insert_uav(i, cell_of(r.x, r.y, r.z), mask, head, next);
}

// ---- query ----
// accumulate one partner into the force / crash state of a UAV (literal predicate and force expression)
__device__ __forceinline__ void apply_partner(const PosRecord& me, const PosRecord& o, int crash, double rebounce, double& fx, double& fy,
                                              double& fz, bool& crashed) {
  const double d0 = me.x - o.x, d1 = me.y - o.y, d2 = me.z - o.z;
  const double dist    = ((0.0 + d0 * d0) + d1 * d1) + d2 * d2;
  const double crit_ij = ((me.arm_length + me.prop_radius) + o.arm_length) + o.prop_radius;
  const double crit_ji = ((o.arm_length + o.prop_radius) + me.arm_length) + me.prop_radius;
  if (crash) {
    // the reference crashes the partner of every qualifying ordered pair (i -> idx); seen from the partner's side:
    // this UAV is crashed iff some j has it as a qualifying partner, i.e. dist < crit(j, i)
    if (dist < crit_ji) crashed = true;
  } else if (dist < crit_ij) {
    double r0 = d0, r1 = d1, r2 = d2;
    const double z = (r0 * r0 + r1 * r1) + r2 * r2;  // Eigen normalized()
    if (z > 0) {
      const double nn = sqrt(z);
      r0 /= nn; r1 /= nn; r2 /= nn;
    }
    const double ratio = o.mass / (me.mass + o.mass);
    fx += ((rebounce * r0) * me.mass) * ratio;
    fy += ((rebounce * r1) * me.mass) * ratio;
    fz += ((rebounce * r2) * me.mass) * ratio;
  }
}

__device__ __forceinline__ bool qualifies(const PosRecord& me, const PosRecord& o, int crash) {
  const double d0 = me.x - o.x, d1 = me.y - o.y, d2 = me.z - o.z;
  const double dist = ((0.0 + d0 * d0) + d1 * d1) + d2 * d2;
  if (!(dist < 3.0)) return false;  // nanoflann RadiusResultSet(3.0) on the squared distance
  const double crit_ij = ((me.arm_length + me.prop_radius) + o.arm_length) + o.prop_radius;
  const double crit_ji = ((o.arm_length + o.prop_radius) + me.arm_length) + me.prop_radius;
  return dist < crit_ij || (crash && dist < crit_ji);
}

// reference path for one lane: repeated sweeps over the 27 bucket chains, each returning the smallest qualifying partner
// index above the previous one (ascending-index accumulation without per-lane arrays).  Correct for any bucket
// occupancy; used when the wave-cooperative path below overflows its LDS lists.
__device__ void query_lane_sweeps(const PosRecord& me, const Cell& c, long long gi, const PosRecord* rec, long long n_total, uint32_t mask,
                                  const uint2* head, const uint2* next, int crash, double rebounce, double& fx, double& fy, double& fz,
                                  bool& crashed) {
  long long prev = -1;
  for (;;) {
    long long best = n_total;
    for (int q = 0; q < 27; q++) {
      const int      cx = c.x + q / 9 - 1, cy = c.y + (q / 3) % 3 - 1, cz = c.z + q % 3 - 1;
      const uint32_t tg = cell_tag(cx, cy, cz) & ~CHAIN_BIT;
      for (uint2 e = head[bucket_of(cx, cy, cz, mask)]; e.x != 0u; e = next[e.x - 1u]) {
        const long long j = (long long)e.x - 1;
        if ((e.y & ~CHAIN_BIT) != tg || j <= prev || j >= best || j == gi) continue;
        const PosRecord o  = rec[j];
        const Cell      oc = cell_of(o.x, o.y, o.z);
        if (oc.x != cx || oc.y != cy || oc.z != cz) continue;  // another cell sharing the bucket (and the tag)
        if (qualifies(me, o, crash)) best = j;
      }
    }
    if (best >= n_total) break;
    apply_partner(me, rec[best], crash, rebounce, fx, fy, fz, crashed);
    prev = best;
  }
}

// Wave-cooperative query, one single-wave workgroup per 64 local UAVs.
//   A  every lane fetches its 27 bucket heads (independent loads, one memory round trip); a head whose tag is not the probed
//      cell's and that has no chain is dropped on the spot (~2 candidates per UAV remain on a 64 m^3/UAV swarm)
//   B  the (owner lane, probed cell, candidate) triples of the whole wave are compacted into an LDS list (wave prefix sum)
//   C  the list is processed 64 x U entries at a time with uniform control flow — this is what removes the 27-way divergent
//      walk in which some lane always had a non-empty bucket and every iteration paid a full memory latency.  An entry
//      of a chained bucket also fetches its `next` link and appends it to the list; qualifying partners go to a small
//      per-owner hit list (LDS atomics)
//   D  owners order their (rare) hits by index and accumulate; overflow of either list falls back to query_lane_sweeps
constexpr int      PAIR_CAP  = 1024;
constexpr int      HIT_CAP   = 6;
constexpr uint32_t META_WALK = 0x10000u;  // pair meta: owner lane | probed cell q << 8 | WALK (follow the `next` link)

__global__ void __launch_bounds__(64) k_query(SwarmDev sw, const PosRecord* rec, long long n_total, long long my_offset, uint32_t mask,
                                              const uint2* head, const uint2* next, uint2* head_to_clear, uint32_t table_size, int crash,
                                              double rebounce) {
  __shared__ PosRecord me_s[64];
  __shared__ int4      me_cell[64];
  __shared__ uint2     pair_e[PAIR_CAP];   // x: candidate index + 1, y: its tag
  __shared__ uint32_t  pair_m[PAIR_CAP];   // meta
  __shared__ uint32_t  hit_j[64][HIT_CAP];
  __shared__ uint32_t  hit_n[64];
  __shared__ uint32_t  list_total, list_overflow, hit_overflow;  // two flags: each is only ever set, never downgraded

  const int       lane   = threadIdx.x;
  const int       i      = blockIdx.x * 64 + lane;
  const bool      active = i < sw.n;
  const long long gi     = my_offset + i;
  PosRecord       me;
  me.x = me.y = me.z = __longlong_as_double(0x7ff8000000000000ll);
  me.mass = me.arm_length = me.prop_radius = 0.0;
  if (active) me = rec[gi];
  const Cell c = cell_of(me.x, me.y, me.z);
  me_s[lane]   = me;
  me_cell[lane] = make_int4(c.x, c.y, c.z, 0);
  hit_n[lane]  = 0;
  if (lane == 0) list_overflow = hit_overflow = 0;

  // A: bucket heads.  Unconditional loads from always-valid addresses: a load under a divergent branch is waited for at
  // the join, which would serialise 27 memory round trips.
  uint2    info[27];
  uint32_t tc = 0;
#pragma unroll
  for (int q = 0; q < 27; q++) info[q] = head[bucket_of(c.x + q / 9 - 1, c.y + (q / 3) % 3 - 1, c.z + q % 3 - 1, mask)];
  // this tick's table has been read: wipe the other one for the next tick (grid-strided, coalesced)
  {
    const uint32_t stride = gridDim.x * 64u;
    for (uint32_t t = blockIdx.x * 64u + lane; t < table_size; t += stride) head_to_clear[t] = make_uint2(0u, 0u);
  }
#pragma unroll
  for (int q = 0; q < 27; q++) {
    const uint32_t tg    = cell_tag(c.x + q / 9 - 1, c.y + (q / 3) % 3 - 1, c.z + q % 3 - 1) & ~CHAIN_BIT;
    const bool     chain = (info[q].y & CHAIN_BIT) != 0u;
    const bool     take  = c.ok && info[q].x != 0u && (chain || info[q].y == tg);
    info[q].y = (info[q].y & ~CHAIN_BIT);
    if (!take) info[q].x = 0u;
    // .x != 0: entry goes to the list;  keep the chain flag in the top bit of .y again for phase B
    if (take && chain) info[q].y |= CHAIN_BIT;
    tc += take ? 1u : 0u;
  }
#if defined(MRS_QUERY_STOP) && MRS_QUERY_STOP == 1
  if (tc == 0xFFFFFFFFu) sw.F[i] = tc;
  return;
#endif
  // B: wave prefix sum -> slots in the pair list
  uint32_t inc = tc;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(inc, off, 64);
    if (lane >= off) inc += o;
  }
  if (lane == 63) list_total = inc;
  uint32_t slot = inc - tc;
#pragma unroll
  for (int q = 0; q < 27; q++) {
    if (info[q].x != 0u) {
      if (slot < (uint32_t)PAIR_CAP) {  // 64 x 27 heads can exceed the list: the total is checked below
        pair_e[slot] = make_uint2(info[q].x, info[q].y & ~CHAIN_BIT);
        pair_m[slot] = (uint32_t)lane | ((uint32_t)q << 8) | ((info[q].y & CHAIN_BIT) ? META_WALK : 0u);
      }
      slot++;
    }
  }
  __syncthreads();
#if defined(MRS_QUERY_STOP) && MRS_QUERY_STOP == 2
  if (pair_e[lane].x == 0xFFFFFFFFu) sw.F[i] = 1;
  return;
#endif
  // C: uniform sweep over the list, U independent entries per lane and iteration so that their loads overlap
  constexpr int U = 4;
  uint32_t total = list_total;
  if (total > PAIR_CAP) {
    list_overflow = 1;
    total         = 0;
  }
  for (uint32_t base = 0; base < total;) {
    uint2    pe[U], nx[U];
    uint32_t pm[U];
    bool     live[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t p = base + u * 64 + lane;
      pe[u] = pair_e[p < total ? p : 0u];
      pm[u] = p < total ? pair_m[p] : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; u++) nx[u] = (pm[u] & META_WALK) ? next[pe[u].x - 1u] : make_uint2(0u, 0u);
    // tag filter: only members of exactly the probed cell survive (false positives of the 31-bit tag are caught by the exact
    // cell comparison below)
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t p = base + u * 64 + lane;
      live[u] = false;
      if (p < total) {
        const int  ow = (int)(pm[u] & 0xFFu), q = (int)((pm[u] >> 8) & 0xFFu);
        const int4 mc = me_cell[ow];
        live[u] = pe[u].y == (cell_tag(mc.x + q / 9 - 1, mc.y + (q / 3) % 3 - 1, mc.z + q % 3 - 1) & ~CHAIN_BIT) &&
                  (long long)pe[u].x - 1 != my_offset + blockIdx.x * 64 + ow;  // idx == i, src/multirotor_simulator.cpp:335
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (nx[u].x != 0u) {  // the chain goes on: its next member joins the list
        const uint32_t k = atomicAdd(&list_total, 1u);
        if (k < PAIR_CAP) {
          pair_e[k] = make_uint2(nx[u].x, nx[u].y & ~CHAIN_BIT);
          pair_m[k] = pm[u];
        } else {
          list_overflow = 1;
        }
      }
      if (!live[u]) continue;
      const int       ow = (int)(pm[u] & 0xFFu), q = (int)((pm[u] >> 8) & 0xFFu);
      const int4      mc = me_cell[ow];
      const PosRecord o  = rec[pe[u].x - 1u];
      const Cell      oc = cell_of(o.x, o.y, o.z);
      if (oc.x != mc.x + q / 9 - 1 || oc.y != mc.y + (q / 3) % 3 - 1 || oc.z != mc.z + q % 3 - 1) continue;  // tag collision
      const PosRecord m = me_s[ow];
      if (!qualifies(m, o, crash)) continue;
      const uint32_t k = atomicAdd(&hit_n[ow], 1u);
      if (k < HIT_CAP)
        hit_j[ow][k] = pe[u].x - 1u;
      else
        hit_overflow = 1;
    }
    base = (base + 64 * U < total) ? base + 64 * U : total;  // entries appended meanwhile start at the old total
    __syncthreads();  // appended entries and the new total are visible
    total = list_total;
    if (list_overflow) total = 0;
    __syncthreads();  // nobody appends before everybody has read the total
  }
  __syncthreads();
#if defined(MRS_QUERY_STOP) && MRS_QUERY_STOP == 3
  if (hit_n[lane] == 0xFFFFFFFFu) sw.F[i] = 1;
  return;
#endif
  // D: owners accumulate their hits in ascending partner index
  double fx = 0.0, fy = 0.0, fz = 0.0;
  bool   crashed = false;
  if (active && c.ok) {
    if (list_overflow || (hit_overflow && hit_n[lane] > HIT_CAP)) {  // dense neighbourhood: the reference path
      query_lane_sweeps(me, c, gi, rec, n_total, mask, head, next, crash, rebounce, fx, fy, fz, crashed);
    } else {
      const uint32_t nh = hit_n[lane];
      uint32_t       prev = 0;
      bool           have_prev = false;
      for (uint32_t r = 0; r < nh; r++) {  // selection by repeated minimum: nh <= 6, almost always 0 or 1
        uint32_t best = 0xFFFFFFFFu;
        for (uint32_t k = 0; k < nh; k++) {
          const uint32_t j = hit_j[lane][k];
          if ((!have_prev || j > prev) && j < best) best = j;
        }
        apply_partner(me, rec[best], crash, rebounce, fx, fy, fz, crashed);
        prev = best;
        have_prev = true;
      }
    }
  }
  if (active) {
    sw.S[(size_t)(F_FEXT + 0) * sw.npad + i] = fx;
    sw.S[(size_t)(F_FEXT + 1) * sw.npad + i] = fy;
    sw.S[(size_t)(F_FEXT + 2) * sw.npad + i] = fz;
    if (crashed) sw.F[i] |= FLAG_CRASHED;
  }
}

}  // namespace

struct CollideWork {
  long long cap_n = 0;
  uint32_t  cap_T = 0;
  int       cur   = 0;  // which head table the next tick fills; the other one is being wiped by that tick's query
  uint2 *   head[2] = {nullptr, nullptr}, *next = nullptr;
};

static void free_work(CollideWork* w) {
  (void)hipFree(w->head[0]); (void)hipFree(w->head[1]); (void)hipFree(w->next);
  w->head[0] = w->head[1] = w->next = nullptr;
}

extern "C" void mrs_collide_free(CollideWork* w) {
  if (!w) return;
  free_work(w);
  delete w;
}

extern "C" hipError_t mrs_launch_flags_update(uint32_t* F, int first, int count, uint32_t and_mask, uint32_t or_mask, hipStream_t st) {
  hipLaunchKernelGGL(k_flags_update, dim3((count + 255) / 256), dim3(256), 0, st, F, first, count, and_mask, or_mask);
  return hipGetLastError();
}

extern "C" hipError_t mrs_launch_pack_positions(SwarmDev sw, PosRecord* out, hipStream_t st) {
  if (sw.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_pack_positions, dim3((sw.n + 255) / 256), dim3(256), 0, st, sw, out);
  return hipGetLastError();
}

#define CK(e)                        \
  do {                               \
    hipError_t _e = (e);             \
    if (_e != hipSuccess) return _e; \
  } while (0)

// rec_is_local_scratch: `rec` is this swarm's own (n_total == sw.n) record buffer that has NOT been packed yet — pack and hash
// are then fused; otherwise `rec` holds ready (gathered) records
extern "C" hipError_t mrs_collide_run(SwarmDev sw, CollideWork** work, const PosRecord* rec, long long n_total, long long my_offset,
                                      int crash, double rebounce, int rec_is_local_scratch, hipStream_t st) {
  if (!*work) *work = new CollideWork();
  CollideWork* w = *work;
  uint32_t     T = 1024;
  while ((long long)T < 4 * n_total) T <<= 1;  // load factor <= 0.25: ~27*0.25 false candidates per UAV
  if (n_total > w->cap_n || T > w->cap_T) {
    CK(hipStreamSynchronize(st));
    free_work(w);
    CK(hipMalloc(&w->head[0], sizeof(uint2) * (size_t)T));
    CK(hipMalloc(&w->head[1], sizeof(uint2) * (size_t)T));
    CK(hipMalloc(&w->next, sizeof(uint2) * (size_t)n_total));
    CK(hipMemsetAsync(w->head[0], 0, sizeof(uint2) * (size_t)T, st));  // afterwards every query wipes the table of the next tick
    CK(hipMemsetAsync(w->head[1], 0, sizeof(uint2) * (size_t)T, st));
    w->cap_n = n_total;
    w->cap_T = T;
    w->cur   = 0;
  }
  T = w->cap_T;  // a larger table from an earlier call is still valid (both tables are empty between ticks)
  const uint32_t mask = T - 1;
  const unsigned gN   = (unsigned)((n_total + 255) / 256);
  uint2*         head = w->head[w->cur];
  uint2*         other = w->head[w->cur ^ 1];
  w->cur ^= 1;
  if (rec_is_local_scratch)
    hipLaunchKernelGGL(k_pack_insert, dim3(gN), dim3(256), 0, st, sw, const_cast<PosRecord*>(rec), mask, head, w->next);
  else
    hipLaunchKernelGGL(k_insert, dim3(gN), dim3(256), 0, st, rec, n_total, mask, head, w->next);
  // `other` was wiped by the previous query except for what that query's own tick left in it: nothing — it is the table of
  // two ticks ago, wiped one tick ago.  This tick's table is wiped by the NEXT query; the very first tick starts from memset.
  hipLaunchKernelGGL(k_query, dim3((sw.n + 63) / 64), dim3(64), 0, st, sw, rec, n_total, my_offset, mask, head, w->next, other, T, crash,
                     rebounce);
  return hipGetLastError();
}
