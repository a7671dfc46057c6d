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
//            independent of the atomic arrival order.  The head table a later search will fill is wiped by the tick before it.
//
// Neighbour lists (single-GPU ticks): UAVs move centimetres per tick, so the partner search above is only REPEATED when it
// has to be.  A rebuild tick runs insert + the list-building query (k_query2: several lanes per UAV, table entries that carry the
// UAV's position inside its cell — see there) with 2.25-m cells and keeps, per UAV, the ascending list of every UAV
// within sqrt(3) + SKIN of it, together with the positions at that moment.  The step kernel compares every new position
// with the recorded one (step_device.inc) and raises a flag once any UAV has moved more than SKIN/2; until then a pair
// closer than sqrt(3) now was closer than sqrt(3) + SKIN at the rebuild, i.e. is in the lists, and a tick is ONE cheap
// pass: current positions of the listed UAVs -> literal predicate -> forces / crash flags, in the same ascending order.
// The decision is taken on the device (both kernels are always launched: on a list tick the insert kernel evaluates the lists
// and the query kernel returns at once), host
// writes to positions or airframe constants force a rebuild, and a UAV with more than LIST_CAP listed neighbours keeps
// the pass in rebuild mode.  Results are identical to searching every tick.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "swarm_layout.h"
#include "collide_device.inc"

namespace {

// cells are floor(pos * INV_CELL): any consistent assignment with an edge above the search radius works, and the multiply
// avoids three ~70-cycle IEEE divisions per cell_of
constexpr double INV_CELL      = 1.0 / 1.75;  // plain search: edge 1.75 m > sqrt(3.0) = 1.7320508
#ifndef MRS_SKIN
#define MRS_SKIN 0.5  // (compile-time so that the cell arithmetic stays in constants; tools/variant_bench.sh sweeps it)
#endif
constexpr double SKIN          = MRS_SKIN;    // neighbour lists: how far apart beyond sqrt(3) a listed pair may be
constexpr double SQRT3_UP      = 1.7320508075688775;                      // >= sqrt(3)
constexpr double INV_CELL_WIDE = 1.0 / (SQRT3_UP + SKIN + 0.0179491924);  // list rebuild: edge 2.25 m > sqrt(3) + SKIN = 2.2320508 (SKIN 0.5)
constexpr double LIST_R2       = (SQRT3_UP + SKIN) * (SQRT3_UP + SKIN) * (1.0 + 1e-9) + 1e-5;  // > (sqrt(3) + SKIN)^2 (4.98206 at SKIN 0.5)
// Sharded swarms keep their lists longer: a search there is two collectives, a host synchronisation and a dozen launches (~200 us
// against 37 us on one GPU), so the wider skin — half as many searches, 1.3 instead of 0.7 listed partners per UAV at 64 m^3 — pays
// (one GPU, SKIN swept: 0.5 m 22.9 us per tick, 1.0 m 23.6).  Mode 2 of the WIDE / LISTS template arguments below.
#ifndef MRS_SKIN_SHARDED
#define MRS_SKIN_SHARDED 1.0
#endif
constexpr double SKIN2          = MRS_SKIN_SHARDED;
constexpr double INV_CELL_WIDE2 = 1.0 / (SQRT3_UP + SKIN2 + 0.0179491924);
constexpr double LIST_R2_2      = (SQRT3_UP + SKIN2) * (SQRT3_UP + SKIN2) * (1.0 + 1e-9) + 1e-5;
constexpr double POS_LIMIT     = MRS_POS_LIMIT;  // |coordinate| beyond this (or non-finite) never collides here
// fused evaluation: a UAV beyond this fraction of the distance that invalidates the lists makes the host queue the next search in
// stream order (no stall, no replay); the remaining 25 % (6 cm) are ten ticks at 6 m/s — more than the host runs ahead of the device
static const double WARN_FRACTION = getenv("MRS_WARN_FRACTION") ? atof(getenv("MRS_WARN_FRACTION")) : 0.75;
constexpr int    LIST_CAP      = 24;          // listed neighbours per UAV (0.7 expected at 64 m^3 per UAV, 4.6 at 10 m^3: P(> 24) ~ 1e-11;
                                              // with 8, one UAV in 10^4 overflowed at 30 m^3 per UAV and kept a 100 k swarm searching)

struct Cell { int x, y, z; bool ok; };

template <int WIDE>  // 0: plain search cells, 1: list cells of one GPU, 2: list cells of a sharded swarm
__device__ __forceinline__ Cell cell_of(double x, double y, double z) {
  constexpr double ic = WIDE == 2 ? INV_CELL_WIDE2 : (WIDE ? INV_CELL_WIDE : INV_CELL);
  Cell c;
  c.ok = (fabs(x) < POS_LIMIT) && (fabs(y) < POS_LIMIT) && (fabs(z) < POS_LIMIT);  // false for NaN/inf
  c.x  = c.ok ? (int)floor(x * ic) : 0;
  c.y  = c.ok ? (int)floor(y * ic) : 0;
  c.z  = c.ok ? (int)floor(z * ic) : 0;
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

// ---- entry format of the LIST-BUILDING searches (LISTS = 1, 2; the plain search above keeps the format above) ----
//   x = (UAV index + 1) in the low `ib` bits (2^ib > n_total) | the top 32 - ib bits of the cell tag
//   y = the UAV's position INSIDE its cell in 1/1024 of the cell edge, 10 bits per axis | CHAIN (bit 31)
// A prober decides from the entry alone whether the member can be within the list radius (integer arithmetic on cell offsets and
// the 10-bit coordinates, a bound that errs on the safe side by the quantisation step), and fetches the 48-B record only of those
// that can: 0.7 record fetches per UAV at 64 m^3 per UAV instead of one per member of every probed cell (4.4).
constexpr uint32_t QBITS = 10, QONE = 1u << QBITS, QMASK = QONE - 1u;

template <int WIDE>  // 1: list cells of one GPU, 2: list cells of a sharded swarm.  Same cell as cell_of<WIDE>; q = 0 for unusable positions
__device__ __forceinline__ Cell cell_q(double x, double y, double z, uint32_t& q) {
  constexpr double ic = WIDE == 2 ? INV_CELL_WIDE2 : INV_CELL_WIDE;
  Cell c;
  c.ok = (fabs(x) < POS_LIMIT) && (fabs(y) < POS_LIMIT) && (fabs(z) < POS_LIMIT);  // false for NaN/inf
  const double tx = x * ic, ty = y * ic, tz = z * ic;
  const double fx = floor(tx), fy = floor(ty), fz = floor(tz);
  c.x = c.ok ? (int)fx : 0;
  c.y = c.ok ? (int)fy : 0;
  c.z = c.ok ? (int)fz : 0;
  // t - floor(t) is in [0, 1] — it ROUNDS to 1 for a tiny negative t (-1e-22 - (-1)) — so the product is clamped to the last unit
  const int qx = min((int)((tx - fx) * (double)QONE), (int)QMASK), qy = min((int)((ty - fy) * (double)QONE), (int)QMASK),
            qz = min((int)((tz - fz) * (double)QONE), (int)QMASK);
  q = c.ok ? ((uint32_t)qx | ((uint32_t)qy << QBITS) | ((uint32_t)qz << (2 * QBITS))) : 0u;
  return c;
}

// Can a member with in-cell coordinates qj, in the cell (ox, oy, oz) cells away, be within the list radius of a UAV with in-cell
// coordinates qi?  Every coordinate difference in units is within (-1, 1) of 1024 x the true difference in cells (two floors), the
// vector of the three errors is shorter than sqrt(3): the squared integer distance is compared with (radius in units + 1.75)^2
// (q_near below; sqrt(LIST_R2) < sqrt(3) + SKIN + 1e-5).

struct __attribute__((aligned(8))) HeadPair {  // two consecutive table entries in one 16-byte request (the table is 8-byte aligned)
  uint32_t x, y, z, w;
  __device__ operator uint4() const { return make_uint4(x, y, z, w); }
};
// hashes of the list-building format, split into a column part (cx, cy) and a cheap per-cell part: a prober computes the first once
// per column of three cells.
__device__ __forceinline__ uint32_t bucket_col(int cx, int cy) {  // bucket_of(cx, cy, cz, mask) == (bucket_col(cx, cy) + cz) & mask
  uint32_t h = ((uint32_t)cx * 73856093u) ^ ((uint32_t)cy * 19349663u);
  h ^= h >> 15;
  h *= 0x2c1b3c6du;
  h ^= h >> 12;
  return h;
}
__device__ __forceinline__ uint32_t tag_col(int cx, int cy) { return (uint32_t)cx * 0x9E3779B1u + (uint32_t)cy * 0x85EBCA77u; }
__device__ __forceinline__ uint32_t tag_fin(uint32_t tcol, int cz, int ib) {  // the top 32 - ib bits of a product: every lower bit of the sum reaches them
  return ((tcol + (uint32_t)cz * 0xC2B2AE3Du) * 0x27D4EB2Fu) >> ib;  // (one quarter-rate multiply costs less issue time than a shift-and-add mix of equal reach)
}
__device__ __forceinline__ uint32_t tag_bits(int cx, int cy, int cz, int ib) { return tag_fin(tag_col(cx, cy), cz, ib); }
// the test described above, with the prober's part of the three differences precomputed (offset x 1024 - own coordinate); 24-bit multiplies
template <int WIDE>
__device__ __forceinline__ bool q_near(int dx, int dy, int dz, uint32_t qj) {
  constexpr double ic  = WIDE == 2 ? INV_CELL_WIDE2 : INV_CELL_WIDE;
  constexpr double ru  = (SQRT3_UP + (WIDE == 2 ? SKIN2 : SKIN) + 1e-5) * ic * (double)QONE + 1.75;
  constexpr int    thr2 = (int)(ru * ru) + 1;
  dx += (int)(qj & QMASK);
  dy += (int)((qj >> QBITS) & QMASK);
  dz += (int)((qj >> (2 * QBITS)) & QMASK);
  return __mul24(dx, dx) + __mul24(dy, dy) + __mul24(dz, dz) <= thr2;
}

__device__ __forceinline__ void insert_uav2(long long j, const Cell& c, uint32_t q, uint32_t mask, int ib, uint2* head, uint2* next) {
  if (!c.ok) {  // never inserted
    next[j] = make_uint2(0u, 0u);
    return;
  }
  const uint32_t b = bucket_of(c.x, c.y, c.z, mask);
  const unsigned long long me = (unsigned long long)(((uint32_t)j + 1u) | (tag_bits(c.x, c.y, c.z, ib) << ib)) | ((unsigned long long)(q & ~CHAIN_BIT) << 32);
  unsigned long long* slot = reinterpret_cast<unsigned long long*>(head + b);
  const unsigned long long old = atomicExch(slot, me);
  next[j] = make_uint2((uint32_t)old, (uint32_t)(old >> 32) & ~CHAIN_BIT);
  if ((uint32_t)old != 0u) atomicOr(slot, (unsigned long long)CHAIN_BIT << 32);  // (see insert_uav)
}

// ctl[0], ctl[1]: "some UAV has left its skin" flags of alternating ticks (written by the step kernel), see the header
template <int LISTS>
__global__ void k_insert(const PosRecord* rec, long long n_total, uint32_t mask, uint2* head, uint2* next) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_total) return;
  insert_uav(j, cell_of<LISTS>(rec[j].x, rec[j].y, rec[j].z), mask, head, next);
}

// ---- query ----
// predicate and force expression: collide_device.inc (shared with the fused evaluation inside the step kernels)
#define apply_partner mrs_apply_partner
#define qualifies mrs_qualifies

// reference path for one lane: repeated sweeps over the 27 bucket chains, each returning the smallest qualifying partner
// index above the previous one (ascending-index accumulation without per-lane arrays).  Correct for any bucket
// occupancy; used when the wave-cooperative path below overflows its LDS lists.
template <int WIDE>
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
        const Cell      oc = cell_of<WIDE>(o.x, o.y, o.z);
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
//   B  the (owner lane, probed cell, candidate) triples of the whole wave are compacted into an LDS list (wave prefix sum),
//      PAIR_CAP at a time
//   C  the list is swept 64 x U entries at a time with uniform control flow — this is what removes the 27-way divergent
//      walk in which some lane always had a non-empty bucket and every iteration paid a full memory latency.  An entry
//      of a chained bucket also fetches its `next` link, which takes over the entry's slot for the next sweep; qualifying
//      partners go to a small per-owner hit list (LDS atomics)
//   D  owners order their (rare) hits by index and accumulate; a UAV with more hits than its list holds falls back to
//      query_lane_sweeps
#ifndef MRS_TABLE_FACTOR
#define MRS_TABLE_FACTOR 4
#endif
// bucket heads taken into the LDS list per pass.  Sized so that the kernel keeps its LDS under 20 KB: eight one-wave blocks per CU,
// 2048 on the chip — the 1563 blocks of a 100 k swarm then run in ONE round (with 1024 entries, 24.6 KB, six blocks per CU = 1536,
// the last 27 blocks waited for a second round: rounds 2-3, when this kernel also built the neighbour lists — k_query2 does now).
#ifndef MRS_PAIR_CAP
#define MRS_PAIR_CAP 640
#endif
constexpr int      PAIR_CAP  = MRS_PAIR_CAP;
constexpr int      HIT_CAP   = 6;
constexpr uint32_t META_WALK = 0x10000u;  // pair meta: owner lane | probed cell q << 8 | WALK (follow the `next` link)

#ifdef MRS_QUERY_CLOCK
// constant-clock timestamp that cannot be scheduled before `dep` is available, nor across memory operations
__device__ __forceinline__ unsigned long long clock_fence(uint32_t dep) {
  unsigned long long t;
  asm volatile("v_readfirstlane_b32 s4, %1\n s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory", "s4");
  return t;
}
#endif

// One tick from the neighbour lists: current positions of the listed UAVs, literal predicate, ascending partner index.
// Most UAVs have nobody listed: they only read their count and write the zero force (applyForce runs for every UAV, :356-358).
__device__ __forceinline__ PosRecord current_record(const SwarmDev& sw, const PosRecord* rec, uint32_t j) {
  const size_t np = (size_t)sw.npad;
  PosRecord    o  = rec[j];  // airframe constants; the position is the CURRENT one
  o.x = sw.S[(size_t)(F_X + 0) * np + j];
  o.y = sw.S[(size_t)(F_X + 1) * np + j];
  o.z = sw.S[(size_t)(F_X + 2) * np + j];
  return o;
}

__device__ __forceinline__ void list_tick(const SwarmDev& sw, const PosRecord* rec, const uint32_t* nbr, uint32_t cnt, uint32_t j0, int i, int crash,
                                          double rebounce, Pos4* pos_now) {
  const size_t np = (size_t)sw.npad;  // cnt / j0: the UAV's list length and first row (always a valid index, stale beyond cnt)
  double fx = 0.0, fy = 0.0, fz = 0.0;
  bool   crashed = false;
  {  // the position the partners of this UAV read in a following fused step + collision launch (step_device.inc *_coll)
    const Pos4 pp = {sw.S[(size_t)(F_X + 0) * np + i], sw.S[(size_t)(F_X + 1) * np + i], sw.S[(size_t)(F_X + 2) * np + i],
                     (double)(sw.F[i] >> FLAG_TYPE_SHIFT)};  // .w: airframe type (collide_device.inc)
    pos_now[i]    = pp;
  }
  if (cnt) {
    const PosRecord me = current_record(sw, rec, (uint32_t)i);
    PosRecord       o  = current_record(sw, rec, j0);  // independent of `me`: one memory round trip for both
    if (cell_of<true>(me.x, me.y, me.z).ok) {
      for (uint32_t k = 0;;) {
        if (cell_of<true>(o.x, o.y, o.z).ok && qualifies(me, o, crash)) apply_partner(me, o, crash, rebounce, fx, fy, fz, crashed);
        if (++k >= cnt) break;
        o = current_record(sw, rec, nbr[(size_t)k * sw.n + i]);
      }
    }
  }
  sw.S[(size_t)(F_FEXT + 0) * np + i] = fx;
  sw.S[(size_t)(F_FEXT + 1) * np + i] = fy;
  sw.S[(size_t)(F_FEXT + 2) * np + i] = fz;
  if (crashed) sw.F[i] |= FLAG_CRASHED;
}

// single-GPU tick: pack and insert in one pass over the state (the records are still written: the query reads them).
// With LISTS the pass only happens on a rebuild tick (the records then double as the reference positions of the skin test);
// on every other tick this light kernel evaluates the neighbour lists and the query kernel that follows returns at once.
template <int LISTS>
__global__ void k_pack_insert(SwarmDev sw, PosRecord* rec, uint32_t mask, uint2* head, uint2* next, uint32_t* ctl, int cur, int force,
                              int table_id, uint2* head_to_clear, uint32_t table_size, const uint32_t* nbr, const uint32_t* nbr_cnt, int crash,
                              double rebounce, Pos4* pos_now, const uint32_t* stall_word, int ib) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  (void)stall_word;  // (the pass always runs in full: the two head tables are wiped alternately, a skipped pass would leave stale chains)
  if (LISTS) {
    // issued before the control words are looked at: on a list tick (the common case) these are the first links of the
    // dependent load chain, and the addresses are valid on a rebuild tick too
    const int      ic  = i < sw.n ? i : sw.n - 1;
    const uint32_t cnt = nbr_cnt[ic], j0 = nbr[ic];
    asm volatile("" ::: "memory");
    const bool rebuild = force || ctl[cur] != 0u;
    if (i == 0) {
      ctl[cur ^ 1]     = 0u;  // the next tick's flag: the step kernel (or this tick's query, on list overflow) raises it
      ctl[4 + table_id] = rebuild ? 1u : 0u;  // "this head table holds entries": the next tick wipes it if so
      if (rebuild) ctl[2] += 1u;              // statistics: number of rebuild ticks
    }
    // the head table of the NEXT rebuild must be empty: wipe it if the previous tick filled it (grid-strided, coalesced)
    if (ctl[4 + (table_id ^ 1)]) {
      const uint32_t stride = gridDim.x * blockDim.x;
      for (uint32_t t = (uint32_t)i; t < table_size; t += stride) head_to_clear[t] = make_uint2(0u, 0u);
    }
    if (!rebuild) {  // wave-uniform: nobody has left its skin since the lists were built — this kernel IS the collision tick
      if (i < sw.n) list_tick(sw, rec, nbr, cnt, j0, i, crash, rebounce, pos_now);
      return;
    }
  }
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
  if (LISTS) {
    const Pos4 pp = {r.x, r.y, r.z, (double)(sw.F[i] >> FLAG_TYPE_SHIFT)};
    pos_now[i]    = pp;
    uint32_t   q;
    const Cell c = cell_q<LISTS ? LISTS : 1>(r.x, r.y, r.z, q);
    insert_uav2(i, c, q, mask, ib, head, next);
  } else {
    insert_uav(i, cell_of<LISTS>(r.x, r.y, r.z), mask, head, next);
  }
}

// ---- neighbour lists over GATHERED records (multi-GPU ticks): every rank holds the current records of all UAVs after the
// all-gather, so the skin test, the rebuild decision and the list tick are local to the rank — ranks may rebuild on different ticks.
__device__ __forceinline__ bool record_usable(const PosRecord& r) { return cell_of<true>(r.x, r.y, r.z).ok; }

// flag (ctl[cur]) raised when any record has left its skin since this rank's last rebuild, appeared / disappeared (NaN padding
// and non-finite positions are "absent"), or changed its airframe constants
__global__ void k_skin_gathered(const PosRecord* rec, const PosRecord* rec_build, long long n_total, uint32_t* ctl, int cur, double lim2) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j == 0) ctl[cur ^ 1] = 0u;  // the next tick's flag (this tick's query raises it on list overflow)
  if (j >= n_total) return;
  const PosRecord c = rec[j], b = rec_build[j];
  const bool      uc = record_usable(c), ub = record_usable(b);
  bool            moved = uc != ub;
  if (uc && ub) {
    const double d0 = c.x - b.x, d1 = c.y - b.y, d2 = c.z - b.z;
    moved = !((d0 * d0 + d1 * d1) + d2 * d2 <= lim2) || c.mass != b.mass || c.arm_length != b.arm_length || c.prop_radius != b.prop_radius;
  }
  if (moved) ctl[cur] = 1u;
}

__device__ __forceinline__ void list_tick_gathered(const SwarmDev& sw, const PosRecord* rec, long long my_offset, const uint32_t* nbr, uint32_t cnt,
                                                   uint32_t j0, int i, int crash, double rebounce) {
  const size_t np = (size_t)sw.npad;
  double fx = 0.0, fy = 0.0, fz = 0.0;
  bool   crashed = false;
  if (cnt) {
    const PosRecord me = rec[my_offset + i];
    PosRecord       o  = rec[j0];
    if (record_usable(me)) {
      for (uint32_t k = 0;;) {
        if (record_usable(o) && qualifies(me, o, crash)) apply_partner(me, o, crash, rebounce, fx, fy, fz, crashed);
        if (++k >= cnt) break;
        o = rec[nbr[(size_t)k * sw.n + i]];
      }
    }
  }
  sw.S[(size_t)(F_FEXT + 0) * np + i] = fx;
  sw.S[(size_t)(F_FEXT + 1) * np + i] = fy;
  sw.S[(size_t)(F_FEXT + 2) * np + i] = fz;
  if (crashed) sw.F[i] |= FLAG_CRASHED;
}

// gathered records, lists on: list tick for the rank's own UAVs, or (rebuild) insert of ALL records + a copy of them as the
// reference of the next skin tests
// Bounding box of this rank's usable records, widened by the list radius: bb[0..2] lower, bb[3..5] upper corner.  Records of other
// ranks outside it cannot be on any list of this rank, so a search inserts the own shard plus its halo instead of the whole swarm
// (1/8 of the inserts and of the table's occupancy at eight slab shards).  Two small launches; they return at once on list ticks.
constexpr int BBOX_BLOCKS = 128;
__device__ __forceinline__ void bbox_reduce_block(double (&l)[3], double (&h)[3], double (*lo)[256], double (*hi)[256]) {
  const int t = threadIdx.x;
  for (int c = 0; c < 3; c++) {
    lo[c][t] = l[c];
    hi[c][t] = h[c];
  }
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s)
      for (int c = 0; c < 3; c++) {
        lo[c][t] = fmin(lo[c][t], lo[c][t + s]);
        hi[c][t] = fmax(hi[c][t], hi[c][t + s]);
      }
    __syncthreads();
  }
}
// stage 1: BBOX_BLOCKS partial boxes over the rank's own records
__global__ void __launch_bounds__(256) k_own_bbox_part(const PosRecord* rec, long long my_offset, int n_own, double* part, const uint32_t* ctl, int cur,
                                                       int force) {
  __shared__ double lo[3][256], hi[3][256];
  if (!force && ctl[cur] == 0u) return;  // a list tick: nobody searches
  double l[3] = {1e300, 1e300, 1e300}, h[3] = {-1e300, -1e300, -1e300};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_own; i += BBOX_BLOCKS * 256) {
    const PosRecord r = rec[my_offset + i];
    if (!record_usable(r)) continue;
    l[0] = fmin(l[0], r.x); l[1] = fmin(l[1], r.y); l[2] = fmin(l[2], r.z);
    h[0] = fmax(h[0], r.x); h[1] = fmax(h[1], r.y); h[2] = fmax(h[2], r.z);
  }
  bbox_reduce_block(l, h, lo, hi);
  if (threadIdx.x < 3) {
    part[blockIdx.x * 6 + threadIdx.x]     = lo[threadIdx.x][0];
    part[blockIdx.x * 6 + 3 + threadIdx.x] = hi[threadIdx.x][0];
  }
}
// stage 2: the box, widened by `margin`; box_out (sharded swarms): the tail of this rank's slot-map block, where the box travels to
// the other ranks — what they choose the halo of the NEXT search by (mrs_collide_halo_*)
__global__ void __launch_bounds__(256) k_own_bbox_final(const double* part, int nparts, double margin, double* bb, uint32_t* ctl, int cur, int force,
                                                        double* box_out) {
  __shared__ double lo[3][256], hi[3][256];
  if (force && threadIdx.x == 0) ctl[cur ^ 1] = 0u;  // (a decided search: the flag word this tick's query may raise starts from 0 — k_skin_gathered's other job)
  if (!force && ctl[cur] == 0u) return;
  double l[3] = {1e300, 1e300, 1e300}, h[3] = {-1e300, -1e300, -1e300};
  for (int p = threadIdx.x; p < nparts; p += 256)
    for (int c = 0; c < 3; c++) {
      l[c] = fmin(l[c], part[p * 6 + c]);
      h[c] = fmax(h[c], part[p * 6 + 3 + c]);
    }
  bbox_reduce_block(l, h, lo, hi);
  if (threadIdx.x < 3) {
    const double a = lo[threadIdx.x][0] - margin, b = hi[threadIdx.x][0] + margin;  // (no usable own record: -+1e300 +- the widening — nothing is inside)
    bb[threadIdx.x]     = a;
    bb[3 + threadIdx.x] = b;
    if (box_out) {
      box_out[threadIdx.x]     = a;
      box_out[3 + threadIdx.x] = b;
    }
  }
}

__global__ void k_insert_gathered_lists(SwarmDev sw, const PosRecord* rec, PosRecord* rec_build, long long n_total, long long my_offset, uint32_t mask,
                                        uint2* head, uint2* next, uint32_t* ctl, int cur, int force, int table_id, uint2* head_to_clear,
                                        uint32_t table_size, const uint32_t* nbr, const uint32_t* nbr_cnt, int crash, double rebounce, const double* bb,
                                        int own_copy_only, int ib) {
  const long long j  = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int       ic = j < sw.n ? (int)j : sw.n - 1;
  const uint32_t  cnt = nbr_cnt[ic], j0 = nbr[ic];
  asm volatile("" ::: "memory");
  const bool rebuild = force || ctl[cur] != 0u;
  if (j == 0) {
    ctl[4 + table_id] = rebuild ? 1u : 0u;
    if (rebuild) ctl[2] += 1u;
  }
  if (ctl[4 + (table_id ^ 1)]) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t t = (uint32_t)j; t < table_size; t += stride) head_to_clear[t] = make_uint2(0u, 0u);
  }
  if (!rebuild) {
    if (j < sw.n) list_tick_gathered(sw, rec, my_offset, nbr, cnt, j0, (int)j, crash, rebounce);
    return;
  }
  if (j >= n_total) return;
  const PosRecord r = rec[j];
  if (!own_copy_only || (j >= my_offset && j < my_offset + sw.n)) rec_build[j] = r;  // (42 MB less to write per search at 8 x 125 k)
  uint32_t        q;
  Cell            c = cell_q<2>(r.x, r.y, r.z, q);
  // outside this rank's widened bounding box: nobody here can list it (a comparison with NaN bounds — no usable own record — is false)
  c.ok = c.ok && r.x >= bb[0] && r.y >= bb[1] && r.z >= bb[2] && r.x <= bb[3] && r.y <= bb[4] && r.z <= bb[5];
  insert_uav2(j, c, q, mask, ib, head, next);
}

// The 27 bucket heads of a UAV's neighbourhood, filtered: .x != 0 marks an entry for the work list (a head of the probed cell,
// or any head of a chained bucket), .y = its tag with CHAIN_BIT kept for chained buckets.  Returns the number of entries.
// Unconditional loads from always-valid addresses: a load under a divergent branch is waited for at the join, which would
// serialise 27 memory round trips.
__device__ __forceinline__ uint32_t load_heads(const Cell& c, uint32_t mask, const uint2* head, uint2 (&info)[27]) {
  uint32_t tc = 0;
#pragma unroll
  for (int q = 0; q < 27; q++) info[q] = head[bucket_of(c.x + q / 9 - 1, c.y + (q / 3) % 3 - 1, c.z + q % 3 - 1, mask)];
#pragma unroll
  for (int q = 0; q < 27; q++) {
    const uint32_t tg    = cell_tag(c.x + q / 9 - 1, c.y + (q / 3) % 3 - 1, c.z + q % 3 - 1) & ~CHAIN_BIT;
    const bool     chain = (info[q].y & CHAIN_BIT) != 0u;
    const bool     take  = c.ok && info[q].x != 0u && (chain || info[q].y == tg);
    info[q].y = (info[q].y & ~CHAIN_BIT);
    if (!take) info[q].x = 0u;
    if (take && chain) info[q].y |= CHAIN_BIT;
    tc += take ? 1u : 0u;
  }
  return tc;
}

// entries [wbase, wbase + PAIR_CAP) of the wave's head sequence -> LDS work list (first_slot: this lane's position in the sequence)
__device__ __forceinline__ void fill_window(const uint2 (&info)[27], uint32_t first_slot, uint32_t wbase, int lane, uint2* pair_e, uint32_t* pair_m) {
  uint32_t slot = first_slot;
#pragma unroll
  for (int q = 0; q < 27; q++) {
    if (info[q].x != 0u) {
      if (slot >= wbase && slot - wbase < (uint32_t)PAIR_CAP) {
        pair_e[slot - wbase] = make_uint2(info[q].x, info[q].y & ~CHAIN_BIT);
        pair_m[slot - wbase] = (uint32_t)lane | ((uint32_t)q << 8) | ((info[q].y & CHAIN_BIT) ? META_WALK : 0u);
      }
      slot++;
    }
  }
}

struct QueryLds {  // the LDS arrays of k_query, handed to the helper below
  uint2*      pair_e;
  uint32_t*   pair_m;
  PosRecord*  me_s;
  int4*       me_cell;
  uint32_t*   hit_j;   // [64][HIT_CAP]
  uint32_t*   hit_n;
  uint32_t*   next_n;
  uint32_t*   hit_overflow;
};

// C: uniform sweeps over one window of the work list, U entries per lane and iteration (see below for U).
// A chained bucket is walked inside the list: after a member has been evaluated, the next member of its chain is written back —
// COMPACTED to the front (slot k < the number of entries read so far, so no unread entry is overwritten) — and the shrunken
// list is swept again until every chain has ended.  Never more entries than the sweep started with: nothing can overflow, and
// the later levels of the walk (a few percent of the heads) cost one short pass each instead of a pass over every head.
__device__ __forceinline__ void sweep_window(const QueryLds& L, uint32_t wn, int lane, const PosRecord* rec, const uint2* next, long long wave_first,
                                             int crash) {
  // entries per lane and iteration.  Four looked right while the kernel waited on single round trips; since the record loads travel
  // with the chain links and the later levels are compacted, ONE is fastest (100 k UAVs: 26.9 us, U = 2: 27.8, 4: 29.0, 6: 31.1; at
  // 16 m^3 per UAV 63.9 vs 68.0): the work list of a wave is 4-5 entries per lane, and what limits it now is the rate of scattered accesses.
#ifndef MRS_SWEEP_U
#define MRS_SWEEP_U 1
#endif
  constexpr int U = MRS_SWEEP_U;
  while (wn != 0u) {
    if (lane == 0) *L.next_n = 0;  // entries of the next sweep
    __syncthreads();
    for (uint32_t base = 0; base < wn; base += 64 * U) {
      uint2    pe[U], nx[U];
      uint32_t pm[U];
      bool     live[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t p = base + u * 64 + lane;
        pe[u] = p < wn ? L.pair_e[p] : make_uint2(0u, 0u);  // .x == 0: the slot's chain has ended
        pm[u] = p < wn ? L.pair_m[p] : 0u;
      }
#pragma unroll
      for (int u = 0; u < U; u++) nx[u] = (pe[u].x != 0u && (pm[u] & META_WALK)) ? next[pe[u].x - 1u] : make_uint2(0u, 0u);
      // tag filter: only members of exactly the probed cell survive (false positives of the 31-bit tag are caught by the exact
      // cell comparison below)
#pragma unroll
      for (int u = 0; u < U; u++) {
        live[u] = false;
        if (pe[u].x != 0u) {
          const int  ow = (int)(pm[u] & 0xFFu), q = (int)((pm[u] >> 8) & 0xFFu);
          const int4 mc = L.me_cell[ow];
          live[u] = pe[u].y == (cell_tag(mc.x + q / 9 - 1, mc.y + (q / 3) % 3 - 1, mc.z + q % 3 - 1) & ~CHAIN_BIT) &&
                    (long long)pe[u].x - 1 != wave_first + ow;  // idx == i, src/multirotor_simulator.cpp:335
        }
      }
      // the candidates' records, requested together with the chain links above (one memory round trip per sweep, not two);
      // unconditional loads from always-valid addresses, see load_heads
      PosRecord ob[U];
#pragma unroll
      for (int u = 0; u < U; u++) ob[u] = rec[live[u] ? (long long)pe[u].x - 1 : wave_first];
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (nx[u].x != 0u) {  // the chain goes on: its next member joins the next sweep
          const uint32_t k = atomicAdd(L.next_n, 1u);
          L.pair_e[k] = make_uint2(nx[u].x, nx[u].y & ~CHAIN_BIT);
          L.pair_m[k] = pm[u];
        }
        if (!live[u]) continue;
        const int       ow = (int)(pm[u] & 0xFFu), q = (int)((pm[u] >> 8) & 0xFFu);
        const int4      mc = L.me_cell[ow];
        const PosRecord o  = ob[u];
        const Cell      oc = cell_of<0>(o.x, o.y, o.z);
        if (oc.x != mc.x + q / 9 - 1 || oc.y != mc.y + (q / 3) % 3 - 1 || oc.z != mc.z + q % 3 - 1) continue;  // tag collision
        const PosRecord m = L.me_s[ow];
        if (!qualifies(m, o, crash)) continue;
        const uint32_t k = atomicAdd(&L.hit_n[ow], 1u);
        if (k < HIT_CAP)
          L.hit_j[ow * HIT_CAP + k] = pe[u].x - 1u;
        else
          *L.hit_overflow = 1;
      }
    }
    __syncthreads();
    wn = *L.next_n;
    __syncthreads();  // nobody resets the counter before everybody has read it
  }
}

// The search-every-tick query (neighbour lists switched off, or the caller-driven gathered form): one lane per UAV, see above.
__global__ void __launch_bounds__(64) k_query(SwarmDev sw, const PosRecord* rec, long long n_total, long long my_offset, uint32_t mask,
                                              const uint2* head, const uint2* next, uint2* head_to_clear, uint32_t table_size, int crash,
                                              double rebounce) {
  __shared__ PosRecord me_s[64];
  __shared__ int4      me_cell[64];
  __shared__ uint2     pair_e[PAIR_CAP];   // x: candidate index + 1, y: its tag
  __shared__ uint32_t  pair_m[PAIR_CAP];   // meta
  __shared__ uint32_t  hit_j[64][HIT_CAP];
  __shared__ uint32_t  hit_n[64];
  __shared__ uint32_t  list_total, hit_overflow;

  const int       lane   = threadIdx.x;
  const int       i      = blockIdx.x * 64 + lane;
  const bool      active = i < sw.n;
  const long long gi     = my_offset + i;
  PosRecord       me;
  me.x = me.y = me.z = __longlong_as_double(0x7ff8000000000000ll);
  me.mass = me.arm_length = me.prop_radius = 0.0;
  if (active) me = rec[gi];
  const Cell c = cell_of<0>(me.x, me.y, me.z);
  me_s[lane]   = me;
  me_cell[lane] = make_int4(c.x, c.y, c.z, 0);
  hit_n[lane]  = 0;
  if (lane == 0) hit_overflow = 0;

  __shared__ uint32_t heads_total;
#ifdef MRS_QUERY_CLOCK
  const unsigned long long t0 = clock_fence(0u);
  unsigned long long       tA = 0, tB = 0;
#endif
  // A: bucket heads
  uint32_t first_slot, n_heads;
  {
    uint2          info[27];
    const uint32_t tc = load_heads(c, mask, head, info);
    // this tick's table has been read by this wave's probes: wipe the other one for the next tick (grid-strided, coalesced)
    {
      const uint32_t stride = gridDim.x * 64u;
      for (uint32_t t = blockIdx.x * 64u + lane; t < table_size; t += stride) head_to_clear[t] = make_uint2(0u, 0u);
    }
#if defined(MRS_QUERY_STOP) && MRS_QUERY_STOP == 1
    if (tc == 0xFFFFFFFFu) sw.F[i] = tc;
    return;
#endif
#ifdef MRS_QUERY_CLOCK
    tA = clock_fence(tc);
#endif
    // B: wave prefix sum -> slots in the work list
    uint32_t inc = tc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(inc, off, 64);
      if (lane >= off) inc += o;
    }
    if (lane == 63) heads_total = inc;
    first_slot = inc - tc;
    __syncthreads();
    n_heads = heads_total;
    if (n_heads <= (uint32_t)PAIR_CAP) fill_window(info, first_slot, 0u, lane, pair_e, pair_m);  // the usual case: one window, `info` dies here
  }
#ifdef MRS_QUERY_CLOCK
  tB = clock_fence(n_heads);
#endif
  const QueryLds lds{pair_e, pair_m, me_s, me_cell, &hit_j[0][0], hit_n, &list_total, &hit_overflow};
  const long long wave_first = my_offset + (long long)blockIdx.x * 64;
  if (n_heads <= (uint32_t)PAIR_CAP) {
    __syncthreads();
    sweep_window(lds, n_heads, lane, rec, next, wave_first, crash);
  } else {
    // dense neighbourhood: more heads than the list holds — windows of PAIR_CAP, the heads are probed again per window
    // (cheaper than keeping 27 head words per lane alive across the sweeps of the usual case)
    for (uint32_t wbase = 0; wbase < n_heads; wbase += PAIR_CAP) {
      uint2 info[27];
      load_heads(c, mask, head, info);
      fill_window(info, first_slot, wbase, lane, pair_e, pair_m);
      __syncthreads();
      sweep_window(lds, n_heads - wbase < (uint32_t)PAIR_CAP ? n_heads - wbase : (uint32_t)PAIR_CAP, lane, rec, next, wave_first, crash);
    }
  }
  __syncthreads();
#if defined(MRS_QUERY_STOP) && MRS_QUERY_STOP == 3
  if (hit_n[lane] == 0xFFFFFFFFu) sw.F[i] = 1;
  return;
#endif
#ifdef MRS_QUERY_CLOCK
  const unsigned long long tC = clock_fence(hit_n[lane]);
#endif
  // D: owners accumulate their hits in ascending partner index
  double fx = 0.0, fy = 0.0, fz = 0.0;
  bool   crashed = false;
  if (active && c.ok) {
    if (hit_overflow && hit_n[lane] > HIT_CAP) {  // more qualifying partners than the hit list holds: the reference path
      query_lane_sweeps<0>(me, c, gi, rec, n_total, mask, head, next, crash, rebounce, fx, fy, fz, crashed);
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
#ifdef MRS_QUERY_CLOCK  // timing build (tools/query_phases.py): the force columns carry phase durations in 10-ns ticks
#if MRS_QUERY_CLOCK == 2
  const unsigned long long tD = clock_fence(0u);
  fx = (double)(tB - t0); fy = (double)(tC - tB); fz = (double)(tD - tC);
#else
  fx = (double)(tA - t0); fy = (double)(tB - tA); fz = (double)(tC - tB);
#endif
#endif
  if (active) {
    sw.S[(size_t)(F_FEXT + 0) * sw.npad + i] = fx;
    sw.S[(size_t)(F_FEXT + 1) * sw.npad + i] = fy;
    sw.S[(size_t)(F_FEXT + 2) * sw.npad + i] = fz;
    if (crashed) sw.F[i] |= FLAG_CRASHED;
  }
}

// ---- the list-building query (LISTS = 1, 2) ----
// LPU lanes per UAV share its nine probe columns, a single-wave block serves 64 / LPU UAVs.  What bounds the kernel is the number of
// instructions its waves issue (rocprofv3 counters, MEASUREMENTS 6.6: a SIMD spent 17 of the first version's 22 us issuing), so the
// code below is written for few of them: hashes split into a column part and a cheap per-cell part, 24-bit multiplies, one LDS
// atomic per lane, and as many UAVs per wave as still leaves every SIMD a few waves to overlap round trips with.
//   1  every lane fetches the bucket heads of its columns (one round trip; three consecutive buckets = two requests) and decides from
//      the entries alone (tag, in-cell coordinates) which members can be within the list radius; those, and the heads of chained
//      buckets, become ITEMS of the wave's LDS work list
//   2  the work list is swept with uniform control flow, one item per lane: the member's position (exact cell, exact squared distance)
//      and its chain link, requested together; a link that goes on is written back — compacted to the front, see sweep_window — and
//      the shrunken list is swept again until every chain has ended.  The literal collision predicate is evaluated on the (rare) members
//      closer than sqrt(3)
//   3  the lanes of a UAV order its list by index and write it out; its first lane accumulates the hits of THIS tick in ascending
//      partner index, exactly as k_query does
constexpr uint32_t IT_VERIFY = 1u << 14, IT_WALK = 1u << 15;  // item meta (16 bits): owner | probe << 6 | flags

// reference path for one UAV: see query_lane_sweeps (the same sweeps over entries of the list-building format)
template <int WIDE>
__device__ void query_lane_sweeps2(const PosRecord& me, const Cell& c, long long gi, const PosRecord* rec, long long n_total, uint32_t mask, int ib,
                                   const uint2* head, const uint2* next, int crash, double rebounce, double& fx, double& fy, double& fz,
                                   bool& crashed) {
  const uint32_t ibm = (1u << ib) - 1u;
  long long      prev = -1;
  for (;;) {
    long long best = n_total;
    for (int q = 0; q < 27; q++) {
      const int      cx = c.x + q / 9 - 1, cy = c.y + (q / 3) % 3 - 1, cz = c.z + q % 3 - 1;
      const uint32_t tg = tag_bits(cx, cy, cz, ib);
      for (uint2 e = head[bucket_of(cx, cy, cz, mask)]; e.x != 0u; e = next[(e.x & ibm) - 1u]) {
        const long long j = (long long)(e.x & ibm) - 1;
        if ((e.x >> ib) != tg || j <= prev || j >= best || j == gi) continue;
        const PosRecord o  = rec[j];
        const Cell      oc = cell_of<WIDE>(o.x, o.y, o.z);
        if (oc.x != cx || oc.y != cy || oc.z != cz) continue;  // another cell sharing the bucket (and the tag)
        if (qualifies(me, o, crash)) best = j;
      }
    }
    if (best >= n_total) break;
    apply_partner(me, rec[best], crash, rebounce, fx, fy, fz, crashed);
    prev = best;
  }
}

#ifdef MRS_Q2_CLOCK
// 100-MHz timestamp that cannot be scheduled before `dep` is available, nor across memory operations
__device__ __forceinline__ unsigned long long clock_dep(uint32_t dep) {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n v_readfirstlane_b32 s4, %1\n s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(dep) : "memory", "s4");
  return t;
}
#endif
template <int LISTS, int LPU>
__global__ void __launch_bounds__(64) k_query2(SwarmDev sw, const PosRecord* rec, long long n_total, long long my_offset, uint32_t mask, int ib,
                                               const uint2* head, const uint2* next, int crash, double rebounce, uint32_t* ctl, int cur, int force,
                                               uint32_t* nbr, uint32_t* nbr_cnt, uint32_t* stall_word, volatile uint32_t* hostw, uint32_t stall_tau) {
  constexpr int UPW    = 64 / LPU;              // UAVs per wave
  constexpr int CPL    = (9 + LPU - 1) / LPU;   // columns (cx, cy) of three probes per lane
  constexpr int WL_CAP = UPW * 27;              // every probe of the wave an item: the list cannot overflow
  if (!force && ctl[cur] == 0u) return;  // list tick: the insert kernel has done the work
#ifdef MRS_Q2_CLOCK  // timing build (tools/search_phases.py): 100-MHz stamps of the wave's phases instead of the first UAV's list
  uint32_t ts[6];
#define MRS_Q2_STAMP(k, dep) ts[k] = (uint32_t)clock_dep(dep)
  MRS_Q2_STAMP(0, 0u);
#else
#define MRS_Q2_STAMP(k, dep)
#endif
  // (a search queued ahead of time behind launches that have stalled: lists as usual, no forces or crash flags — see k_query)
  const bool muted = stall_word && *stall_word != 0u;
  __shared__ int4      me_cell[UPW];  // cell, .w = in-cell coordinates
  __shared__ uint32_t  wl_j[WL_CAP + 1];  // work list: member (record index),   (+1: where the items nobody wants are written)
  __shared__ uint16_t  wl_m[WL_CAP + 1];  //            meta
  __shared__ uint32_t  wl_n, next_n, hit_overflow;
  __shared__ uint32_t  nl_j[UPW][LIST_CAP], nl_n[UPW];
  constexpr int HITS = 4;  // hits kept per UAV before the reference path takes over (LDS is allocated in 1280-byte steps: with three lanes
                           // per UAV these 6.3 KB are 6 400, 24 blocks per CU — one round for a 125 k-UAV shard; six hits made it 21)
  __shared__ uint32_t  hit_j[UPW][HITS], hit_n[UPW];

  const int       lane = threadIdx.x, r = lane % LPU;
  const bool      seated = lane / LPU < UPW;  // (three lanes per UAV leave the 64th lane without one)
  const int       u      = seated ? lane / LPU : 0;
  const int       i      = blockIdx.x * UPW + u;
  const bool      active = seated && i < sw.n;
  const long long gi     = my_offset + i;
  const long long wave_first = my_offset + (long long)blockIdx.x * UPW;
  const uint32_t  ibm    = (1u << ib) - 1u;
  double          mx, my, mz;
  mx = my = mz = __longlong_as_double(0x7ff8000000000000ll);
  if (active) {
    const double*  pm = reinterpret_cast<const double*>(rec + gi);
    const double2 xy = *reinterpret_cast<const double2*>(pm);
    mx = xy.x;
    my = xy.y;
    mz = pm[2];
  }
  uint32_t   qme;
  const Cell c = cell_q<LISTS>(mx, my, mz, qme);
  MRS_Q2_STAMP(1, qme);
  if (r == 0 && seated) {
    me_cell[u] = make_int4(c.x, c.y, c.z, (int)qme);
    nl_n[u]    = 0;
    hit_n[u]   = 0;
  }
  if (lane == 0) {
    wl_n         = 0;
    hit_overflow = 0;
  }
  __syncthreads();
#if defined(MRS_Q2_STOP) && MRS_Q2_STOP == 1  // (ablation builds of tools/build_variants.sh: timing only, wrong results)
  if (qme == 0xFFFFFFFFu) nbr_cnt[i] = 1;
  return;
#endif
  // 1: bucket heads, a column (cx, cy) at a time: its three cells sit in consecutive buckets (bucket_of), so two requests — 16 B and
  //    8 B — fetch them; the table carries one spare entry behind its end for the 16-B request of the last bucket, whose second half
  //    is bucket 0 (fetched on its own, once in T columns).  Unconditional loads from always-valid addresses: columns beyond the ninth
  //    repeat the lane's first.
  {
    uint2    e[CPL][3];
    uint32_t tcol[CPL];
#pragma unroll
    for (int k = 0; k < CPL; k++) {
      const int      cq = (r + k * LPU) < 9 ? (r + k * LPU) : r;
      const int      cx = c.x + cq / 3 - 1, cy = c.y + cq % 3 - 1;
      const uint32_t b0 = (bucket_col(cx, cy) + (uint32_t)(c.z - 1)) & mask;
      tcol[k]           = tag_col(cx, cy);
      const uint4    a  = *reinterpret_cast<const HeadPair*>(head + b0);
      e[k][0] = make_uint2(a.x, a.y);
      e[k][1] = make_uint2(a.z, a.w);
      e[k][2] = head[(b0 + 2u) & mask];
      if (b0 == mask) e[k][1] = head[0];
    }
    MRS_Q2_STAMP(2, e[0][0].x ^ e[CPL - 1][2].x);
    const int qix = (int)(qme & QMASK), qiy = (int)((qme >> QBITS) & QMASK), qiz = (int)((qme >> (2 * QBITS)) & QMASK);
    uint32_t  itj[CPL * 3], itm[CPL * 3], nit = 0;
#pragma unroll
    for (int k = 0; k < CPL; k++) {
      const int cq = r + k * LPU;
      const int ox = (cq < 9 ? cq : r) / 3 - 1, oy = (cq < 9 ? cq : r) % 3 - 1;
      const int dx = ox * (int)QONE - qix, dy = oy * (int)QONE - qiy;
#pragma unroll
      for (int dz = 0; dz < 3; dz++) {
        const uint2    en    = e[k][dz];
        const uint32_t j     = (en.x & ibm) - 1u;
        const bool     match = (en.x >> ib) == tag_fin(tcol[k], c.z + dz - 1, ib) && j != (uint32_t)gi &&  // idx == i, src/multirotor_simulator.cpp:335
                               q_near<LISTS>(dx, dy, (dz - 1) * (int)QONE - qiz, en.y);
        const bool     chain = (en.y & CHAIN_BIT) != 0u;
        const bool     take  = cq < 9 && c.ok && en.x != 0u && (match || chain);
        itj[k * 3 + dz] = j;
        itm[k * 3 + dz] = take ? ((uint32_t)u | ((uint32_t)(cq * 3 + dz) << 6) | (match ? IT_VERIFY : 0u) | (chain ? IT_WALK : 0u)) : 0u;
        nit += take ? 1u : 0u;
      }
    }
    uint32_t pos = nit ? atomicAdd(&wl_n, nit) : 0u;  // ONE atomic per lane; the items go to consecutive slots
#pragma unroll
    for (int t = 0; t < CPL * 3; t++) {
      const uint32_t at = itm[t] ? pos : (uint32_t)WL_CAP;  // (branch-free: an item nobody wants goes to the spare slot)
      wl_j[at] = itj[t];
      wl_m[at] = (uint16_t)itm[t];
      pos += itm[t] ? 1u : 0u;
    }
  }
  __syncthreads();
#if defined(MRS_Q2_STOP) && MRS_Q2_STOP == 2
  if (active && r == 0) nbr_cnt[i] = wl_n;
  return;
#endif
  MRS_Q2_STAMP(3, wl_n);
  // 2: sweeps.  An item asks for what it needs and nothing else — the position of a member to verify (24 B), the link of a chain to
  //    walk (8 B); the other address of each lane is one the whole wave shares
  for (uint32_t wn = wl_n; wn != 0u;) {
    if (lane == 0) next_n = 0;
    __syncthreads();
    for (uint32_t base = 0; base < wn; base += 64) {
      const uint32_t p = base + lane;
      if (p < wn) {
        const uint2   it = make_uint2(wl_j[p], (uint32_t)wl_m[p]);
        const int     ow = (int)(it.y & 0x3Fu), q = (int)((it.y >> 6) & 0x1Fu);
        const bool    ver = (it.y & IT_VERIFY) != 0u, walk = (it.y & IT_WALK) != 0u;
        const uint2   nx = next[walk ? (long long)it.x : wave_first];
        const double* po = reinterpret_cast<const double*>(rec + (ver ? (long long)it.x : wave_first));
        const double2 oxy = *reinterpret_cast<const double2*>(po);
        const double  oz_ = po[2];
        const double* pm  = reinterpret_cast<const double*>(rec + (wave_first + ow));  // (the owner's position: a cache hit — LDS is what limits the waves per CU)
        const double2 mxy = *reinterpret_cast<const double2*>(pm);
        const double  mz_ = pm[2];
        const int     cq = q / 3, ox = cq / 3 - 1, oy = cq % 3 - 1, oz = q % 3 - 1;
        const int4    mc = me_cell[ow];
        if (ver) {
          const Cell oc = cell_of<LISTS>(oxy.x, oxy.y, oz_);
          if (oc.ok && oc.x == mc.x + ox && oc.y == mc.y + oy && oc.z == mc.z + oz) {  // (else: another cell that shares bucket and tag bits)
            const double d0 = mxy.x - oxy.x, d1 = mxy.y - oxy.y, d2 = mz_ - oz_;
            const double dd = ((0.0 + d0 * d0) + d1 * d1) + d2 * d2;
            if (dd < (LISTS == 2 ? LIST_R2_2 : LIST_R2)) {
              const uint32_t k = atomicAdd(&nl_n[ow], 1u);
              if (k < (uint32_t)LIST_CAP) nl_j[ow][k] = it.x;
            }
            if (dd < 3.0) {  // close enough for the collision predicate (rare): now the airframe constants of both
              if (qualifies(rec[wave_first + ow], rec[it.x], crash)) {
                const uint32_t k = atomicAdd(&hit_n[ow], 1u);
                if (k < (uint32_t)HITS)
                  hit_j[ow][k] = it.x;
                else
                  hit_overflow = 1;
              }
            }
          }
        }
        if (walk && nx.x != 0u) {  // the chain goes on: its next member joins the next sweep
          const uint32_t j     = (nx.x & ibm) - 1u;
          const uint32_t qo    = (uint32_t)mc.w;
          const bool     match = (nx.x >> ib) == tag_fin(tag_col(mc.x + ox, mc.y + oy), mc.z + oz, ib) && (long long)j != wave_first + ow &&
                                 q_near<LISTS>(ox * (int)QONE - (int)(qo & QMASK), oy * (int)QONE - (int)((qo >> QBITS) & QMASK),
                                               oz * (int)QONE - (int)((qo >> (2 * QBITS)) & QMASK), nx.y);
          const uint32_t k = atomicAdd(&next_n, 1u);
          wl_j[k] = j;
          wl_m[k] = (uint16_t)((it.y & 0x7FFu) | IT_WALK | (match ? IT_VERIFY : 0u));
        }
      }
    }
    __syncthreads();
    wn = next_n;
    __syncthreads();  // nobody resets the counter before everybody has read it
  }
#if defined(MRS_Q2_STOP) && MRS_Q2_STOP == 3
  if (active && r == 0) nbr_cnt[i] = nl_n[u] + hit_n[u];
  return;
#endif
  MRS_Q2_STAMP(4, nl_n[u]);
  // 3: forces of this tick (first lane of every UAV), lists (all its lanes)
  double fx = 0.0, fy = 0.0, fz = 0.0;
  bool   crashed = false;
  if (active && r == 0 && c.ok) {
    const uint32_t nh = hit_n[u];
    if (nh) {
      const PosRecord me = rec[gi];
      if (hit_overflow && nh > (uint32_t)HITS) {  // more qualifying partners than the hit list holds: the reference path
        query_lane_sweeps2<LISTS>(me, c, gi, rec, n_total, mask, ib, head, next, crash, rebounce, fx, fy, fz, crashed);
      } else {
        uint32_t prev = 0;
        for (uint32_t h = 0; h < nh; h++) {  // selection by repeated minimum: nh <= 4, almost always 1
          uint32_t best = 0xFFFFFFFFu;
          for (uint32_t k = 0; k < nh; k++) {
            const uint32_t j = hit_j[u][k];
            if ((h == 0 || j > prev) && j < best) best = j;
          }
          apply_partner(me, rec[best], crash, rebounce, fx, fy, fz, crashed);
          prev = best;
        }
      }
    }
  }
  uint32_t cnt = nl_n[u];
  if (cnt > (uint32_t)LIST_CAP) {  // an incomplete list keeps the next tick in rebuild mode
    if (r == 0 && seated) {
      ctl[cur ^ 1] = 1u;
      atomicAdd(&ctl[6], 1u);  // statistics: UAVs over the list capacity
      if (stall_word) {        // a search queued in stream order: the fused launches behind it must not use the incomplete lists
        *stall_word = stall_tau;
        __hip_atomic_store(&hostw[CTL_STALL], stall_tau, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    cnt = 0;
  }
  if (active) {
    for (uint32_t k = (uint32_t)r; k < cnt; k += LPU) {  // ascending: an entry's row is the number of smaller ones (indices are distinct)
      const uint32_t j = nl_j[u][k];
      uint32_t       row = 0;
      for (uint32_t m = 0; m < cnt; m++) row += nl_j[u][m] < j ? 1u : 0u;
      nbr[(size_t)row * sw.n + i] = j;
    }
    if (r == 0) nbr_cnt[i] = cnt;
  }
  if (active && r == 0 && !muted) {
    sw.S[(size_t)(F_FEXT + 0) * sw.npad + i] = fx;
    sw.S[(size_t)(F_FEXT + 1) * sw.npad + i] = fy;
    sw.S[(size_t)(F_FEXT + 2) * sw.npad + i] = fz;
    if (crashed) sw.F[i] |= FLAG_CRASHED;
  }
#ifdef MRS_Q2_CLOCK
  MRS_Q2_STAMP(5, cnt);
  if (lane == 0 && active) {
    for (int k = 0; k < 6; k++) nbr[(size_t)k * sw.n + i] = ts[k];
    nbr_cnt[i] = 6;
  }
#endif
#undef MRS_Q2_STAMP
}

// lanes per UAV: three — every lane exactly three columns, 21 UAVs per wave, and the 4 762 / 5 953 waves of a 100 k swarm / a 125 k
// shard resident at once (24 blocks per CU: 106 SGPRs allow six waves per SIMD, 6.3 KB of LDS 24 blocks) — up to half a million UAVs;
// two beyond, where the waves come in rounds anyway.  Same box, 100 k / 125 k / 200 k / 1 M UAVs: two lanes 38.1 / 48.1 / 62.2 / 317 us
// per search, three 35.2 / 40.6 / 59.0 / 327, four 37.3 / 44.3 / 60.6 / 316 (MEASUREMENTS 6.6).
template <int LISTS>
static void launch_query2(int lpu, SwarmDev sw, const PosRecord* rec, long long n_total, long long my_offset, uint32_t mask, int ib, const uint2* head,
                          const uint2* next, int crash, double rebounce, uint32_t* ctl, int cur, int force, uint32_t* nbr, uint32_t* nbr_cnt,
                          uint32_t* stall_word, volatile uint32_t* hostw, uint32_t stall_tau, hipStream_t st) {
#define MRS_Q2_LAUNCH(L)                                                                                                                        \
  hipLaunchKernelGGL((k_query2<LISTS, L>), dim3((sw.n + 64 / L - 1) / (64 / L)), dim3(64), 0, st, sw, rec, n_total, my_offset, mask, ib, head, next, crash, \
                     rebounce, ctl, cur, force, nbr, nbr_cnt, stall_word, hostw, stall_tau)
  if (lpu == 1)
    MRS_Q2_LAUNCH(1);
  else if (lpu == 2)
    MRS_Q2_LAUNCH(2);
  else if (lpu == 3)
    MRS_Q2_LAUNCH(3);
  else
    MRS_Q2_LAUNCH(4);
#undef MRS_Q2_LAUNCH
}
static int query_lpu(long long n_own) {
  const char* e = getenv("MRS_QUERY_LPU");  // tuning aid, and how the tests reach every instantiation (read per search: a search is rare)
  const int   forced = e ? atoi(e) : 0;
  if (forced >= 1 && forced <= 4) return forced;
  return n_own > 500000 ? 2 : 3;
}

}  // namespace

struct CollideWork {
  long long cap_n = 0;
  uint32_t  cap_T = 0;
  int       cur   = 0;  // which head table the next tick fills; the other one is being wiped by that tick's query
  uint2 *   head[2] = {nullptr, nullptr}, *next = nullptr;
  // neighbour lists (single-GPU ticks)
  PosRecord* rec_build = nullptr;  // records of the last rebuild: reference positions of the skin test + airframe constants
  uint32_t * nbr = nullptr, *nbr_cnt = nullptr, *ctl = nullptr;
  int        fcur = 0;             // which of ctl[0..1] the next tick reads
  bool       lists_live = false;   // rec_build / nbr describe this swarm as of some earlier tick
  double*    g_bbox = nullptr;       // gathered mode: this rank's bounding box widened by the list radius (6 doubles)
  PosRecord* g_rec_build = nullptr;  // gathered mode: all records as of this rank's last rebuild
  long long  g_cap = 0;
  bool       g_lists_live = false;
  // halo exchange of a search tick (mrs_collide_halo_*): instead of every rank's ALL records, the records that can be within the list
  // radius of another rank's UAVs travel — [header | entries] of 64 B, the header's `j` = count, `pad` = flags
  HaloEntry* h_send = nullptr;   // [1 + h_cap]
  HaloEntry* h_recv = nullptr;   // [world][1 + h_cap]
  long long  h_cap = 0, h_alloc = 0;  // entries per block in use; entries allocated (all blocks together, headers included)
  int        h_world = 0;
  uint32_t*  h_ctl = nullptr;    // [0] entries appended, [1] flags (MRS_HALO_*)
  double*    g_box_out = nullptr;  // where a search of the export-set exchange also leaves its box (mrs_collide_set_box_out)
  double*    h_part = nullptr;     // partial boxes of k_halo_select, one per block
  long long  h_part_cap = 0;
  bool       g_export_form = false;  // the last gathered search was one of the export-set exchange: lists end up in slot form, and of
                                     // the record copy only this rank's own range (the skin references) is kept
  // fused step + collision evaluation (step_device.inc *_coll): double-buffered positions, control words, pinned host mirror
  // (three buffers: in a split sharded tick the interior launch of tick t+1 writes its output while the boundary launch of tick t
  //  still reads its input — with two buffers those would be the same array)
  Pos4*     P[3]  = {nullptr, nullptr, nullptr};
  int       pcur  = 0;        // P[pcur] holds the positions after the most recent step (when the host says they are valid)
  long long p_cap = 0;
  uint32_t* fctl  = nullptr;  // CTL_WORDS device words
  uint32_t* hostw = nullptr;  // CTL_WORDS pinned host words (stall, progress mirrored by the kernels)
  // export-set exchange (multi-GPU ticks between searches): own UAVs listed by another rank, their slots in the padded collective
  uint32_t*     exp_slot = nullptr;   // [n_local]
  long long     exp_slot_cap = 0;
  // split sharded ticks: class of every 64-UAV block, list of the boundary blocks, epoch word per block (swarm_layout.h)
  uint32_t *    blk_class = nullptr, *blk_list = nullptr, *epoch = nullptr;
  uint32_t*     host_heads = nullptr;  // pinned: heads of the slot maps + boundary-block count of the last search
  long long     blk_cap = 0;
  Pos4*         x_send = nullptr;     // [1 + x_cap]: header + exported positions of this rank
  Pos4*         x_recv = nullptr;     // [world][1 + x_cap]
  PartnerConst* x_const = nullptr;    // [world][1 + x_cap]
  long long     x_cap = 0;            // export slots per rank in the collective
  int           x_world = 0;
};

static void free_work(CollideWork* w) {
  (void)hipFree(w->head[0]); (void)hipFree(w->head[1]); (void)hipFree(w->next);
  (void)hipFree(w->rec_build); (void)hipFree(w->nbr); (void)hipFree(w->nbr_cnt); (void)hipFree(w->ctl); (void)hipFree(w->g_rec_build);
  (void)hipFree(w->g_bbox);
  w->g_bbox = nullptr;
  (void)hipFree(w->h_send); (void)hipFree(w->h_ctl); (void)hipFree(w->h_part);  // (h_recv lives in h_send's allocation)
  w->h_part = nullptr; w->h_part_cap = 0;
  w->h_send = w->h_recv = nullptr; w->h_ctl = nullptr; w->h_cap = w->h_alloc = 0;
  (void)hipFree(w->exp_slot); (void)hipFree(w->x_send);  // (x_recv and x_const live in x_send's allocation)
  w->exp_slot = nullptr; w->x_send = w->x_recv = nullptr; w->x_const = nullptr;
  w->exp_slot_cap = w->x_cap = 0;
  (void)hipFree(w->blk_class); (void)hipFree(w->blk_list); (void)hipFree(w->epoch);
  w->blk_class = w->blk_list = w->epoch = nullptr;
  w->blk_cap = 0;
  (void)hipFree(w->P[0]); (void)hipFree(w->P[1]); (void)hipFree(w->P[2]); (void)hipFree(w->fctl);
  if (w->hostw) (void)hipHostFree(w->hostw);
  if (w->host_heads) (void)hipHostFree(w->host_heads);
  w->host_heads = nullptr;
  w->P[0] = w->P[1] = w->P[2] = nullptr;
  w->p_cap = 0;
  w->fctl = w->hostw = nullptr;
  w->g_rec_build = nullptr;
  w->g_cap = 0;
  w->g_lists_live = false;
  w->head[0] = w->head[1] = w->next = nullptr;
  w->rec_build = nullptr;
  w->nbr = w->nbr_cnt = w->ctl = nullptr;
  w->lists_live = false;
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

// the mode word the collision code passes around as `crash` (collide_device.inc): the flavour of the force expression follows the swarm
static inline int mode_word(const SwarmDev& sw, int crash) { return (crash ? MRS_MODE_CRASH : 0) | (sw.fast ? MRS_MODE_FAST : 0); }

#define CK(e)                        \
  do {                               \
    hipError_t _e = (e);             \
    if (_e != hipSuccess) return _e; \
  } while (0)

// What the step kernel needs for the skin test (all null/zero while no neighbour lists are live).
extern "C" void mrs_collide_step_hook(const CollideWork* w, const PosRecord** rec, uint32_t** flag, double* lim2) {
  const bool on = w && w->lists_live;
  *rec  = on ? w->rec_build : nullptr;
  *flag = on ? w->ctl + w->fcur : nullptr;
  *lim2 = (0.5 * SKIN) * (0.5 * SKIN) * (1.0 - 1e-9);
}

// bits of a record index + 1 in an entry of the list-building format (2^ib > n_total; the cell tag keeps the other 32 - ib)
static inline int index_bits(long long n_total) {
  int ib = 1;
  while (ib < 31 && (1ll << ib) <= n_total) ib++;
  return ib;
}

static hipError_t ensure_tables(CollideWork* w, long long n_total, hipStream_t st) {
  uint32_t T = 1024;
  while ((long long)T < MRS_TABLE_FACTOR * n_total) T <<= 1;  // load factor <= 0.25 (2x and 8x tables measured slower: more chains / more misses)
  if (n_total > w->cap_n || T > w->cap_T) {
    CK(hipStreamSynchronize(st));
    free_work(w);
    CK(hipMalloc(&w->head[0], sizeof(uint2) * ((size_t)T + 1)));  // (+1: k_query2 reads two entries at a time, the spare one stays empty)
    CK(hipMalloc(&w->head[1], sizeof(uint2) * ((size_t)T + 1)));
    CK(hipMalloc(&w->next, sizeof(uint2) * (size_t)n_total));
    CK(hipMemsetAsync(w->head[0], 0, sizeof(uint2) * ((size_t)T + 1), st));  // afterwards every query wipes the table of the next tick
    CK(hipMemsetAsync(w->head[1], 0, sizeof(uint2) * ((size_t)T + 1), st));
    w->cap_n = n_total;
    w->cap_T = T;
    w->cur   = 0;
  }
  return hipSuccess;
}

// buffers of the fused step + collision evaluation for n local UAVs
static hipError_t ensure_fused(CollideWork* w, long long n, hipStream_t st) {
  if (!w->fctl) {
    CK(hipMalloc(&w->fctl, sizeof(uint32_t) * CTL_WORDS));
    CK(hipMemsetAsync(w->fctl, 0, sizeof(uint32_t) * CTL_WORDS, st));
    CK(hipHostMalloc(&w->hostw, sizeof(uint32_t) * CTL_WORDS, hipHostMallocMapped | hipHostMallocCoherent));
    for (int k = 0; k < CTL_WORDS; k++) w->hostw[k] = 0u;
  }
  if (n > w->p_cap) {
    CK(hipStreamSynchronize(st));
    (void)hipFree(w->P[0]); (void)hipFree(w->P[1]); (void)hipFree(w->P[2]);
    CK(hipMalloc(&w->P[0], sizeof(Pos4) * (size_t)n));
    CK(hipMalloc(&w->P[1], sizeof(Pos4) * (size_t)n));
    CK(hipMalloc(&w->P[2], sizeof(Pos4) * (size_t)n));
    w->p_cap = n;
    w->pcur  = 0;
  }
  return hipSuccess;
}

// Plain search every tick over ready (gathered) records — the multi-GPU path, and single-GPU ticks with lists switched off
// (rec_is_local_scratch: `rec` is this swarm's own, not yet packed, record buffer; pack and insert are then fused).
extern "C" hipError_t mrs_collide_run(SwarmDev sw, CollideWork** work, const PosRecord* rec, long long n_total, long long my_offset,
                                      int crash, double rebounce, int rec_is_local_scratch, hipStream_t st) {
  if (!*work) *work = new CollideWork();
  CollideWork* w = *work;
  crash = mode_word(sw, crash);
  CK(ensure_tables(w, n_total, st));
  w->lists_live = false;  // the step kernel stops testing; a later list tick starts with a rebuild
  w->g_lists_live = false;
  const uint32_t T = w->cap_T, mask = T - 1;  // a larger table from an earlier call is still valid (both are empty between ticks)
  const unsigned gN   = (unsigned)((n_total + 255) / 256);
  uint2*         head = w->head[w->cur];
  uint2*         other = w->head[w->cur ^ 1];
  w->cur ^= 1;
  if (rec_is_local_scratch)
    hipLaunchKernelGGL(k_pack_insert<false>, dim3(gN), dim3(256), 0, st, sw, const_cast<PosRecord*>(rec), mask, head, w->next, nullptr, 0, 1, 0,
                       nullptr, 0u, nullptr, nullptr, 0, 0.0, nullptr, nullptr, 0);
  else
    hipLaunchKernelGGL(k_insert<false>, dim3(gN), dim3(256), 0, st, rec, n_total, mask, head, w->next);
  hipLaunchKernelGGL(k_query, dim3((sw.n + 63) / 64), dim3(64), 0, st, sw, rec, n_total, my_offset, mask, head, w->next, other, T, crash, rebounce);
  return hipGetLastError();
}

// Single-GPU tick with neighbour lists.  force_rebuild: the host changed positions or airframe constants since the last call.
// number of list rebuilds so far (device counter; synchronises the stream)
// debugging aid (not part of the public header): the eight control words
extern "C" hipError_t mrs_collide_debug_words(const CollideWork* w, hipStream_t st, unsigned* out8) {
  for (int k = 0; k < 8; k++) out8[k] = 0;
  if (!w || !w->ctl) return hipSuccess;
  CK(hipMemcpyAsync(out8, w->ctl, 8 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
  return hipStreamSynchronize(st);
}

extern "C" void mrs_collide_list_geometry(int* list_cap, double* list_radius) {
  *list_cap    = LIST_CAP;
  *list_radius = SQRT3_UP + SKIN;
}
// test hook: the lists of the last single-GPU search (synchronises the stream)
extern "C" hipError_t mrs_collide_copy_lists(const CollideWork* w, long long n, uint32_t* count, uint32_t* nbr, int rows, hipStream_t st) {
  if (!w || !w->nbr || !w->lists_live) return hipErrorInvalidValue;
  CK(hipMemcpyAsync(count, w->nbr_cnt, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
  for (int r = 0; r < rows && r < LIST_CAP; r++)  // (row r of the device array starts at r * n: the lists are stored row-major over the swarm's n)
    CK(hipMemcpyAsync(nbr + (size_t)r * (size_t)n, w->nbr + (size_t)r * (size_t)n, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToHost, st));
  return hipStreamSynchronize(st);
}

extern "C" hipError_t mrs_collide_rebuilds(const CollideWork* w, hipStream_t st, unsigned* out) {
  *out = 0;
  if (!w || !w->ctl) return hipSuccess;
  CK(hipMemcpyAsync(out, w->ctl + 2, sizeof(unsigned), hipMemcpyDeviceToHost, st));
  return hipStreamSynchronize(st);
}

// guard_tau != 0: the pass is queued in stream order behind fused step launches (tick index of the last one = guard_tau): it does
// nothing if those have stalled, and stalls what follows if the new lists come out incomplete
extern "C" hipError_t mrs_collide_run_lists(SwarmDev sw, CollideWork** work, int crash, double rebounce, int force_rebuild, unsigned guard_tau,
                                            hipStream_t st) {
  if (!*work) *work = new CollideWork();
  CollideWork*    w = *work;
  const long long n = sw.n;
  crash = mode_word(sw, crash);
  CK(ensure_tables(w, n, st));
  if (!w->rec_build) CK(hipMalloc(&w->rec_build, sizeof(PosRecord) * (size_t)w->cap_n));
  if (!w->nbr) {
    CK(hipMalloc(&w->nbr, sizeof(uint32_t) * (size_t)LIST_CAP * (size_t)w->cap_n));
    CK(hipMalloc(&w->nbr_cnt, sizeof(uint32_t) * (size_t)w->cap_n));
    CK(hipMemsetAsync(w->nbr, 0, sizeof(uint32_t) * (size_t)LIST_CAP * (size_t)w->cap_n, st));  // rows beyond a UAV's count are read (not used)
    CK(hipMemsetAsync(w->nbr_cnt, 0, sizeof(uint32_t) * (size_t)w->cap_n, st));
    CK(hipMalloc(&w->ctl, sizeof(uint32_t) * 8));  // [0..1] skin flags, [2] rebuild counter, [4..5] "head table t holds entries"
    CK(hipMemsetAsync(w->ctl, 0, sizeof(uint32_t) * 8, st));
    w->fcur = 0;
  }
  CK(ensure_fused(w, w->cap_n, st));
  if (!w->lists_live) {  // first list tick, or plain-search ticks came in between: start from empty tables and flags
    CK(hipMemsetAsync(w->head[0], 0, sizeof(uint2) * (size_t)w->cap_T, st));
    CK(hipMemsetAsync(w->head[1], 0, sizeof(uint2) * (size_t)w->cap_T, st));
    CK(hipMemsetAsync(w->ctl, 0, sizeof(uint32_t) * 8, st));
    w->fcur = 0;
  }
  const int force = (force_rebuild || !w->lists_live) ? 1 : 0;
  const uint32_t T = w->cap_T, mask = T - 1;
  const int      tid  = w->cur;
  uint2*         head = w->head[tid];
  uint2*         other = w->head[tid ^ 1];
  w->cur ^= 1;
  const int ib = index_bits(n);
  hipLaunchKernelGGL(k_pack_insert<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, sw, w->rec_build, mask, head, w->next, w->ctl,
                     w->fcur, force, tid, other, T, w->nbr, w->nbr_cnt, crash, rebounce, w->P[w->pcur], guard_tau ? w->fctl + CTL_STALL : nullptr, ib);
  launch_query2<1>(query_lpu(n), sw, w->rec_build, n, 0ll, mask, ib, head, w->next, crash, rebounce, w->ctl, w->fcur, force, w->nbr, w->nbr_cnt,
                   guard_tau ? w->fctl + CTL_STALL : nullptr, w->hostw, guard_tau, st);
  w->fcur ^= 1;  // steps launched from now on report into the flag the next tick reads
  w->lists_live = true;
  w->g_lists_live = false;
  return hipGetLastError();
}

// Multi-GPU tick with neighbour lists: `rec` = the gathered current records of all ranks (NaN padding included).
extern "C" hipError_t mrs_collide_run_lists_gathered(SwarmDev sw, CollideWork** work, const PosRecord* rec, long long n_total, long long my_offset,
                                                     int crash, double rebounce, int force_rebuild, hipStream_t st) {
  if (!*work) *work = new CollideWork();
  CollideWork* w = *work;
  crash = mode_word(sw, crash);
  CK(ensure_tables(w, n_total, st));
  if (!w->nbr) {
    CK(hipMalloc(&w->nbr, sizeof(uint32_t) * (size_t)LIST_CAP * (size_t)w->cap_n));
    CK(hipMalloc(&w->nbr_cnt, sizeof(uint32_t) * (size_t)w->cap_n));
    CK(hipMemsetAsync(w->nbr, 0, sizeof(uint32_t) * (size_t)LIST_CAP * (size_t)w->cap_n, st));
    CK(hipMemsetAsync(w->nbr_cnt, 0, sizeof(uint32_t) * (size_t)w->cap_n, st));
    CK(hipMalloc(&w->ctl, sizeof(uint32_t) * 8));
    CK(hipMemsetAsync(w->ctl, 0, sizeof(uint32_t) * 8, st));
  }
  if (n_total > w->g_cap) {
    CK(hipStreamSynchronize(st));
    (void)hipFree(w->g_rec_build);
    CK(hipMalloc(&w->g_rec_build, sizeof(PosRecord) * (size_t)n_total));
    w->g_cap        = n_total;
    w->g_lists_live = false;
  }
  const bool export_form = force_rebuild != 0;  // (the export-set exchange searches on its own decision, and only then comes here)
  if (!w->g_lists_live || w->g_export_form != export_form) {
    // first gathered search, or other modes came in between: empty tables and flags, rebuild.  Consecutive searches of the export-set
    // exchange skip this: the two head tables keep wiping each other, and nobody compares against the foreign part of the record copy.
    CK(hipMemsetAsync(w->head[0], 0, sizeof(uint2) * (size_t)w->cap_T, st));
    CK(hipMemsetAsync(w->head[1], 0, sizeof(uint2) * (size_t)w->cap_T, st));
    CK(hipMemsetAsync(w->ctl, 0, sizeof(uint32_t) * 8, st));
    if (!export_form) CK(hipMemsetAsync(w->g_rec_build, 0xFF, sizeof(PosRecord) * (size_t)w->g_cap, st));  // NaN records
    w->fcur = 0;
    w->g_lists_live = false;
  }
  const int      force = (w->g_lists_live && !force_rebuild) ? 0 : 1;
  const uint32_t T = w->cap_T, mask = T - 1;
  const int      tid  = w->cur;
  uint2*         head = w->head[tid];
  uint2*         other = w->head[tid ^ 1];
  w->cur ^= 1;
  const unsigned gN = (unsigned)((n_total + 255) / 256);
  const int      ib = index_bits(n_total);
  const double   lim2 = (0.5 * SKIN2) * (0.5 * SKIN2) * (1.0 - 1e-9);
  if (!w->g_bbox) CK(hipMalloc(&w->g_bbox, sizeof(double) * 6 * (BBOX_BLOCKS + 1)));  // the box, then the partial boxes
  if (force) {
    // (the search is decided: the comparison of all records with those of the last search would only cost time — 14 us at 1 M records;
    //  its other job, clearing the flag word the query may raise, is done by k_own_bbox_final)
  } else {
    hipLaunchKernelGGL(k_skin_gathered, dim3(gN), dim3(256), 0, st, rec, w->g_rec_build, n_total, w->ctl, w->fcur, lim2);
  }
  // (also on list ticks: the device may decide on a search)
  hipLaunchKernelGGL(k_own_bbox_part, dim3(BBOX_BLOCKS), dim3(256), 0, st, rec, my_offset, sw.n, w->g_bbox + 6, w->ctl, w->fcur, force);
  hipLaunchKernelGGL(k_own_bbox_final, dim3(1), dim3(256), 0, st, w->g_bbox + 6, BBOX_BLOCKS, SQRT3_UP + SKIN2 + 1e-6, w->g_bbox, w->ctl, w->fcur, force,
                     export_form ? w->g_box_out : nullptr);
  hipLaunchKernelGGL(k_insert_gathered_lists, dim3(gN), dim3(256), 0, st, sw, rec, w->g_rec_build, n_total, my_offset, mask, head, w->next, w->ctl,
                     w->fcur, force, tid, other, T, w->nbr, w->nbr_cnt, crash, rebounce, w->g_bbox, export_form ? 1 : 0, ib);
  launch_query2<2>(query_lpu(sw.n), sw, rec, n_total, my_offset, mask, ib, head, w->next, crash, rebounce, w->ctl, w->fcur, force, w->nbr, w->nbr_cnt,
                   nullptr, nullptr, 0u, st);
  w->fcur ^= 1;
  w->g_lists_live  = true;
  w->g_export_form = export_form;
  w->lists_live    = false;  // the local-mode skin hook of the step kernel is off
  return hipGetLastError();
}

// ---- halo exchange of a search tick (sharded swarms, export-set exchange) ----
// A search needs, on every rank, the records of all UAVs within the list radius of one of its own.  The full exchange gathers
// EVERYTHING (48 B x n_total per rank: 42 MB received at 8 x 125 000) and lets every rank scan it.  With spatially coherent shards
// almost none of it matters: rank q can only list a record that lies in q's bounding box widened by the list radius.  Every rank knows
// every rank's box of the LAST search (it travels at the tail of the slot-map blocks).  Rank q's new box lies inside its old one grown
// by `margin` as long as none of q's UAVs is more than `margin` outside the hull of their old positions — q CHECKS exactly that on its
// own records (MRS_HALO_MOVED otherwise: a search is due when somebody has moved SKIN2/2, so a margin of SKIN2 is rarely left; a host
// write can).  So a rank sends the records inside some other rank's old box grown by `margin`, tagged with their local index; receivers write them into the
// SAME table the full gather would have filled (record of UAV j of rank q at q x n_max + j — lists, export marks and translation
// keep their global indices) and insert them, and their own records, into the hash.  Any rank's flag (moved, more entries than the
// block holds) makes every rank repeat the search with the full exchange (tick_sharded.hip).
#define MRS_HALO_MOVED    1u  // an own record lies outside this rank's box of the last search widened by the motion margin
#define MRS_HALO_OVERFLOW 2u  // more entries than the block holds
namespace {
__device__ __forceinline__ const double* halo_box(const uint32_t* maps, long long stride, int boxw, int q) {
  return reinterpret_cast<const double*>(maps + (size_t)q * (size_t)stride + (size_t)boxw);
}
// ONE pass over the rank's own UAVs: record from the state (what k_pack_positions writes) -> own part of the table; the records some
// other rank may list -> the send block; the hull of the block's records -> part[block] (k_own_bbox_final reduces them).
// (The header is a launch of its own, k_halo_header: written by "the block that finishes last" it needed a device-scope release in
//  every wave — an L2 write-back each on this chip — and the kernel took 44.6 us instead of 8.)
__global__ void __launch_bounds__(256) k_halo_select(SwarmDev sw, PosRecord* table_own, const uint32_t* maps, long long stride, int boxw, int world, int rank,
                                                     double margin, double own_margin, HaloEntry* send, unsigned hcap, uint32_t* hctl, double* part) {
  __shared__ double lo[3][256], hi[3][256];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double    l[3] = {1e300, 1e300, 1e300}, h[3] = {-1e300, -1e300, -1e300};
  if (i < sw.n) {
    const TypeParams& P = sw.T[sw.F[i] >> FLAG_TYPE_SHIFT];
    PosRecord         r;
    r.x = sw.S[(size_t)(F_X + 0) * sw.npad + i];
    r.y = sw.S[(size_t)(F_X + 1) * sw.npad + i];
    r.z = sw.S[(size_t)(F_X + 2) * sw.npad + i];
    r.mass        = P.mass;
    r.arm_length  = P.arm_length;
    r.prop_radius = P.prop_radius;
    table_own[i]  = r;
    if (record_usable(r)) {
      l[0] = h[0] = r.x; l[1] = h[1] = r.y; l[2] = h[2] = r.z;
      bool wanted = false;
      for (int q = 0; q < world; q++) {
        const double* b  = halo_box(maps, stride, boxw, q);  // the box of q's last search: hull of its records then, widened by the list radius
        const double  m  = q == rank ? own_margin : margin;  // (own_margin = margin - that widening: this record within `margin` of the old hull)
        const bool    in = r.x >= b[0] - m && r.y >= b[1] - m && r.z >= b[2] - m && r.x <= b[3] + m && r.y <= b[4] + m && r.z <= b[5] + m;
        if (q == rank) {
          if (!in) atomicOr(&hctl[1], MRS_HALO_MOVED);  // (also when this rank had no usable record then: nothing is inside its box)
        } else if (in) {
          wanted = true;
        }
      }
      if (wanted) {
        const uint32_t k = atomicAdd(&hctl[0], 1u);
        if (k >= hcap) {
          atomicOr(&hctl[1], MRS_HALO_OVERFLOW);
        } else {
          HaloEntry e;
          e.x = r.x; e.y = r.y; e.z = r.z; e.mass = r.mass; e.arm_length = r.arm_length; e.prop_radius = r.prop_radius;
          e.j = (unsigned long long)i;
          e.pad = 0ull;
          send[1u + k] = e;
        }
      }
    }
  }
  bbox_reduce_block(l, h, lo, hi);
  if (threadIdx.x < 3) {
    part[blockIdx.x * 6 + threadIdx.x]     = lo[threadIdx.x][0];
    part[blockIdx.x * 6 + 3 + threadIdx.x] = hi[threadIdx.x][0];
  }
}
__global__ void k_halo_header(HaloEntry* send, uint32_t* hctl, unsigned force_flags) {
  HaloEntry hd;
  hd.x = hd.y = hd.z = hd.mass = hd.arm_length = hd.prop_radius = 0.0;
  hd.j   = hctl[0];  // (the number WANTED: more than the block holds with MRS_HALO_OVERFLOW — the capacity the repeat needs)
  hd.pad = hctl[1] | force_flags;
  send[0] = hd;
  hctl[0] = hctl[1] = 0u;  // (ready for the next search)
}
// own records and the received entries into the hash (the insert of k_insert_gathered_lists without the scan of n_total records)
__global__ void __launch_bounds__(256) k_halo_insert(SwarmDev sw, PosRecord* table, PosRecord* rec_build, const HaloEntry* recv, int world, int rank, unsigned hcap,
                                                     long long n_max, uint32_t mask, uint2* head, uint2* next, uint32_t* ctl, int table_id, uint2* head_to_clear,
                                                     uint32_t table_size, const double* bb, int ib) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) {
    ctl[4 + table_id] = 1u;
    ctl[2] += 1u;
  }
  if (ctl[4 + (table_id ^ 1)]) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t u = (uint32_t)t; u < table_size; u += stride) head_to_clear[u] = make_uint2(0u, 0u);
  }
  long long g;
  PosRecord r;
  if (t < sw.n) {
    g            = (long long)rank * n_max + t;
    r            = table[g];
    rec_build[g] = r;  // the reference of the skin tests (own range only, as in the export form of the full exchange)
  } else {
    const long long e = t - sw.n;
    const int       q = (int)(e / hcap);
    const unsigned  k = (unsigned)(e - (long long)q * hcap);
    if (q >= world || q == rank) return;
    const HaloEntry* blk = recv + (size_t)q * (size_t)(1u + hcap);
    if ((unsigned long long)k >= blk[0].j) return;  // (k < hcap: an overflowing block is read as far as it goes, the search is repeated anyway)
    const HaloEntry en = blk[1u + k];
    if (en.j >= (unsigned long long)n_max) return;  // (never: a block that is not one of ours)
    g = (long long)q * n_max + (long long)en.j;
    r.x = en.x; r.y = en.y; r.z = en.z; r.mass = en.mass; r.arm_length = en.arm_length; r.prop_radius = en.prop_radius;
    table[g] = r;
  }
  uint32_t qc;
  Cell     c = cell_q<2>(r.x, r.y, r.z, qc);
  c.ok = c.ok && r.x >= bb[0] && r.y >= bb[1] && r.z >= bb[2] && r.x <= bb[3] && r.y <= bb[4] && r.z <= bb[5];
  insert_uav2(g, c, qc, mask, ib, head, next);
}
}  // namespace

extern "C" hipError_t mrs_collide_halo_prepare(CollideWork** work, int world, long long cap, hipStream_t st) {
  if (!*work) *work = new CollideWork();
  CollideWork* w = *work;
  if (cap < 1 || world < 1) return hipErrorInvalidValue;
  const long long need = (cap + 1) * (long long)(1 + world);
  if (need > w->h_alloc) {
    CK(hipStreamSynchronize(st));
    (void)hipFree(w->h_send);
    w->h_send = nullptr;
    w->h_alloc = 0;
    const long long alloc = need + need / 4;
    CK(hipMalloc(&w->h_send, sizeof(HaloEntry) * (size_t)alloc));
    CK(hipMemsetAsync(w->h_send, 0, sizeof(HaloEntry) * (size_t)alloc, st));
    w->h_alloc = alloc;
  }
  w->h_recv  = w->h_send + (cap + 1);  // (a collective writes every block whole: what another capacity left behind is overwritten)
  w->h_cap   = cap;
  w->h_world = world;
  if (!w->h_ctl) {
    CK(hipMalloc(&w->h_ctl, sizeof(uint32_t) * 4));
    CK(hipMemsetAsync(w->h_ctl, 0, sizeof(uint32_t) * 4, st));
  }
  return hipSuccess;
}
extern "C" long long mrs_collide_halo_capacity(const CollideWork* w) { return w ? w->h_cap : 0; }
extern "C" void*     mrs_collide_halo_send(const CollideWork* w) { return w ? (void*)w->h_send : nullptr; }
extern "C" void*     mrs_collide_halo_recv(const CollideWork* w) { return w ? (void*)w->h_recv : nullptr; }

// own records (packed into `own`) -> the table's own part + the send block; `maps` = every rank's slot-map block of the LAST search
// (its tail holds that rank's widened box), `margin` = how far a UAV may be from where it was then
// not_ready: this rank cannot search on a halo (its tables are not the ones a full search of the export-set exchange left) — its header says so
extern "C" hipError_t mrs_collide_halo_select(SwarmDev sw, CollideWork* w, PosRecord* table, long long n_max, int rank, int world, const uint32_t* maps,
                                              long long stride, int boxw, int not_ready, hipStream_t st) {
  if (!w || !w->h_send) return hipErrorInvalidValue;
  const double    margin = SKIN2, widening = SQRT3_UP + SKIN2 + 1e-6;  // (the widening of k_own_bbox_final)
  const long long blocks = sw.n > 0 ? (sw.n + 255) / 256 : 1;
  if (blocks > w->h_part_cap) {
    CK(hipStreamSynchronize(st));
    (void)hipFree(w->h_part);
    w->h_part = nullptr;
    w->h_part_cap = 0;
    CK(hipMalloc(&w->h_part, sizeof(double) * 6 * (size_t)blocks));
    w->h_part_cap = blocks;
  }
  hipLaunchKernelGGL(k_halo_select, dim3((unsigned)blocks), dim3(256), 0, st, sw, table + (size_t)rank * (size_t)n_max, maps, stride, boxw, world, rank, margin,
                     margin - widening - 1e-9, w->h_send, (unsigned)w->h_cap, w->h_ctl, w->h_part);
  hipLaunchKernelGGL(k_halo_header, dim3(1), dim3(1), 0, st, w->h_send, w->h_ctl, not_ready ? MRS_HALO_MOVED : 0u);
  return hipGetLastError();
}
// where the searches of the export-set exchange leave their box for the other ranks (the tail of this rank's slot-map block; null: nowhere)
extern "C" void mrs_collide_set_box_out(CollideWork** work, double* box_out) {
  if (!*work) *work = new CollideWork();
  (*work)->g_box_out = box_out;
}
extern "C" int mrs_collide_halo_ready(const CollideWork* w, long long n_total) {
  return w && w->nbr && w->g_rec_build && n_total <= w->g_cap && w->g_lists_live && w->g_export_form && w->g_bbox ? 1 : 0;
}

// the search itself on the table the halo exchange has filled: own box, insert (own records + received entries), list-building query
extern "C" hipError_t mrs_collide_run_lists_halo(SwarmDev sw, CollideWork** work, PosRecord* table, long long n_total, long long n_max, int rank, int world,
                                                 int crash, double rebounce, hipStream_t st) {
  if (!*work) return hipErrorInvalidValue;
  CollideWork* w = *work;
  if (!w->h_recv || !mrs_collide_halo_ready(w, n_total)) return hipErrorInvalidValue;  // (a full search came first)
  crash = mode_word(sw, crash);
  CK(ensure_tables(w, n_total, st));
  const uint32_t T = w->cap_T, mask = T - 1;
  const int      tid = w->cur;
  w->cur ^= 1;
  const int       ib = index_bits(n_total);
  const long long my_offset = (long long)rank * n_max;
  hipLaunchKernelGGL(k_own_bbox_final, dim3(1), dim3(256), 0, st, w->h_part, (int)((sw.n + 255) / 256), SQRT3_UP + SKIN2 + 1e-6, w->g_bbox, w->ctl, w->fcur, 1,
                     w->g_box_out);  // (the partial boxes: k_halo_select)
  const long long threads = (long long)sw.n + (long long)world * w->h_cap;
  hipLaunchKernelGGL(k_halo_insert, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, sw, table, w->g_rec_build, w->h_recv, world, rank, (unsigned)w->h_cap, n_max,
                     mask, w->head[tid], w->next, w->ctl, tid, w->head[tid ^ 1], T, w->g_bbox, ib);
  launch_query2<2>(query_lpu(sw.n), sw, table, n_total, my_offset, mask, ib, w->head[tid], w->next, crash, rebounce, w->ctl, w->fcur, 1, w->nbr, w->nbr_cnt, nullptr,
                   nullptr, 0u, st);
  w->fcur ^= 1;
  return hipGetLastError();
}
extern "C" hipError_t mrs_collide_fused_dev(const SwarmDev* sw, CollideWork* w, unsigned tau, int eval, int crash, double rebounce, CollDev* cd) {
  if (!w || !w->lists_live || !w->fctl || !w->P[0]) return hipErrorInvalidValue;
  memset(cd, 0, sizeof *cd);
  cd->nbr      = w->nbr;
  cd->nbr_cnt  = w->nbr_cnt;
  cd->rec      = w->rec_build;
  cd->p_in     = w->P[w->pcur];
  cd->p_out    = w->P[(w->pcur + 1) % 3];
  cd->ctl      = w->fctl;
  cd->hostw    = w->hostw;
  cd->rebounce = rebounce;
  cd->lim2     = (0.5 * SKIN) * (0.5 * SKIN) * (1.0 - 1e-9);
  cd->lim2_warn = cd->lim2 * (WARN_FRACTION * WARN_FRACTION);
  cd->tau      = tau;
  cd->n        = sw->n;
  cd->eval     = eval;
  cd->crash    = crash;
  cd->world    = 1;
  cd->block    = 0;
  cd->write_force = 0;  // the host re-derives the force when it is asked for (mrs_collide_latch_force)
  return hipSuccess;
}
extern "C" void mrs_collide_fused_advance(CollideWork* w) { w->pcur = (w->pcur + 1) % 3; }

// pinned host mirror of the stall / progress words (read without synchronising: the kernels store them with system scope)
extern "C" const volatile unsigned* mrs_collide_host_words(const CollideWork* w) { return w ? w->hostw : nullptr; }

// forget a stall (the host has synchronised the stream and is about to repeat the search)
extern "C" hipError_t mrs_collide_fused_reset(CollideWork* w, hipStream_t st) {
  if (!w || !w->fctl) return hipSuccess;
  CK(hipMemsetAsync(w->fctl, 0, sizeof(uint32_t) * 2, st));                                   // stall, progress
  CK(hipMemsetAsync(w->fctl + CTL_WARN, 0, sizeof(uint32_t), st));  // tau restarts at 1: a stale warning index would swallow the next warning of that index
  w->hostw[CTL_STALL]    = 0u;
  w->hostw[CTL_PROGRESS] = 0u;
  w->hostw[CTL_WARN]     = 0u;
  w->hostw[CTL_STALL2] = w->hostw[CTL_WARN2] = 0u;
  return hipSuccess;
}

// ================================================================================================================================
// Export-set exchange (multi-GPU): between two searches a rank only needs the positions of the FOREIGN UAVs its neighbour lists
// name, and only has to publish the own UAVs some other rank lists.  Those two sets mirror each other — "j within the list radius
// of i" is decided by the same squared distance on both sides, from the same gathered records — so a rank finds its export set in
// its own lists: own UAV i is exported iff its list holds a foreign UAV.  A search tick (full gather, mrs_collide_run_lists_gathered)
// is followed by
//   k_export_mark      : export slot e_i for every own UAV with a foreign neighbour (any injective numbering will do)
//   all-gather         : the slot maps of all ranks (4 B per UAV), headed by each rank's count
//   k_export_translate : list entries (global record slots) -> local UAV index | FOREIGN + slot in the padded export collective;
//                        airframe constants and search-time positions of the foreign partners are copied next to those slots
// and every tick until the next search all-gathers 32 B per EXPORTED UAV instead of 48 B per UAV.
// ================================================================================================================================
namespace {

__global__ void k_fill_positions(SwarmDev sw, Pos4* pos_now) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sw.n) return;
  const size_t np = (size_t)sw.npad;
  const Pos4   pp = {sw.S[(size_t)(F_X + 0) * np + i], sw.S[(size_t)(F_X + 1) * np + i], sw.S[(size_t)(F_X + 2) * np + i],
                     (double)(sw.F[i] >> FLAG_TYPE_SHIFT)};
  pos_now[i]      = pp;
}

// one launch: control words (the error word stays: it is reported at the end of the call), slot map (padding UAVs: no slot), block
// classes, and the export allocation — send block, gathered blocks, partner constants — zeroed (headers!)
__global__ void k_search_reset(uint32_t* fctl, uint32_t* map, long long n_map, uint32_t* blk_class, int n_blocks, uint4* xalloc, long long n_xvec) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_map) map[i] = MRS_NO_SLOT;
  if (i < n_blocks) blk_class[i] = 0u;
  if (i < CTL_WORDS && i != CTL_ERROR && i != CTL_BADSLOT) fctl[i] = 0u;  // (sticky: the host reads both once per call, a call may hold several searches)
  for (long long v = i; v < n_xvec; v += (long long)gridDim.x * blockDim.x) xalloc[v] = make_uint4(0u, 0u, 0u, 0u);
}

// map: [0] export count of this rank, [1] lanes over the list capacity so far, [2 + i] slot of own UAV i
// ... and the position records of the UAVs as the search found them (what the first fused launch after the search reads)
// ... and the displacement bound on the state the search found (pred_hdt >= 0): the lists are new, every UAV sits on its reference
// position — if nobody can leave its skin within MRS_PRED_HORIZON steps, no stall index <= MRS_PRED_HORIZON can exist and the ticks
// right after the search need no serial phase (CTL_PRED, sent to every rank with the head of the slot map)
__global__ void k_export_mark(SwarmDev sw, Pos4* pos_now, long long n_max, int rank, const uint32_t* nbr, const uint32_t* nbr_cnt, uint32_t* exp_slot,
                              uint32_t* map, uint32_t* fctl, uint32_t* blk_class, double pred_hdt, double pred_lim, double rebounce) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = sw.n;
  if (i >= n) return;
  const size_t   np = (size_t)sw.npad;
  const uint32_t fl = sw.F[i];
  double         y[18];
#pragma unroll
  for (int c = 0; c < 18; c++) y[c] = (pred_hdt >= 0.0 || c < 3) ? sw.S[(size_t)(c < 6 ? F_X + c : (c < 15 ? F_R + (c - 6) : F_W + (c - 15))) * np + i] : 0.0;
  {
    const Pos4 pp = {y[0], y[1], y[2], (double)(fl >> FLAG_TYPE_SHIFT)};
    pos_now[i]    = pp;
  }
  const uint32_t cnt = nbr_cnt[i];
  if (pred_hdt >= 0.0) {
    const TypeParams& P = sw.T[fl >> FLAG_TYPE_SHIFT];
    double            thrust = 0.0;  // allocation * rpm^2 with the motor speeds as they are (multirotor_model.hpp:332-335)
    for (int m = 0; m < P.n_motors; m++) {
      const double r = sw.S[(size_t)(F_RPM + m) * np + i];
      thrust += P.alloc[3 * MRS_MAXM + m] * (r * r);
    }
    const bool   takeoff = (fl & FLAG_TAKEOFF) != 0u;
    const double init_z  = takeoff ? sw.S[(size_t)F_INITZ * np + i] : 0.0;
    const bool   usable  = mrs_pos_usable(y[0], y[1], y[2]);
    if (usable && mrs_may_leave(y, 0.0, thrust, cnt, rebounce, pred_hdt, pred_lim, P.pred_a0, P.pred_thr, P.pred_drag, P.ground_enabled, P.ground_z, takeoff, init_z))
      fctl[CTL_PRED] = 1u;  // (same value from every lane that finds one)
  }
  bool           exported = false;
  for (uint32_t k = 0; k < cnt; k++) {
    const uint32_t g = nbr[(size_t)k * (size_t)n + (size_t)i];
    if ((long long)g / n_max != (long long)rank) exported = true;
  }
  uint32_t e = MRS_NO_SLOT;
  if (exported) e = atomicAdd(&fctl[CTL_EXPORTS], 1u);
  exp_slot[i] = e;
  map[2 + i]  = e;
  if (exported) atomicOr(&blk_class[i >> 6], MRS_BLK_BOUNDARY);  // the block is stepped by the boundary launch of a split tick
}

// Behind the marking launch (its block classes are complete): the boundary blocks in a list (any order) and their number; the interior
// blocks that list a UAV of a boundary block (MRS_BLK_LAYER1: they wait for that block's epoch word in a split tick) and their number
// (CTL_NL1: the residency bound of the split form); the head of the rank's slot map — export count (final), bit 31: some own UAV may
// leave its skin within the horizon; lanes over the list capacity.  One thread per own UAV; list entries are still global record slots.
__global__ void k_class_list(int n, long long n_max, int rank, const uint32_t* nbr, const uint32_t* nbr_cnt, uint32_t* blk_class, uint32_t* blk_list,
                             uint32_t* fctl, uint32_t* map, const uint32_t* ctl) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    map[0] = fctl[CTL_EXPORTS] | (fctl[CTL_PRED] ? 0x80000000u : 0u);
    map[1] = ctl[6];  // lanes over the list capacity (cumulative, collide.hip k_query)
  }
  if (i >= n) return;
  const uint32_t mine = blk_class[i >> 6];
  if ((i & 63) == 0 && (mine & MRS_BLK_BOUNDARY)) blk_list[atomicAdd(&fctl[CTL_NBND], 1u)] = (uint32_t)(i >> 6);
  if (mine & MRS_BLK_BOUNDARY) return;
  const uint32_t cnt = nbr_cnt[i];
  bool           l1  = false;
  for (uint32_t k = 0; k < cnt; k++) {
    const long long g = (long long)nbr[(size_t)k * (size_t)n + (size_t)i], q = g / n_max;
    if (q == (long long)rank && (blk_class[(g - q * n_max) >> 6] & MRS_BLK_BOUNDARY)) l1 = true;
  }
  // (the layer-1 blocks are listed from the back of the block list — the boundary blocks fill it from the front, a block is never both)
  if (l1 && !(atomicOr(&blk_class[i >> 6], MRS_BLK_LAYER1) & MRS_BLK_LAYER1)) blk_list[(uint32_t)((n + 63) / 64) - 1u - atomicAdd(&fctl[CTL_NL1], 1u)] = (uint32_t)(i >> 6);
}

// start of a run of split ticks behind launch `tau`: every block counts as finished by that launch, nobody has arrived yet
__global__ void k_handoff_init(uint32_t* fctl, uint32_t* epoch, int n_blocks, uint32_t tau) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < n_blocks) epoch[b] = tau;
  if (b == 0) fctl[CTL_I_STARTED] = tau;
}

__global__ void k_export_header(uint32_t* map, const uint32_t* fctl, const uint32_t* ctl) {  // (a rank without UAVs)
  map[0] = fctl[CTL_EXPORTS];
  map[1] = ctl[6];
}

__global__ void k_export_translate(int n, long long n_max, int rank, long long map_stride, int block, uint32_t* nbr, const uint32_t* nbr_cnt,
                                   const uint32_t* maps, const PosRecord* rec_all, Pos4* x_recv, PartnerConst* x_const, uint32_t* fctl) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t cnt = nbr_cnt[i];
  for (uint32_t k = 0; k < cnt; k++) {
    const size_t    at = (size_t)k * (size_t)n + (size_t)i;
    const uint32_t  g  = nbr[at];
    const long long q  = (long long)g / n_max, j = (long long)g - q * n_max;
    if (q == (long long)rank) {
      nbr[at] = (uint32_t)j;
      continue;
    }
    const uint32_t e = maps[(size_t)q * (size_t)map_stride + 2 + (size_t)j];
    if (e == MRS_NO_SLOT || (long long)e + 1 >= (long long)block) {  // cannot happen (symmetry / capacity checked by the host): keep the entry harmless
      atomicAdd(&fctl[CTL_BADSLOT], 1u);
      nbr[at] = MRS_NBR_FOREIGN | (uint32_t)(q * block);  // the owner's header record: w = stall word, position (0,0,0) + zero constants
      continue;
    }
    const uint32_t slot = (uint32_t)(q * block + 1 + e);
    nbr[at] = MRS_NBR_FOREIGN | slot;
    const PosRecord r = rec_all[g];  // several lanes may write the same slot: same values
    const Pos4         pp = {r.x, r.y, r.z, 0.0};
    const PartnerConst cc = {r.mass, r.arm_length, r.prop_radius, 0.0};
    x_recv[slot]  = pp;
    x_const[slot] = cc;
  }
}

// handleCollisions of the tick after the most recent step, evaluated on its own from the lists (local partners: position records,
// foreign partners: gathered export buffer — both current): the settle step at the end of a run of sharded ticks
// own_from_records: the UAV's own position comes from the position records too (cd.p_in) and crash flags are left alone — the
// force a fused launch evaluated but did not latch (CollDev::write_force == 0), re-derived from the very positions it used
template <bool OWN_FROM_RECORDS>
__global__ void k_list_eval_cd(SwarmDev sw, CollDev cd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sw.n) return;
  const size_t      np = (size_t)sw.npad;
  const TypeParams& P  = sw.T[sw.F[i] >> FLAG_TYPE_SHIFT];
  PosRecord         me;
  if (OWN_FROM_RECORDS) {
    const Pos4 pp = cd.p_in[i];
    me.x = pp.x; me.y = pp.y; me.z = pp.z;
  } else {
    me.x = sw.S[(size_t)(F_X + 0) * np + i];
    me.y = sw.S[(size_t)(F_X + 1) * np + i];
    me.z = sw.S[(size_t)(F_X + 2) * np + i];
  }
  me.mass = P.mass; me.arm_length = P.arm_length; me.prop_radius = P.prop_radius;
  double f[3];
  bool   crashed;
  const uint32_t cnt = cd.nbr_cnt[i];
  double         ox, oy, oz, om, oa, op;
  mrs_partner_flat(cd, cnt ? cd.nbr[i] : (uint32_t)i, ox, oy, oz, om, oa, op);
  mrs_list_eval(cd, i, me.x, me.y, me.z, me.mass, me.arm_length, me.prop_radius, cnt, ox, oy, oz, om, oa, op, f, crashed);
  sw.S[(size_t)(F_FEXT + 0) * np + i] = f[0];
  sw.S[(size_t)(F_FEXT + 1) * np + i] = f[1];
  sw.S[(size_t)(F_FEXT + 2) * np + i] = f[2];
  if (crashed && !OWN_FROM_RECORDS) sw.F[i] |= FLAG_CRASHED;
}

// the stall words of all ranks (headers of the gathered export buffer) folded into this rank's control words: run at the end of a
// batch of ticks, whose last launch nobody has looked behind yet
// progress_tau != 0: also stands in for the fused launch of a rank that holds no UAVs (it reports progress and the warning word)
__global__ void k_fold_stall(const Pos4* x_recv, int world, int block, Pos4* x_send, uint32_t* fctl, volatile uint32_t* hostw, uint32_t progress_tau) {
  uint32_t* own   = (uint32_t*)x_send;  // the rank's own header words (MRS_HDR_*): what it knows, what its next collective carries
  uint32_t  stall = own[MRS_HDR_STALL], warn = own[MRS_HDR_WARN], herr = 0u;
  // (a communicator of ONE rank: the launches keep their words where a single GPU keeps them)
  if (fctl[CTL_STALL] != 0u && (stall == 0u || fctl[CTL_STALL] < stall)) stall = fctl[CTL_STALL];
  if (fctl[CTL_WARN] != 0u && (warn == 0u || fctl[CTL_WARN] < warn)) warn = fctl[CTL_WARN];
  for (int q = 0; q < world; q++) {
    const uint32_t* hq = (const uint32_t*)(x_recv + (size_t)q * (size_t)block);
    const uint32_t  h = hq[MRS_HDR_STALL], wq = hq[MRS_HDR_WARN];
    herr |= hq[MRS_HDR_ERROR];
    if (h != 0u && (stall == 0u || h < stall)) stall = h;
    if (wq != 0u && (warn == 0u || wq < warn)) warn = wq;
  }
  if ((herr & 3u) != 0u) fctl[CTL_ERROR] |= (herr & 3u) << 8;  // some rank's kernels reported an error: every rank's call fails
  own[MRS_HDR_STALL] = stall;
  own[MRS_HDR_WARN]  = warn;
  __hip_atomic_store(&hostw[CTL_STALL], stall, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&hostw[CTL_WARN], warn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&hostw[CTL_STALL2], stall, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // (nothing else runs: both chains' mirrors agree)
  __hip_atomic_store(&hostw[CTL_WARN2], warn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (progress_tau != 0u && (stall == 0u || progress_tau <= stall))
    __hip_atomic_store(&hostw[CTL_PROGRESS], progress_tau, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

// sizes of the export-set exchange for `world` ranks and `cap` export slots per rank; buffers zeroed (headers!)
// zero == 0: the caller's next launch is the search's reset kernel, which zeroes the allocation itself (one launch less per search)
extern "C" hipError_t mrs_collide_export_prepare(SwarmDev sw, CollideWork** work, int world, long long cap, int zero, hipStream_t st) {
  if (!*work) *work = new CollideWork();
  CollideWork* w = *work;
  CK(ensure_fused(w, sw.n > 0 ? sw.n : 1, st));
  if ((long long)sw.n > w->exp_slot_cap) {
    CK(hipStreamSynchronize(st));
    (void)hipFree(w->exp_slot);
    CK(hipMalloc(&w->exp_slot, sizeof(uint32_t) * (size_t)(sw.n > 0 ? sw.n : 1)));
    w->exp_slot_cap = sw.n;
  }
  const long long n_blocks = ((long long)(sw.n > 0 ? sw.n : 1) + 63) / 64;
  if (n_blocks > w->blk_cap) {
    CK(hipStreamSynchronize(st));
    (void)hipFree(w->blk_class); (void)hipFree(w->blk_list); (void)hipFree(w->epoch);
    CK(hipMalloc(&w->blk_class, sizeof(uint32_t) * (size_t)n_blocks));
    CK(hipMalloc(&w->blk_list, sizeof(uint32_t) * (size_t)n_blocks));
    CK(hipMalloc(&w->epoch, sizeof(uint32_t) * (size_t)n_blocks));
    CK(hipMemsetAsync(w->blk_class, 0, sizeof(uint32_t) * (size_t)n_blocks, st));
    CK(hipMemsetAsync(w->epoch, 0, sizeof(uint32_t) * (size_t)n_blocks, st));
    w->blk_cap = n_blocks;
  }
  if (cap > w->x_cap || world != w->x_world) {
    CK(hipStreamSynchronize(st));
    (void)hipFree(w->x_send);  // (one allocation: send block, gathered blocks, partner constants — zeroed by one launch per search)
    const size_t block = (size_t)cap + 1;
    char* base = nullptr;
    CK(hipMalloc(&base, sizeof(Pos4) * block * (size_t)(1 + 2 * world)));
    w->x_send  = (Pos4*)base;
    w->x_recv  = w->x_send + block;
    w->x_const = (PartnerConst*)(w->x_recv + block * (size_t)world);
    w->x_cap   = cap;
    w->x_world = world;
  }
  const size_t block = (size_t)w->x_cap + 1;
  static_assert(sizeof(Pos4) == sizeof(PartnerConst), "one stride for the three parts of the export allocation");
  if (zero) CK(hipMemsetAsync(w->x_send, 0, sizeof(Pos4) * block * (size_t)(1 + 2 * world), st));
  return hipSuccess;
}

// one wave that watches the 100 MHz wall clock for `microseconds` (mrs_debug_stream_delay: stands in for a collective's latency)
namespace {
__global__ void k_stream_delay(long long ticks) {
  const long long t0 = wall_clock64();
  unsigned        k  = 0;
  while (wall_clock64() - t0 < ticks && k < 400000000u) k++;
}
}  // namespace
namespace {
// The collective of the measurement stand-in as ONE kernel that lasts `ticks` of the 100 MHz clock (one wave watches it — a real
// collective keeps a few waves busy, not the chip): the rank's own block of `bytes` bytes is copied to the places
// of ranks rank-1, rank, rank+1 of `recv`; when the blocks are 48-byte records (`kind` 1), absent ranks read as NaN records and the
// two images are moved one slab width to either side.  Likewise the 64-byte entries of a halo exchange (kind 2; absent ranks: an
// empty header) and the search box at the tail of a slot map (kind 3, `aux` = its first 16-byte vector; absent ranks: NaN bounds).
__global__ void k_standin_gather(const uint4* send, uint4* recv, long long vec_per_rank, int rank, int world, int kind, long long aux, double width,
                                 long long ticks) {
  const long long t_start = wall_clock64();
  const long long total = vec_per_rank * (long long)world;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (long long)gridDim.x * blockDim.x) {
    const int       q = (int)(v / vec_per_rank);
    const long long j = v - (long long)q * vec_per_rank;
    const int       d = q - rank;
    if (d < -1 || d > 1) {
      if (kind == 1 || (kind == 3 && j >= aux && j < aux + 3)) recv[v] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
      if (kind == 2 && j < 4) recv[v] = make_uint4(0u, 0u, 0u, 0u);
      continue;
    }
    uint4 x = send[j];
    // a 48-byte record = three 16-byte vectors, a halo entry four, the first holds x and y; a box = xmin ymin | zmin xmax | ymax zmax
    const bool lo = d != 0 && ((kind == 1 && j % 3 == 0) || (kind == 2 && j % 4 == 0) || (kind == 3 && j == aux));
    const bool hi = d != 0 && kind == 3 && j == aux + 1;
    if (lo) {
      double px = __builtin_bit_cast(double, make_uint2(x.x, x.y));
      px += (double)d * width;
      const uint2 b = __builtin_bit_cast(uint2, px);
      x.x = b.x; x.y = b.y;
    }
    if (hi) {
      double px = __builtin_bit_cast(double, make_uint2(x.z, x.w));
      px += (double)d * width;
      const uint2 b = __builtin_bit_cast(uint2, px);
      x.z = b.x; x.w = b.y;
    }
    recv[v] = x;
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {  // ONE wave keeps the kernel alive until the collective's latency is over
    unsigned k = 0;
    while (wall_clock64() - t_start < ticks && k < 400000000u) k++;
  }
}
}  // namespace
extern "C" hipError_t mrs_launch_standin_gather(const void* send, void* recv, size_t bytes, int rank, int world, double latency_us, int kind, long long aux,
                                                double width, hipStream_t st) {
  if (bytes % 16 != 0) return hipErrorInvalidValue;
  const long long vec = (long long)(bytes / 16);
  long long       blocks = (vec * world + 255) / 256;
  if (blocks > 512) blocks = 512;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_standin_gather, dim3((unsigned)blocks), dim3(256), 0, st, (const uint4*)send, (uint4*)recv, vec, rank, world, kind, aux, width,
                     (long long)(latency_us * 100.0));
  return hipGetLastError();
}
// ---- peer-window all-gather: the exchange of a sharded swarm WITHOUT a collective library in the tick ----
// Every rank owns a WINDOW in its own device memory (fine-grained, mapped into every peer: hipIpcOpenMemHandle across processes,
// plain pointers inside one): [world flag lines of 64 B | pad to 4096 | 2 parities x world slots of slot_bytes].  One kernel per
// rank and collective, no host in between.  A group of `bpp` blocks serves ONE peer q (the group of the own rank copies the own
// block into the receive buffer and is done):
//   1. push   — the group's shares of the rank's block go straight into slot [seq & 1][rank] of q's window (system-scope stores:
//               one hop over xGMI);
//   2. signal — every block fences its stores (system scope); the last block of the group (a ticket when bpp > 1) writes seq into
//               flag[rank] of q's window;
//   3. wait   — one lane polls flag[q] of the OWN window until q has signalled seq (bounded: 10 s, then the error word is set and
//               the kernel ends — the call reports it), acquire;
//   4. pull   — q's slot is copied from the own window into the receive buffer (system-scope loads: another device wrote the lines).
// On exit the receive buffer holds what an all-gather would have put there, so the caller's kernels do not know the difference;
// no block waits for another block of its own launch except through the ticket, which needs no residency (it is taken after the work).
// Two parities suffice: a peer can only be one collective ahead (it cannot finish seq+1 without this rank's flag for seq+1, which
// is written by this rank's kernel seq+1, i.e. after its kernel seq has ended), so what it writes while this rank still pulls seq
// goes to the other parity.
// Cost: one one-way latency + the copy, where a ring all-gather pays 2 (world - 1) hops behind a kernel launch of its own.
namespace {
typedef __attribute__((address_space(1))) unsigned long long peer_u64;
typedef __attribute__((address_space(1))) unsigned           peer_u32;
template <class U> struct PeerWord;
template <> struct PeerWord<unsigned long long> { typedef peer_u64 G; };
template <> struct PeerWord<unsigned>           { typedef peer_u32 G; };
#define MRS_PEER_THREADS 512

template <class U>
__global__ __launch_bounds__(MRS_PEER_THREADS) void k_peer_allgather(MrsPeerWindows pw, const U* __restrict__ send, U* __restrict__ recv, long long units,
                                                                     int rank, int bpp, unsigned seq, unsigned long long slot_bytes,
                                                                     unsigned* tickets, unsigned ticket_target, unsigned* err_host, int world) {
  typedef typename PeerWord<U>::G G;
  const int       q = (int)blockIdx.x / bpp, j = (int)blockIdx.x - q * bpp;
  const long long share = (units + bpp - 1) / bpp, lo = (long long)j * share, hi = lo + share < units ? lo + share : units;
  if (q == rank) {
    for (long long u = lo + threadIdx.x; u < hi; u += MRS_PEER_THREADS) recv[(long long)rank * units + u] = send[u];
    return;
  }
  // 1. push
  G* there = (G*)((char*)pw.win[q] + 4096ull + ((unsigned long long)(seq & 1u) * (unsigned)world + (unsigned)rank) * slot_bytes);
  for (long long u = lo + threadIdx.x; u < hi; u += MRS_PEER_THREADS) __hip_atomic_store(there + u, send[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // 2. signal
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    bool last = true;
    if (bpp > 1) last = __hip_atomic_fetch_add((peer_u32*)(tickets + q), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == ticket_target;
    if (last) __hip_atomic_store((peer_u32*)((char*)pw.win[q] + 64 * rank), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // 3. wait
    const peer_u32* flag = (const peer_u32*)((const char*)pw.win[rank] + 64 * q);
    const long long t0   = wall_clock64();
    // (an exchange of this rank has given up before: the results are void already, the call will say so — what is still queued
    //  must not wait its 10 s again, launch after launch)
    //  (the mark is kept in device memory too — tickets[MRS_MAX_PEERS] —: the pinned host word is a PCIe round trip away)
    const bool dead = __hip_atomic_load((peer_u32*)(tickets + MRS_MAX_PEERS), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
    while (!dead && (int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      if (wall_clock64() - t0 > MRS_WAIT_TICKS) {
        // (pinned host words, plain stores — no PCIe atomic: [0] = set, [1] = the collective, [2] = the peer, [3] = what its flag said)
        __hip_atomic_store((peer_u32*)err_host + 1, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store((peer_u32*)err_host + 2, (unsigned)q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store((peer_u32*)err_host + 3, __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store((peer_u32*)err_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store((peer_u32*)(tickets + MRS_MAX_PEERS), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  // 4. pull
  const G* here = (const G*)((const char*)pw.win[rank] + 4096ull + ((unsigned long long)(seq & 1u) * (unsigned)world + (unsigned)q) * slot_bytes);
  for (long long u = lo + threadIdx.x; u < hi; u += MRS_PEER_THREADS)
    recv[(long long)q * units + u] = __hip_atomic_load(here + u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace
// `ticket_total`: tickets every peer's word has seen from all earlier launches of this rank (the caller adds the returned `bpp`)
extern "C" hipError_t mrs_launch_peer_allgather(const MrsPeerWindows* pw, const void* send, void* recv, size_t bytes, int rank, int world, unsigned seq,
                                                size_t slot_bytes, unsigned* tickets, unsigned ticket_total, unsigned* err_host, unsigned* bpp_out,
                                                hipStream_t st) {
  if (bytes == 0 || bytes % 4 != 0 || bytes > slot_bytes || world < 1 || world > MRS_MAX_PEERS) return hipErrorInvalidValue;
  const bool      wide  = bytes % 8 == 0;
  const long long units = (long long)(bytes / (wide ? 8 : 4));
  long long       bpp   = ((long long)bytes + 131071) / 131072;  // one block per peer up to 128 KiB (the per-tick export blocks), then one per 128 KiB
  if (bpp > 16) bpp = 16;
  *bpp_out = bpp > 1 ? (unsigned)bpp : 0u;  // (a group of one block takes no ticket)
  const dim3 grid((unsigned)(bpp * world)), block(MRS_PEER_THREADS);
  if (wide)
    hipLaunchKernelGGL(k_peer_allgather<unsigned long long>, grid, block, 0, st, *pw, (const unsigned long long*)send, (unsigned long long*)recv, units, rank,
                       (int)bpp, seq, (unsigned long long)slot_bytes, tickets, ticket_total + (unsigned)bpp, err_host, world);
  else
    hipLaunchKernelGGL(k_peer_allgather<unsigned>, grid, block, 0, st, *pw, (const unsigned*)send, (unsigned*)recv, units, rank, (int)bpp, seq,
                       (unsigned long long)slot_bytes, tickets, ticket_total + (unsigned)bpp, err_host, world);
  return hipGetLastError();
}
extern "C" hipError_t mrs_launch_stream_delay(hipStream_t st, double microseconds) {
  hipLaunchKernelGGL(k_stream_delay, dim3(1), dim3(64), 0, st, (long long)(microseconds * 100.0));
  return hipGetLastError();
}

namespace {
__global__ void k_heads_to_host(const uint32_t* maps, long long stride, int world, const uint32_t* fctl, volatile uint32_t* host, const HaloEntry* halo,
                                unsigned hcap) {
  const int q = threadIdx.x;
  {  // the halo headers of a search that ran on a halo exchange: the largest number of entries any rank wanted to send, all flags
    unsigned long long cnt = 0ull, fl = 0ull;
    if (halo && q < world) {
      const HaloEntry h = halo[(size_t)q * (size_t)(1u + hcap)];
      cnt = h.j > 0xFFFFFFFFull ? 0xFFFFFFFFull : h.j;
      fl  = h.pad;
    }
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long c2 = __shfl_xor(cnt, o), f2 = __shfl_xor(fl, o);
      cnt = c2 > cnt ? c2 : cnt;
      fl |= f2;
    }
    if (q == 0) {
      __hip_atomic_store(&host[2 * world + 2], (uint32_t)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&host[2 * world + 3], (uint32_t)fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (q < world) {
    __hip_atomic_store(&host[2 * q], maps[(size_t)q * (size_t)stride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&host[2 * q + 1], maps[(size_t)q * (size_t)stride + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (q == 0) {
    __hip_atomic_store(&host[2 * world], fctl[CTL_NBND], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&host[2 * world + 1], fctl[CTL_NL1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
}  // namespace
// what the host needs of a search — every rank's export count and overflow counter, this rank's boundary-block count — in pinned host
// memory after ONE small launch (two device-to-host copies cost a search 30 us); valid after the stream has been synchronised
// halo != 0: the search ran on a halo exchange; words [2 world + 2] = most entries wanted by a rank, [2 world + 3] = the flags of all ranks
extern "C" hipError_t mrs_collide_heads_to_host(CollideWork* w, const uint32_t* maps, long long stride, int world, int halo, const uint32_t** out, hipStream_t st) {
  if (!w || world > 64) return hipErrorInvalidValue;
  if (!w->host_heads) CK(hipHostMalloc(&w->host_heads, sizeof(uint32_t) * 160, hipHostMallocMapped | hipHostMallocCoherent));
  hipLaunchKernelGGL(k_heads_to_host, dim3(1), dim3(64), 0, st, maps, stride, world, w->fctl, w->host_heads, halo ? w->h_recv : nullptr, (unsigned)w->h_cap);
  *out = w->host_heads;
  return hipGetLastError();
}

extern "C" const uint32_t* mrs_collide_host_heads(const CollideWork* w) { return w ? w->host_heads : nullptr; }
// a search has replaced the lists: launch indices restart at 1, the pinned mirrors of the control words start from nothing
extern "C" void mrs_collide_host_words_reset(CollideWork* w) {
  if (!w || !w->hostw) return;
  w->hostw[CTL_STALL] = w->hostw[CTL_PROGRESS] = w->hostw[CTL_WARN] = w->hostw[CTL_STALL2] = w->hostw[CTL_WARN2] = 0u;
}
extern "C" long long mrs_collide_export_capacity(const CollideWork* w) { return w ? w->x_cap : 0; }
extern "C" const uint32_t* mrs_collide_ctl_words(const CollideWork* w) { return w ? w->fctl : nullptr; }
extern "C" void*     mrs_collide_export_send(const CollideWork* w) { return w ? (void*)w->x_send : nullptr; }
extern "C" void*     mrs_collide_export_recv(const CollideWork* w) { return w ? (void*)w->x_recv : nullptr; }

// after a search over gathered records: mark the export set, write this rank's slot map (2 + n_max words) for the all-gather
// pred_hdt >= 0: also the displacement bound over that time on the state the search found (k_export_mark)
extern "C" hipError_t mrs_collide_export_mark(SwarmDev sw, CollideWork* w, long long n_max, long long map_words, int rank, uint32_t* map_send, double pred_hdt,
                                              double rebounce, hipStream_t st) {
  // (the host mirrors of the control words are reset by the host once it has read what the old segment left in them: mrs_collide_host_words_reset)
  const long long n_xvec = (long long)(sizeof(Pos4) * ((size_t)w->x_cap + 1) * (size_t)(1 + 2 * w->x_world) / sizeof(uint4));
  long long       grid   = (map_words + 255) / 256;
  if (grid < 64) grid = 64;
  hipLaunchKernelGGL(k_search_reset, dim3((unsigned)grid), dim3(256), 0, st, w->fctl, map_send, map_words, w->blk_class, (sw.n + 63) / 64, (uint4*)w->x_send, n_xvec);
  if (sw.n > 0) {
    const int n_blocks = (sw.n + 63) / 64;
    (void)n_blocks;
    hipLaunchKernelGGL(k_export_mark, dim3((sw.n + 255) / 256), dim3(256), 0, st, sw, w->P[w->pcur], n_max, rank, w->nbr, w->nbr_cnt, w->exp_slot, map_send,
                       w->fctl, w->blk_class, pred_hdt, 0.5 * SKIN2 * (1.0 - 1e-9), rebounce);
    hipLaunchKernelGGL(k_class_list, dim3((sw.n + 255) / 256), dim3(256), 0, st, sw.n, n_max, rank, w->nbr, w->nbr_cnt, w->blk_class, w->blk_list, w->fctl,
                       map_send, w->ctl ? w->ctl : w->fctl);
  } else {
    hipLaunchKernelGGL(k_export_header, dim3(1), dim3(1), 0, st, map_send, w->fctl, w->ctl ? w->ctl : w->fctl);  // (never searched: word 6 of fctl is 0)
  }
  return hipGetLastError();
}

// after the all-gather of the slot maps (and with buffers of sufficient capacity): rewrite the lists, seed the gathered export buffer
extern "C" hipError_t mrs_collide_export_translate(SwarmDev sw, CollideWork* w, long long n_max, long long map_stride, int rank, const uint32_t* maps,
                                                   const PosRecord* rec_all, hipStream_t st) {
  if (sw.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_export_translate, dim3((sw.n + 255) / 256), dim3(256), 0, st, sw.n, n_max, rank, map_stride, (int)(w->x_cap + 1), w->nbr, w->nbr_cnt,
                     maps, rec_all, w->x_recv, w->x_const, w->fctl);
  return hipGetLastError();
}

// CollDev of a sharded fused launch (export-set exchange); rec_own = this rank's records as of the search (gathered buffer + offset)
extern "C" hipError_t mrs_collide_export_dev(const SwarmDev* sw, CollideWork* w, long long my_offset, unsigned tau, int eval, int crash, double rebounce,
                                             CollDev* cd) {
  if (!w || !w->fctl || !w->P[0] || !w->x_send || !w->g_rec_build) return hipErrorInvalidValue;
  memset(cd, 0, sizeof *cd);
  cd->nbr      = w->nbr;
  cd->nbr_cnt  = w->nbr_cnt;
  cd->rec      = w->g_rec_build + my_offset;
  cd->p_in     = w->P[w->pcur];
  cd->p_out    = w->P[(w->pcur + 1) % 3];
  cd->ctl      = w->fctl;
  cd->hostw    = w->hostw;
  cd->g_pos    = w->x_recv;
  cd->g_const  = w->x_const;
  cd->send     = w->x_send;
  cd->exp_slot = w->exp_slot;
  cd->rebounce = rebounce;
  cd->lim2     = (0.5 * SKIN2) * (0.5 * SKIN2) * (1.0 - 1e-9);
  cd->lim2_warn = cd->lim2 * (WARN_FRACTION * WARN_FRACTION);  // the warning travels in the collective's headers (tick_sharded.hip: export_ticks)
  cd->tau      = tau;
  cd->n        = sw->n;
  cd->eval     = eval;
  cd->crash    = crash;
  cd->world    = w->x_world;
  cd->block    = (int)(w->x_cap + 1);
  cd->part      = MRS_PART_FULL;
  cd->blk_class = w->blk_class;
  cd->blk_list  = w->blk_list;
  cd->epoch     = w->epoch;
  cd->pred_lim  = 0.5 * SKIN2 * (1.0 - 1e-9);
  cd->pred_hdt  = -1.0;  // (nothing announced unless the caller says so: mrs_collide_export_part)
  return hipSuccess;
}

// the part of a split tick this launch is (MRS_PART_*) and the step of the displacement bound
// announce: the protocol runs split ticks (on any rank), so "may leave its skin within MRS_PRED_HORIZON steps" has to be reported ahead
extern "C" void mrs_collide_export_part(CollDev* cd, int part, unsigned n_bnd, double dt, int announce) {
  cd->part          = part;
  cd->n_bnd         = n_bnd;
  cd->pred_hdt      = announce ? (double)MRS_PRED_HORIZON * dt : -1.0;
}

// a run of split ticks starts behind launch `tau` (everything before it has completed in stream order)
extern "C" hipError_t mrs_collide_handoff_init(CollideWork* w, int n, unsigned tau, hipStream_t st) {
  if (!w || !w->epoch) return hipErrorInvalidValue;
  const int n_blocks = (n + 63) / 64;
  hipLaunchKernelGGL(k_handoff_init, dim3((n_blocks + 255) / 256), dim3(256), 0, st, w->fctl, w->epoch, n_blocks, tau);
  return hipGetLastError();
}

extern "C" hipError_t mrs_collide_export_eval(SwarmDev sw, CollDev cd, hipStream_t st) {
  if (sw.n <= 0) return hipSuccess;
  cd.crash = mode_word(sw, cd.crash);
  hipLaunchKernelGGL(k_list_eval_cd<false>, dim3((sw.n + 255) / 256), dim3(256), 0, st, sw, cd);
  return hipGetLastError();
}

// the force of the collision tick a fused launch evaluated without latching it: same lists, same position records (index `pin`)
extern "C" hipError_t mrs_collide_latch_force(SwarmDev sw, CollideWork* w, int pin, int crash, double rebounce, hipStream_t st) {
  if (sw.n <= 0 || !w || !w->P[0]) return hipSuccess;
  CollDev cd;
  memset(&cd, 0, sizeof cd);
  cd.nbr = w->nbr; cd.nbr_cnt = w->nbr_cnt; cd.rec = w->rec_build; cd.p_in = w->P[pin % 3];
  cd.rebounce = rebounce; cd.n = sw.n; cd.eval = 1; cd.crash = mode_word(sw, crash); cd.world = 1;
  hipLaunchKernelGGL(k_list_eval_cd<true>, dim3((sw.n + 255) / 256), dim3(256), 0, st, sw, cd);
  return hipGetLastError();
}
extern "C" int mrs_collide_fused_pin(const CollideWork* w) { return w ? w->pcur : 0; }

extern "C" hipError_t mrs_collide_export_fold_stall(CollideWork* w, unsigned progress_tau, hipStream_t st) {
  hipLaunchKernelGGL(k_fold_stall, dim3(1), dim3(1), 0, st, w->x_recv, w->x_world, (int)(w->x_cap + 1), w->x_send, w->fctl, w->hostw, progress_tau);
  return hipGetLastError();
}

// control words of the fused machinery (synchronises the stream)
extern "C" hipError_t mrs_collide_fused_words(const CollideWork* w, hipStream_t st, unsigned* out8) {
  for (int k = 0; k < CTL_WORDS; k++) out8[k] = 0;
  if (!w || !w->fctl) return hipSuccess;
  CK(hipMemcpyAsync(out8, w->fctl, CTL_WORDS * sizeof(unsigned), hipMemcpyDeviceToHost, st));
  return hipStreamSynchronize(st);
}

extern "C" void mrs_collide_invalidate_gathered(CollideWork* w) {
  if (w) w->g_lists_live = false;
}
