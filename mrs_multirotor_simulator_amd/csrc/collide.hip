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
// Pipeline per tick (all on the swarm's stream, HBM-bound integer/index work):
//   pack      : SoA state + type table -> 48-B PosRecord per UAV (the multi-GPU all-gather payload; fused with
//               hash_count on a single GPU)
//   hash_count: cell = floor(pos / 1.75 m) (> sqrt(3), so partners sit in the 27 adjacent cells);
//               bucket = hash(cell) & (T-1); rank = atomicAdd(count[bucket])
//   alloc     : every 1024-bucket block scans its counts and reserves its slice of `sorted` with one atomicAdd
//               (buckets need disjoint slices, not ordered ones); writes {start,count} per bucket, re-zeroes count[]
//   scatter   : sorted[start[bucket] + rank] = j
//   query     : one lane per local UAV fetches the 27 bucket descriptors, walks the non-empty ones with an exact cell
//               match (dedupes buckets shared by several cells) and the literal predicate; partners are consumed in
//               ascending index, which makes the result independent of the atomic arrival order
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

// second, independent hash of the exact cell: stored next to every index in `sorted`, it lets the query discard the members of
// OTHER cells that share a bucket (about 5 per UAV at load factor 0.19) without fetching their 48-B records
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

__global__ void k_hash_count(const PosRecord* rec, long long n_total, uint32_t mask, uint32_t* key, uint32_t* rank, uint32_t* tag,
                             uint32_t* count, uint32_t* cursor) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j == 0) *cursor = 0;  // the allocation cursor of this tick (k_alloc_buckets runs after this kernel)
  if (j >= n_total) return;
  const Cell c = cell_of(rec[j].x, rec[j].y, rec[j].z);
  if (!c.ok) {
    key[j] = 0xFFFFFFFFu;  // never inserted
    return;
  }
  const uint32_t b = bucket_of(c.x, c.y, c.z, mask);
  key[j]  = b;
  tag[j]  = cell_tag(c.x, c.y, c.z);
  rank[j] = atomicAdd(&count[b], 1u);
}

// single-GPU tick: pack and hash in one pass over the state (the records are still written: the query reads them)
__global__ void k_pack_hash_count(SwarmDev sw, PosRecord* rec, uint32_t mask, uint32_t* key, uint32_t* rank, uint32_t* tag,
                                  uint32_t* count, uint32_t* cursor) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *cursor = 0;
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
  const Cell c = cell_of(r.x, r.y, r.z);
  if (!c.ok) {
    key[i] = 0xFFFFFFFFu;
    return;
  }
  const uint32_t b = bucket_of(c.x, c.y, c.z, mask);
  key[i]  = b;
  tag[i]  = cell_tag(c.x, c.y, c.z);
  rank[i] = atomicAdd(&count[b], 1u);
}

// ---- bucket storage allocation: one kernel instead of a full prefix sum ----
// Buckets need disjoint slices of `sorted`, not slices in bucket order: each 1024-bucket block scans its own counts
// in registers/LDS and reserves its total with ONE atomicAdd on a global cursor.  Writes start<<6|count per bucket and
// re-zeroes count[] for the next tick (this kernel is its last reader).
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t* total) {
  __shared__ uint32_t wsum[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t  inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(inc, off, 64);
    if (lane >= off) inc += o;
  }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int q = 0; q < wv; q++) base += wsum[q];
  *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  return base + inc - v;
}

__global__ void __launch_bounds__(256) k_alloc_buckets(uint32_t* count, uint2* cell, uint32_t* cursor) {
  __shared__ uint32_t block_base;
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  const uint4  c = reinterpret_cast<const uint4*>(count)[q];
  uint32_t     total;
  const uint32_t ex = block_excl_scan_256(c.x + c.y + c.z + c.w, &total);
  if (threadIdx.x == 0) block_base = total ? atomicAdd(cursor, total) : 0u;
  __syncthreads();
  const uint32_t s0 = block_base + ex;
  uint2* o = cell + q * 4;
#define MRS_DESC(start, cnt) make_uint2(((start) << 6) | ((cnt) < 63u ? (cnt) : 63u), 0u)
  o[0] = MRS_DESC(s0, c.x);
  o[1] = MRS_DESC(s0 + c.x, c.y);
  o[2] = MRS_DESC(s0 + c.x + c.y, c.z);
  o[3] = MRS_DESC(s0 + c.x + c.y + c.z, c.w);
#undef MRS_DESC
  reinterpret_cast<uint4*>(count)[q] = make_uint4(0, 0, 0, 0);
}

// Descriptor of a bucket: x = start << 6 | min(count, 63), y = cell tag of its rank-0 member.  A bucket with ONE member (the
// common non-empty case) can then be rejected by the query from the descriptor alone when that member belongs to another cell.
__global__ void k_scatter(long long n_total, const uint32_t* key, const uint32_t* rank, const uint32_t* tag, uint2* cell, uint2* sorted) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_total) return;
  const uint32_t b = key[j];
  if (b == 0xFFFFFFFFu) return;
  const uint32_t r = rank[j];
  sorted[(cell[b].x >> 6) + r] = make_uint2((uint32_t)j, tag[j]);  // (scattering the 48-B records too was measured 6x slower)
  if (r == 0) cell[b].y = tag[j];
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

// reference path for one lane: repeated sweeps over the 27 buckets, each returning the smallest qualifying partner
// index above the previous one (ascending-index accumulation without per-lane arrays).  Correct for any bucket
// occupancy; used when the wave-cooperative path below overflows its LDS lists.
__device__ void query_lane_sweeps(const PosRecord& me, const Cell& c, long long gi, const PosRecord* rec, long long n_total, uint32_t mask,
                                  const uint2* cell, const uint2* sorted, uint32_t n_entries, int crash, double rebounce, double& fx, double& fy,
                                  double& fz, bool& crashed) {
  long long prev = -1;
  for (;;) {
    long long best = n_total;
    for (int q = 0; q < 27; q++) {
      const int   cx = c.x + q / 9 - 1, cy = c.y + (q / 3) % 3 - 1, cz = c.z + q % 3 - 1;
      const uint2    info = cell[bucket_of(cx, cy, cz, mask)];
      const uint32_t s0 = info.x >> 6;
      uint32_t       cn = info.x & 63u;
      if (cn == 63u) {  // saturated count field: walk until the members stop hashing to this bucket
        cn = 0;
        while (s0 + cn < n_entries) {  // entries [0, n_entries) were written this tick; slices of different buckets are disjoint
          const PosRecord t = rec[sorted[s0 + cn].x];
          const Cell      tc = cell_of(t.x, t.y, t.z);
          if (!tc.ok || bucket_of(tc.x, tc.y, tc.z, mask) != bucket_of(cx, cy, cz, mask)) break;
          cn++;
        }
      }
      for (uint32_t e = 0; e < cn; e++) {
        const long long j = sorted[s0 + e].x;
        if (j <= prev || j >= best || j == gi) continue;
        const PosRecord o  = rec[j];
        const Cell      oc = cell_of(o.x, o.y, o.z);
        if (oc.x != cx || oc.y != cy || oc.z != cz) continue;  // another cell sharing the bucket
        if (qualifies(me, o, crash)) best = j;
      }
    }
    if (best >= n_total) break;
    apply_partner(me, rec[best], crash, rebounce, fx, fy, fz, crashed);
    prev = best;
  }
}

// Wave-cooperative query, one single-wave workgroup per 64 local UAVs.
//   A  every lane fetches its 27 bucket descriptors and the first entry of each non-empty bucket (independent loads,
//      one memory round trip), and counts its candidates (~2 on a 64 m^3/UAV swarm: nearly all buckets are empty)
//   B  the (owner lane, candidate) pairs of the whole wave are compacted into an LDS list (wave prefix sum)
//   C  the list is processed 64 pairs at a time with uniform control flow — this is what removes the 27-way divergent
//      walk in which some lane always had a non-empty bucket and every iteration paid a full memory latency;
//      qualifying partners go to a small per-owner hit list (LDS atomics)
//   D  owners order their (rare) hits by index and accumulate; overflow of either list falls back to query_lane_sweeps
constexpr int PAIR_CAP = 1024;
constexpr int HIT_CAP  = 6;

__global__ void __launch_bounds__(64) k_query(SwarmDev sw, const PosRecord* rec, long long n_total, long long my_offset, uint32_t mask,
                                              const uint2* cell, const uint2* sorted, const uint32_t* n_entries_p, int crash,
                                              double rebounce) {
  __shared__ PosRecord me_s[64];
  __shared__ int4      me_cell[64];
  __shared__ uint2     pair_e[PAIR_CAP];  // x: position in `sorted`, y: owner lane | probed cell q << 8
  __shared__ uint32_t  hit_j[64][HIT_CAP];
  __shared__ uint32_t  hit_n[64];
  __shared__ uint32_t  wave_total, overflow;

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
  if (lane == 0) overflow = 0;

  // A: descriptors and candidate count.  Unconditional loads from always-valid addresses: a load under a divergent
  // branch is waited for at the join, which would serialise 27 memory round trips.  A single-member bucket whose member
  // carries another cell's tag is dropped right here; members of multi-member buckets are tag-checked in phase C.
  uint2    info[27];
  uint32_t tc = 0;
  bool     dense = false;  // some probed bucket has a saturated count field
#pragma unroll
  for (int q = 0; q < 27; q++) info[q] = cell[bucket_of(c.x + q / 9 - 1, c.y + (q / 3) % 3 - 1, c.z + q % 3 - 1, mask)];
#pragma unroll
  for (int q = 0; q < 27; q++) {
    uint32_t cn = c.ok ? (info[q].x & 63u) : 0u;
    if (cn == 1u && info[q].y != cell_tag(c.x + q / 9 - 1, c.y + (q / 3) % 3 - 1, c.z + q % 3 - 1)) cn = 0u;
    dense |= (cn == 63u);
    info[q].y = cn;           // .y now holds the number of candidates taken from this bucket
    info[q].x = info[q].x >> 6;
    tc += cn;
  }
#if defined(MRS_QUERY_STOP) && MRS_QUERY_STOP == 1
  if (tc == 0xFFFFFFFFu) sw.F[i] = tc;
  return;
#endif
  // B: wave prefix sum -> slots in the pair list; a pair is (owner lane, probed cell q, position in `sorted`)
  uint32_t inc = tc;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = __shfl_up(inc, off, 64);
    if (lane >= off) inc += o;
  }
  if (lane == 63) wave_total = dense ? 0xFFFFFFFFu : inc;
  if (dense) overflow = 2;
  __syncthreads();
  const uint32_t total = overflow == 2 ? 0xFFFFFFFFu : wave_total;
  if (total > PAIR_CAP) {  // wave-uniform: dense neighbourhood, take the reference path
    double fx = 0, fy = 0, fz = 0;
    bool   crashed = false;
    if (active && c.ok) query_lane_sweeps(me, c, gi, rec, n_total, mask, cell, sorted, *n_entries_p, crash, rebounce, fx, fy, fz, crashed);
    if (active) {
      sw.S[(size_t)(F_FEXT + 0) * sw.npad + i] = fx;
      sw.S[(size_t)(F_FEXT + 1) * sw.npad + i] = fy;
      sw.S[(size_t)(F_FEXT + 2) * sw.npad + i] = fz;
      if (crashed) sw.F[i] |= FLAG_CRASHED;
    }
    return;
  }
#if defined(MRS_QUERY_STOP) && MRS_QUERY_STOP == 4
  if (total == 0xFFFFFFF0u) sw.F[i] = inc;
  return;
#endif
  uint32_t slot = inc - tc;
  bool     extras = false;
#pragma unroll
  for (int q = 0; q < 27; q++) {  // branch-light: the first member of every accepted bucket
    const uint32_t cn = info[q].y;
    if (cn) pair_e[slot] = make_uint2(info[q].x, (uint32_t)lane | ((uint32_t)q << 8));
    slot += cn ? 1u : 0u;
    extras |= cn > 1u;
  }
#if defined(MRS_QUERY_STOP) && MRS_QUERY_STOP == 5
  if (slot == 0xFFFFFFF0u) sw.F[i] = pair_e[lane].x;
  return;
#endif
  if (extras) {  // rare: further members of multi-member buckets
#pragma unroll
    for (int q = 0; q < 27; q++) {
      const uint32_t cn = info[q].y;
      for (uint32_t e = 1; e < cn; e++) pair_e[slot++] = make_uint2(info[q].x + e, (uint32_t)lane | ((uint32_t)q << 8));
    }
  }
  __syncthreads();
#if defined(MRS_QUERY_STOP) && MRS_QUERY_STOP == 2
  if (pair_e[lane].x == 0xFFFFFFFFu) sw.F[i] = 1;
  return;
#endif
  // C: uniform sweep over the pairs, four independent pairs per lane and iteration so that their loads overlap
  constexpr int U = 4;
  for (uint32_t base = 0; base < total; base += 64 * U) {
    uint2 pe[U], ent[U];
    bool  live[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t p = base + u * 64 + lane;
      pe[u] = pair_e[p < total ? p : 0u];
    }
#pragma unroll
    for (int u = 0; u < U; u++) ent[u] = sorted[pe[u].x];
    // tag filter: only members of exactly the probed cell survive (false positives of the 32-bit tag are caught by the exact
    // cell comparison below)
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t p = base + u * 64 + lane;
      live[u] = false;
      if (p < total) {
        const int  ow = (int)(pe[u].y & 0xFFu), q = (int)(pe[u].y >> 8);
        const int4 mc = me_cell[ow];
        live[u] = ent[u].y == cell_tag(mc.x + q / 9 - 1, mc.y + (q / 3) % 3 - 1, mc.z + q % 3 - 1) &&
                  (long long)ent[u].x != my_offset + blockIdx.x * 64 + ow;  // idx == i, src/multirotor_simulator.cpp:335
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (!live[u]) continue;
      const int       ow = (int)(pe[u].y & 0xFFu), q = (int)(pe[u].y >> 8);
      const int4      mc = me_cell[ow];
      const PosRecord o  = rec[ent[u].x];
      const Cell      oc = cell_of(o.x, o.y, o.z);
      if (oc.x != mc.x + q / 9 - 1 || oc.y != mc.y + (q / 3) % 3 - 1 || oc.z != mc.z + q % 3 - 1) continue;  // tag collision
      const PosRecord m = me_s[ow];
      if (!qualifies(m, o, crash)) continue;
      const uint32_t k = atomicAdd(&hit_n[ow], 1u);
      if (k < HIT_CAP)
        hit_j[ow][k] = ent[u].x;
      else
        overflow = 1;
    }
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
    if (overflow == 1 && hit_n[lane] > HIT_CAP) {
      query_lane_sweeps(me, c, gi, rec, n_total, mask, cell, sorted, *n_entries_p, crash, rebounce, fx, fy, fz, crashed);
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
  uint32_t *key = nullptr, *rank = nullptr, *tag = nullptr, *count = nullptr, *cursor = nullptr;
  uint2 *   cell = nullptr, *sorted = nullptr;
};

static void free_work(CollideWork* w) {
  (void)hipFree(w->key); (void)hipFree(w->rank); (void)hipFree(w->tag); (void)hipFree(w->sorted); (void)hipFree(w->count); (void)hipFree(w->cursor);
  (void)hipFree(w->cell);
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
    CK(hipMalloc(&w->key, sizeof(uint32_t) * (size_t)n_total));
    CK(hipMalloc(&w->rank, sizeof(uint32_t) * (size_t)n_total));
    CK(hipMalloc(&w->tag, sizeof(uint32_t) * (size_t)n_total));
    CK(hipMalloc(&w->sorted, sizeof(uint2) * (size_t)n_total));
    CK(hipMalloc(&w->count, sizeof(uint32_t) * (size_t)T));
    CK(hipMalloc(&w->cell, sizeof(uint2) * (size_t)T));
    CK(hipMalloc(&w->cursor, sizeof(uint32_t)));
    CK(hipMemsetAsync(w->count, 0, sizeof(uint32_t) * (size_t)T, st));  // k_alloc_buckets re-zeroes it after every use
    w->cap_n = n_total;
    w->cap_T = T;
  }
  T = w->cap_T;  // a larger table from an earlier call is still valid (count[] is all zero between ticks)
  const uint32_t mask = T - 1;
  const unsigned gN   = (unsigned)((n_total + 255) / 256);
  if (rec_is_local_scratch)
    hipLaunchKernelGGL(k_pack_hash_count, dim3(gN), dim3(256), 0, st, sw, const_cast<PosRecord*>(rec), mask, w->key, w->rank, w->tag, w->count,
                       w->cursor);
  else
    hipLaunchKernelGGL(k_hash_count, dim3(gN), dim3(256), 0, st, rec, n_total, mask, w->key, w->rank, w->tag, w->count, w->cursor);
  hipLaunchKernelGGL(k_alloc_buckets, dim3(T / 1024), dim3(256), 0, st, w->count, w->cell, w->cursor);
  hipLaunchKernelGGL(k_scatter, dim3(gN), dim3(256), 0, st, n_total, w->key, w->rank, w->tag, w->cell, w->sorted);
  hipLaunchKernelGGL(k_query, dim3((sw.n + 63) / 64), dim3(64), 0, st, sw, rec, n_total, my_offset, mask, w->cell, w->sorted, w->cursor,
                     crash, rebounce);
  return hipGetLastError();
}
