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
//   pack      : SoA state + type table -> 48-B PosRecord per UAV            (also the multi-GPU all-gather payload)
//   hash_count: cell = floor(pos / 1.75 m) (> sqrt(3), so partners sit in the 27 adjacent cells);
//               bucket = hash(cell) & (T-1); rank = atomicAdd(count[bucket])
//   scan      : exclusive prefix sum of count[T] (three small kernels)
//   scatter   : sorted[start[bucket] + rank] = j ; each bucket then ordered by index (deterministic)
//   query     : one lane per local UAV walks the 27 buckets, exact cell match (dedupes shared buckets), literal predicate
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "swarm_layout.h"

namespace {

constexpr double CELL_EDGE = 1.75;       // > sqrt(3.0) = 1.7320508
constexpr double POS_LIMIT = 1.0e9;      // |coordinate| beyond this (or non-finite) never collides here

struct Cell { int x, y, z; bool ok; };

__device__ __forceinline__ Cell cell_of(double x, double y, double z) {
  Cell c;
  c.ok = (fabs(x) < POS_LIMIT) && (fabs(y) < POS_LIMIT) && (fabs(z) < POS_LIMIT);  // false for NaN/inf
  c.x  = c.ok ? (int)floor(x / CELL_EDGE) : 0;
  c.y  = c.ok ? (int)floor(y / CELL_EDGE) : 0;
  c.z  = c.ok ? (int)floor(z / CELL_EDGE) : 0;
  return c;
}

__device__ __forceinline__ uint32_t bucket_of(int cx, int cy, int cz, uint32_t mask) {
  return (((uint32_t)cx * 73856093u) ^ ((uint32_t)cy * 19349663u) ^ ((uint32_t)cz * 83492791u)) & mask;
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

__global__ void k_hash_count(const PosRecord* rec, long long n_total, uint32_t mask, uint32_t* key, uint32_t* rank, uint32_t* count) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_total) return;
  const Cell c = cell_of(rec[j].x, rec[j].y, rec[j].z);
  if (!c.ok) {
    key[j] = 0xFFFFFFFFu;  // never inserted
    return;
  }
  const uint32_t b = bucket_of(c.x, c.y, c.z, mask);
  key[j]  = b;
  rank[j] = atomicAdd(&count[b], 1u);
}

// ---- exclusive scan over T = nblocks * 1024 counters ----
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
  __syncthreads();
  return base + inc - v;
}

__global__ void __launch_bounds__(256) k_scan_block_sums(const uint32_t* count, uint32_t* bsum) {
  const uint4 c = reinterpret_cast<const uint4*>(count)[(size_t)blockIdx.x * 256 + threadIdx.x];
  uint32_t    total;
  block_excl_scan_256(c.x + c.y + c.z + c.w, &total);
  if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256) k_scan_top(uint32_t* bsum, int nblocks) {  // one workgroup
  uint32_t carry = 0;
  for (int base = 0; base < nblocks; base += 256) {
    const int      idx = base + threadIdx.x;
    const uint32_t v   = idx < nblocks ? bsum[idx] : 0u;
    uint32_t       total;
    const uint32_t ex = block_excl_scan_256(v, &total);
    if (idx < nblocks) bsum[idx] = carry + ex;
    carry += total;
  }
}

__global__ void __launch_bounds__(256) k_scan_finish(const uint32_t* count, const uint32_t* bsum, uint32_t* start) {
  const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
  const uint4  c = reinterpret_cast<const uint4*>(count)[q];
  uint32_t     total;
  const uint32_t ex = block_excl_scan_256(c.x + c.y + c.z + c.w, &total) + bsum[blockIdx.x];
  uint4 o;
  o.x = ex;
  o.y = ex + c.x;
  o.z = o.y + c.y;
  o.w = o.z + c.z;
  reinterpret_cast<uint4*>(start)[q] = o;
}

__global__ void k_scatter(long long n_total, const uint32_t* key, const uint32_t* rank, const uint32_t* start, uint32_t* sorted) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_total) return;
  const uint32_t b = key[j];
  if (b == 0xFFFFFFFFu) return;
  sorted[start[b] + rank[j]] = (uint32_t)j;
}

// order every bucket by UAV index so that the force sums do not depend on atomic arrival order
__global__ void k_sort_buckets(uint32_t T, const uint32_t* count, const uint32_t* start, uint32_t* sorted) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= T) return;
  const uint32_t c = count[b];
  if (c < 2) return;
  uint32_t* p = sorted + start[b];
  for (uint32_t a = 1; a < c; a++) {
    const uint32_t val = p[a];
    uint32_t       q   = a;
    while (q > 0 && p[q - 1] > val) {
      p[q] = p[q - 1];
      q--;
    }
    p[q] = val;
  }
}

// one lane per local UAV.  Partners are consumed in ascending global index: each round finds the smallest hit
// index above the previous one (no per-lane arrays; >= 1 partner is rare, so normally a single sweep).
__global__ void __launch_bounds__(256) k_query(SwarmDev sw, const PosRecord* rec, long long n_total, long long my_offset, uint32_t mask,
                                               const uint32_t* count, const uint32_t* start, const uint32_t* sorted, int crash,
                                               double rebounce) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sw.n) return;
  const long long gi = my_offset + i;
  const PosRecord me = rec[gi];
  const Cell      c  = cell_of(me.x, me.y, me.z);
  double fx = 0.0, fy = 0.0, fz = 0.0;
  bool   crashed = false;
  if (c.ok) {
    long long prev = -1;
    for (;;) {
      long long best = n_total;  // smallest qualifying partner index > prev
      for (int dx = -1; dx <= 1; dx++)
        for (int dy = -1; dy <= 1; dy++)
          for (int dz = -1; dz <= 1; dz++) {
            const int      cx = c.x + dx, cy = c.y + dy, cz = c.z + dz;
            const uint32_t b  = bucket_of(cx, cy, cz, mask);
            const uint32_t s0 = start[b], cn = count[b];
            for (uint32_t e = 0; e < cn; e++) {
              const long long j = sorted[s0 + e];
              if (j <= prev || j >= best || j == gi) continue;  // buckets are index-ordered but cells interleave
              const PosRecord o  = rec[j];
              const Cell      oc = cell_of(o.x, o.y, o.z);
              if (oc.x != cx || oc.y != cy || oc.z != cz) continue;  // other cell sharing the bucket
              const double d0 = me.x - o.x, d1 = me.y - o.y, d2 = me.z - o.z;
              const double dist = ((0.0 + d0 * d0) + d1 * d1) + d2 * d2;
              if (!(dist < 3.0)) continue;
              const double crit_ij = ((me.arm_length + me.prop_radius) + o.arm_length) + o.prop_radius;
              const double crit_ji = ((o.arm_length + o.prop_radius) + me.arm_length) + me.prop_radius;
              if (dist < crit_ij || (crash && dist < crit_ji)) best = j;
            }
          }
      if (best >= n_total) break;
      const PosRecord o  = rec[best];
      const double    d0 = me.x - o.x, d1 = me.y - o.y, d2 = me.z - o.z;
      const double    dist    = ((0.0 + d0 * d0) + d1 * d1) + d2 * d2;
      const double    crit_ij = ((me.arm_length + me.prop_radius) + o.arm_length) + o.prop_radius;
      const double    crit_ji = ((o.arm_length + o.prop_radius) + me.arm_length) + me.prop_radius;
      if (crash) {
        // the reference crashes the partner of every qualifying ordered pair (i -> idx); seen from the partner's
        // side: this UAV is crashed iff some j has it as a qualifying partner, i.e. dist < crit(j, i)
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
      prev = best;
    }
  }
  sw.S[(size_t)(F_FEXT + 0) * sw.npad + i] = fx;
  sw.S[(size_t)(F_FEXT + 1) * sw.npad + i] = fy;
  sw.S[(size_t)(F_FEXT + 2) * sw.npad + i] = fz;
  if (crashed) sw.F[i] |= FLAG_CRASHED;
}

}  // namespace

struct CollideWork {
  long long cap_n = 0;
  uint32_t  cap_T = 0;
  uint32_t *key = nullptr, *rank = nullptr, *sorted = nullptr, *count = nullptr, *start = nullptr, *bsum = nullptr;
};

extern "C" void mrs_collide_free(CollideWork* w) {
  if (!w) return;
  (void)hipFree(w->key); (void)hipFree(w->rank); (void)hipFree(w->sorted); (void)hipFree(w->count); (void)hipFree(w->start); (void)hipFree(w->bsum);
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

extern "C" hipError_t mrs_collide_run(SwarmDev sw, CollideWork** work, const PosRecord* rec, long long n_total, long long my_offset,
                                      int crash, double rebounce, hipStream_t st) {
  if (!*work) *work = new CollideWork();
  CollideWork* w = *work;
  uint32_t     T = 1024;
  while ((long long)T < 2 * n_total) T <<= 1;
  if (n_total > w->cap_n || T > w->cap_T) {
    CK(hipStreamSynchronize(st));
    (void)hipFree(w->key); (void)hipFree(w->rank); (void)hipFree(w->sorted); (void)hipFree(w->count); (void)hipFree(w->start); (void)hipFree(w->bsum);
    CK(hipMalloc(&w->key, sizeof(uint32_t) * (size_t)n_total));
    CK(hipMalloc(&w->rank, sizeof(uint32_t) * (size_t)n_total));
    CK(hipMalloc(&w->sorted, sizeof(uint32_t) * (size_t)n_total));
    CK(hipMalloc(&w->count, sizeof(uint32_t) * (size_t)T));
    CK(hipMalloc(&w->start, sizeof(uint32_t) * (size_t)T));
    CK(hipMalloc(&w->bsum, sizeof(uint32_t) * (size_t)(T / 1024)));
    w->cap_n = n_total;
    w->cap_T = T;
  }
  const uint32_t mask    = T - 1;
  const int      nblocks = (int)(T / 1024);
  const unsigned gN      = (unsigned)((n_total + 255) / 256);
  CK(hipMemsetAsync(w->count, 0, sizeof(uint32_t) * (size_t)T, st));
  hipLaunchKernelGGL(k_hash_count, dim3(gN), dim3(256), 0, st, rec, n_total, mask, w->key, w->rank, w->count);
  hipLaunchKernelGGL(k_scan_block_sums, dim3(nblocks), dim3(256), 0, st, w->count, w->bsum);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, st, w->bsum, nblocks);
  hipLaunchKernelGGL(k_scan_finish, dim3(nblocks), dim3(256), 0, st, w->count, w->bsum, w->start);
  hipLaunchKernelGGL(k_scatter, dim3(gN), dim3(256), 0, st, n_total, w->key, w->rank, w->start, w->sorted);
  hipLaunchKernelGGL(k_sort_buckets, dim3(T / 256), dim3(256), 0, st, T, w->count, w->start, w->sorted);
  hipLaunchKernelGGL(k_query, dim3((sw.n + 255) / 256), dim3(256), 0, st, sw, rec, n_total, my_offset, mask, w->count, w->start,
                     w->sorted, crash, rebounce);
  return hipGetLastError();
}
