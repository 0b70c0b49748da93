// K2 — spatial-hash exact 1-NN with an exact hierarchical fallback (the production NN path).
//
// Replaces nanoflann's KD-tree (include/nanoflann.hpp:859-867 build, :900-911/:1199-1247 search) behind
// Frame::getClosestPoint (src/internal/frame.cpp:187-206) and the per-query transform of
// Frame::computeClosestPointsToNeighbours (frame.cpp:117-118,131,136).  Same bit-exact contract as K1:
// fp64 distance d0*d0+d1*d1+d2*d2 left to right, no fma (include/frame.h:70-76).  Inside the kernels the lowest original
// index wins exact ties — independent of traversal order because every comparison is the total order (d2, index) — and a
// query whose best distance was met twice (second == best) is reported to nn_tie.hip, which decides it the way nanoflann does.
//
// Per cloud, built ONCE at upload (clouds are static in their local frame, like the reference's lazily
// built tree, frame.cpp:188-193):
//   * points in a balanced k-d ORDER (kd_order below; grid_curve 0 / 1: Morton / Hilbert index of the grid cell instead):
//     `spts` (sorted xyz) + `sidx` (original index) + `srec` (both, 32-B records).  The sorted order is also the QUERY
//     order of a source cloud: neighbouring lanes ask about neighbouring places, so hash slots / point runs / tree nodes
//     are shared inside a wave and stay in L2.
//   * an open-addressing hash table  cell -> (start, count)  (16-B entries, <= 50 % load): the spatial hash (cell edge
//     h ~ a few point spacings).  Its runs index `crec`, the records in Hilbert-of-cells order (the same array as `srec`
//     when that IS the sorted order).
//   * an implicit complete 8-ary box tree over the sorted array (leaf j = the aligned run of 8 / 16 / 32 sorted points
//     [j L, (j+1) L), float AABBs rounded OUTWARD, 32 B per node, heap-indexed: no pointers).
// Query = (0) temporal cache: if last round's neighbour is provably still nearest, re-evaluate its distance and stop;
// (1) scan the 2x2x2 cell block around the query through the hash; the best candidate is provably the global NN iff
// it is closer than the distance to the block's faces (>= h/2); (2) otherwise the query joins a compacted far list
// and an octet of lanes runs an exact branch-and-bound descent of the 8-ary tree seeded with that candidate (or with
// the cutoff bound: matches at or beyond the cutoff are discarded by frame.cpp:156 anyway).
// Pruning is exact in floating point: the box lower bound is evaluated with the SAME rounded operations as
// the point distance, and every rounding is monotone, so lb <= d2 for every point in the box; nodes are
// skipped only when lb > best (ties are still visited for the index rule).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <system_error>
#include <thread>
#include <unordered_set>

#include "common.h"
#include "nn_list.h"
#include "nn_tie.h"

namespace mvicp {

namespace {

#ifndef MVICP_GRID_THREADS
#define MVICP_GRID_THREADS 256
#endif
constexpr int NT = MVICP_GRID_THREADS;
constexpr unsigned long long EMPTY = ~0ull;

struct HashEntry { unsigned long long key; unsigned int start, count; };

struct BrickEntry { unsigned long long mask; unsigned int tab; unsigned int pad; };   // 4x4x4 cells: bit (x&3) | (y&3)<<2 | (z&3)<<4

struct GridView {  // device view of one cloud's structure
  const double* spts; const int* sidx; const PointRec* srec; int n;   // canonical (sorted-position) order
  const PointRec* crec;                                                  // the records in CELL order: what the hash runs index (== srec for the curve orders)
  const HashEntry* table; unsigned int mask; int shift;
  double ox, oy, oz, h, inv_h;
  int dx, dy, dz;
  const float* oct; int oct_leaf; long long oct_first_leaf;    // implicit 8-ary box tree (32-B boxes), phase 2; leaf j = sorted points [j * oct_leaf, (j + 1) * oct_leaf)
  const BrickEntry* bricks; const uint2* celltab; int bx, by, bz;   // dense brick map (null: none), nn_cell_kernel
};

struct GridJob {
  GridView dst;
  const double* q; const int* qidx;  // queries (sorted source points + their original index) or raw queries (qidx null)
  const double* xf;                  // kEdgeXf or null
  int n;
  int* out_idx; double* out_d2;
  const int* inv;      // target original index -> sorted position (null: emit original indices, raw-query API)
  // "did anything change?" bookkeeping of the edge's compacted list (all null for the raw-query API)
  const int* qpos; int* second; double* cd2; const int* dirty; int* dirty_slots;  // dirty: host-forced flag; slots: one per NT queries
  double* stream; long long total_cap; const double* dst_nor;   // the edge's slice of the packed operand stream (linearize.hip) + sorted dst normals
  float* out_lb;       // per query: lower bound on the distance to every target other than out_idx, fp32 ROUNDED DOWN (null: no cache)
  int seed;            // out_idx still holds last round's neighbours (from any kernel): a starting candidate for far queries
  TieRef tie;          // where queries whose best distance was met by more than one target are reported (nn_tie.h)
};

__host__ __device__ __forceinline__ unsigned long long cell_key(int ix, int iy, int iz) {
  return (unsigned long long)ix | ((unsigned long long)iy << 21) | ((unsigned long long)iz << 42);
}
__host__ __device__ __forceinline__ unsigned int hash_slot(unsigned long long k, int shift) {
  return (unsigned int)((k * 0x9E3779B97F4A7C15ull) >> shift);
}

__device__ __forceinline__ void xf_point(const double* __restrict__ x, double p0, double p1, double p2, double& q0, double& q1, double& q2) {
  double g[3], u[3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
    g[i] = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(x[i], p0), __dmul_rn(x[i + 3], p1)), __dmul_rn(x[i + 6], p2)), x[9 + i]);
#pragma unroll
  for (int i = 0; i < 3; ++i) u[i] = __dsub_rn(g[i], x[21 + i]);
  q0 = __dadd_rn(__dadd_rn(__dmul_rn(x[12 + 0], u[0]), __dmul_rn(x[12 + 3], u[1])), __dmul_rn(x[12 + 6], u[2]));
  q1 = __dadd_rn(__dadd_rn(__dmul_rn(x[12 + 1], u[0]), __dmul_rn(x[12 + 4], u[1])), __dmul_rn(x[12 + 7], u[2]));
  q2 = __dadd_rn(__dadd_rn(__dmul_rn(x[12 + 2], u[0]), __dmul_rn(x[12 + 5], u[1])), __dmul_rn(x[12 + 8], u[2]));
}

__device__ __forceinline__ double dist2(double qx, double qy, double qz, double x, double y, double z) {
  const double d0 = __dsub_rn(qx, x), d1 = __dsub_rn(qy, y), d2 = __dsub_rn(qz, z);
  return __dadd_rn(__dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1)), __dmul_rn(d2, d2));
}

// sum over the ACTIVE lanes of the wave (lanes may have exited early)
__device__ __forceinline__ unsigned long long __reduce_add_u64(unsigned long long v) {
  unsigned long long total = 0;
  unsigned long long mask = __ballot(1);
  while (mask) {
    const int l = __ffsll((long long)mask) - 1;
    total += __shfl(v, l, 64);
    mask &= mask - 1;
  }
  return total;
}
__device__ __forceinline__ bool __lane0() {
  const unsigned long long mask = __ballot(1);
  return (int)(threadIdx.x & 63) == __ffsll((long long)mask) - 1;
}

// list maintenance shared with the tile kernel: nn_list.h
__device__ __forceinline__ void update_list(const GridJob& job, int i, int idx_new, double d2_new, double bound, bool same_neighbour = false) {
  const ListRef R{job.qpos, job.second, job.cd2, job.dirty, job.dirty_slots, job.stream, job.total_cap, job.dst_nor, job.dst.srec};
  update_list_entry(R, i, idx_new, d2_new, bound, same_neighbour);
}

// Also leaves the slots zeroed for the next round (no memsets in the per-round launch sequence).
__global__ void dirty_reduce_kernel(int E, const int* __restrict__ slot_off, int* __restrict__ slots, int* __restrict__ dirty) {
  const int e = blockIdx.x;
  int any = 0;
  for (int k = slot_off[e] + threadIdx.x; k < slot_off[e + 1]; k += blockDim.x) {
    const int v = slots[k];
    if (v) { any = 1; slots[k] = 0; }
  }
  if (__syncthreads_or(any) && threadIdx.x == 0) dirty[e] = 1;
}

// ---- phase 1: spatial-hash block lookup for every query; queries it cannot prove optimal go to the far list ----
template <bool TREE_ONLY>
__global__ __launch_bounds__(NT) void nn_grid_kernel(const GridJob* __restrict__ jobs, double bound, double search, unsigned long long* __restrict__ stats, int skip_far,
                                                     int2* __restrict__ far_list, unsigned int* __restrict__ far_count, double prune_rho) {
  __shared__ double sxf[kEdgeXf];
  __shared__ uint2 s_rng[8][NT];   // per lane: (start, count) of the 8 block cells (written and read by the same thread only)
  const GridJob& job = jobs[blockIdx.y];
  const int i = blockIdx.x * NT + threadIdx.x;
  if (blockIdx.x * NT >= job.n) return;
  const bool has_xf = job.xf != nullptr;
  if (has_xf && threadIdx.x < kEdgeXf) sxf[threadIdx.x] = job.xf[threadIdx.x];
  __syncthreads();
  if (i >= job.n) return;
  const GridView& g = job.dst;

  double qx, qy, qz;
  {
    const double p0 = job.q[3 * (size_t)i], p1 = job.q[3 * (size_t)i + 1], p2 = job.q[3 * (size_t)i + 2];
    if (has_xf) xf_point(sxf, p0, p1, p2, qx, qy, qz);
    else { qx = p0; qy = p1; qz = p2; }
  }
  const int out = i;   // results live in the SORTED order of the source cloud (coalesced; the pipeline stays in that order)

  double best = search;     // nothing at or beyond the search radius (>= the cutoff: search_bound()) needs resolving; frame.cpp:156 filters later
  int bi = 0x7fffffff;
  bool resolved = false;
  unsigned int n_cand = 0;
  double second = 1.7976931348623157e308;

  // ---- temporal cache.  Last search left, per query, its neighbour p1 and a lower bound L on the distance to every
  // OTHER target.  Since then the query moved by at most eps (pose update), so every other target is still >= L - eps
  // away; if the re-evaluated distance to p1 is strictly below that, p1 is still the unique nearest neighbour and its
  // exact squared distance (reference arithmetic) is the answer — no search.  Relative 1e-12 slack covers sqrt rounding.
  const double slack = has_xf ? sxf[24] : -1.0;
  // Last round's neighbour p1 (sorted position in out_idx, left there by whichever kernel ran) serves twice:
  //  * temporal cache (needs last round's lower bounds, i.e. a grid round): re-evaluate the distance and stop;
  //  * otherwise its distance bounds the search: the true neighbour lies within |q - p1| of q, so hash cells of the block
  //    farther than r_p = |q - p1| + rho need not be probed.  Everything in a skipped cell is farther than r_p, which
  //    enters the lower bound handed to the next round's cache: the margin rho (a fraction of the cell edge) is what lets
  //    that bound survive the next pose update.
  double rp2 = -1.0;   // < 0: scan the whole block
  if (!TREE_ONLY && job.seed) {
    const int pi = job.out_idx[out];   // sorted position of last round's neighbour
    if (pi < 0 && slack == 0.0 && job.out_lb != nullptr && job.out_lb[out] == -1.f) {
      // last search found NO target within the search radius, and the host found this edge's query transform bit-identical to that
      // search's (slack 0: dM = dv = 0): the same query has the same answer — nothing to search, nothing to write
      if (stats) {
        unsigned long long c1 = __reduce_add_u64(1ull);
        const size_t slot = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (NT / 64) + (threadIdx.x >> 6);
        if (__lane0()) atomicAdd(&stats[8 * slot + 3], c1);
      }
      return;
    }
    if (pi >= 0 && pi < g.n) {
      // (the 24-B point of the sorted cloud, not its 32-B record: a cache hit needs no index — round 3 byte diet, 68 -> 56 B per hit)
      const double* tp = g.spts + 3 * (size_t)pi;
      const double d = dist2(qx, qy, qz, tp[0], tp[1], tp[2]);
      if (slack >= 0.0 && job.out_lb != nullptr) {
        // how far THIS query moved since the last search: |dM p + dv| (exactly, up to the rounding allowance)
        const double p0 = job.q[3 * (size_t)i], p1 = job.q[3 * (size_t)i + 1], p2 = job.q[3 * (size_t)i + 2];
        const double e0 = sxf[25] * p0 + sxf[28] * p1 + sxf[31] * p2 + sxf[34];
        const double e1 = sxf[26] * p0 + sxf[29] * p1 + sxf[32] * p2 + sxf[35];
        const double e2 = sxf[27] * p0 + sxf[30] * p1 + sxf[33] * p2 + sxf[36];
        const double eps = sqrt(e0 * e0 + e1 * e1 + e2 * e2) * (1.0 + 1e-9) + slack;
        const double nlb = (double)job.out_lb[out] - eps;
        // eps == 0 only when the host found the edge's query transform bit-identical to last search's (slack 0, dM = dv = 0): the query is
        // the same bit for bit, so last search's exact answer — whatever its bound — its distance, its bound and its list entry are exactly
        // what is stored already: nothing to search, nothing to write
        if (eps == 0.0 || sqrt(d) * (1.0 + 1e-12) < nlb) {
          if (eps != 0.0) {
            job.out_d2[out] = d;
            job.out_lb[out] = __double2float_rd(nlb);
            if (job.dirty) update_list(job, i, pi, d, bound, true);
          }
          if (stats) {
            unsigned long long c1 = __reduce_add_u64(1ull);
            const size_t slot = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (NT / 64) + (threadIdx.x >> 6);
            if (__lane0()) atomicAdd(&stats[8 * slot + 3], c1);
          }
          return;
        }
      }
      if (prune_rho > 0.0) { const double rp = sqrt(d) + prune_rho * g.h; rp2 = rp * rp * 1.002; }
    }
  }

  double m2 = 0.0;
  double skipped = 1.7976931348623157e308;   // smallest distance bound among the block cells not probed
  if (!TREE_ONLY) {
    // 2x2x2 block of cells nearest to the query, through the spatial hash
    const double cx = (qx - g.ox) * g.inv_h - 0.5, cy = (qy - g.oy) * g.inv_h - 0.5, cz = (qz - g.oz) * g.inv_h - 0.5;
    // queries far outside the grid cannot be resolved by the block test; clamp so the int conversion is safe
    const double lim = 2.0e6;
    const int bx = (int)floor(fmin(fmax(cx, -lim), lim)), by = (int)floor(fmin(fmax(cy, -lim), lim)), bz = (int)floor(fmin(fmax(cz, -lim), lim));
    // all 8 first probes are issued together (independent 16-B loads), collisions are resolved afterwards
    unsigned long long key[8];
    unsigned int slot[8];
    HashEntry ent[8];
    // squared distance from q to each cell of the block: per axis 0 on q's side of the block's mid-plane, else the distance
    // to that plane (0.2 % slack dwarfs the rounding of the cell assignment, like `m` below)
    const double ax = qx - (g.ox + (bx + 1) * g.h), ay = qy - (g.oy + (by + 1) * g.h), az = qz - (g.oz + (bz + 1) * g.h);
    const double ax2 = ax * ax * 0.998, ay2 = ay * ay * 0.998, az2 = az * az * 0.998;
    const int home = (ax >= 0.0 ? 1 : 0) | (ay >= 0.0 ? 2 : 0) | (az >= 0.0 ? 4 : 0);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int ix = bx + (c & 1), iy = by + ((c >> 1) & 1), iz = bz + (c >> 2);
      bool in = !(ix < 0 || iy < 0 || iz < 0 || ix >= g.dx || iy >= g.dy || iz >= g.dz);
      if (rp2 >= 0.0) {
        const int df = c ^ home;
        const double lbc = ((df & 1) ? ax2 : 0.0) + ((df & 2) ? ay2 : 0.0) + ((df & 4) ? az2 : 0.0);
        if (lbc > rp2) { in = false; skipped = fmin(skipped, lbc); }
      }
      key[c] = in ? cell_key(ix, iy, iz) : EMPTY;
      slot[c] = hash_slot(key[c], g.shift) & g.mask;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      ent[c].key = EMPTY; ent[c].start = 0; ent[c].count = 0;
      if (key[c] != EMPTY) ent[c] = g.table[slot[c]];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      while (ent[c].key != key[c] && ent[c].key != EMPTY) {
        slot[c] = (slot[c] + 1) & g.mask;
        ent[c] = g.table[slot[c]];
      }
      if (ent[c].key == EMPTY) ent[c].count = 0;
    }
    // One flattened candidate loop per lane.  Scanning cell after cell would cost, per wave, the SUM over the 8 cells of the
    // longest run any lane has in that cell; walking each lane's own runs back to back costs the longest TOTAL of any lane
    // (about half), and lets the cells pruned above actually save time.  The (start, count) pairs go through LDS because
    // a register array cannot be indexed per lane.
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      s_rng[c][threadIdx.x] = make_uint2(ent[c].start, ent[c].count);
      n_cand += ent[c].count;
    }
    {
      int c = -1;
      unsigned int j = 0, je = 0;
      for (;;) {
        while (j == je && c < 7) { ++c; const uint2 r = s_rng[c][threadIdx.x]; j = r.x; je = r.x + r.y; }
        if (j == je) break;
        const double2* p = reinterpret_cast<const double2*>(g.crec + j);  // two 16-B loads per candidate
        const double2 a = p[0], b = p[1];
        const double d = dist2(qx, qy, qz, a.x, a.y, b.x);
        const int oi = (int)__double_as_longlong(b.y);
        if (d < best || (d == best && oi < bi)) {
          if (bi != 0x7fffffff) second = fmin(second, best);
          best = d; bi = oi;
        } else {
          second = fmin(second, d);
        }
        ++j;
      }
    }
    // every point outside the block differs from q by at least `m` along some axis (cells are assigned with
    // the same rounded expression; 0.1 % slack dwarfs any rounding in it)
    const double fx = g.ox + bx * g.h, fy = g.oy + by * g.h, fz = g.oz + bz * g.h;
    double m = fmin(fmin(qx - fx, fx + 2.0 * g.h - qx), fmin(fmin(qy - fy, fy + 2.0 * g.h - qy), fmin(qz - fz, fz + 2.0 * g.h - qz)));
    m *= 0.999;
    m2 = m > 0.0 ? m * m : 0.0;
    resolved = (m > 0.0) && (best < m2) && (bi != 0x7fffffff);
  }
  if (!TREE_ONLY && !resolved && bi == 0x7fffffff && job.seed) {
    // nothing in the block: rather than starting the tree descent from the cutoff bound, start it from last round's
    // neighbour (an ordinary candidate; phase 2 knows how to meet its own seed again)
    const int pi = job.out_idx[out];
    if (pi >= 0 && pi < g.n) {
      const double2* tp = reinterpret_cast<const double2*>(g.srec + pi);
      const double2 ta = tp[0], tb = tp[1];
      const double d = dist2(qx, qy, qz, ta.x, ta.y, tb.x);
      if (d < best) { best = d; bi = (int)__double_as_longlong(tb.y); }
    }
  }
  // provisional (or final) result; phase 2 re-reads it as the seed of the tree descent
  job.out_idx[out] = bi == 0x7fffffff ? -1 : (job.inv ? job.inv[bi] : bi);
  job.out_d2[out] = best;
  // every other target is either a scanned candidate (>= second) or outside the block (>= m)
  if (job.out_lb != nullptr) job.out_lb[out] = resolved ? __double2float_rd(sqrt(fmin(fmin(second, m2), skipped)) * (1.0 - 1e-12)) : 0.f;
  if (job.dirty && (resolved || skip_far)) update_list(job, i, bi == 0x7fffffff ? -1 : job.inv[bi], best, bound);
  if (resolved && bi != 0x7fffffff && second == best) tie_report(job.tie, (unsigned int)i);   // another target at exactly the same distance
  if (!resolved && !skip_far) {
    // wave-aggregated append: one atomic per wave
    const unsigned long long mask = __ballot(1);
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(far_count, (unsigned int)__popcll(mask));
    base = __shfl(base, leader, 64);
    far_list[base + rank] = make_int2((int)blockIdx.y, i);
  }
  if (stats) {
    // candidate census for the algorithmic-byte model: one slot per wave, no atomics
    // (hot-address atomics would throttle the kernel being measured); summed by census_sum_kernel.
    unsigned long long c = n_cand, far = resolved ? 0 : 1;
    c = __reduce_add_u64(c); far = __reduce_add_u64(far);
    const size_t slot = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (NT / 64) + (threadIdx.x >> 6);
    if (__lane0()) { atomicAdd(&stats[8 * slot], c); atomicAdd(&stats[8 * slot + 2], far); }
  }
}


// ---- phase 1 on clouds with a brick map: WAVE-COOPERATIVE CELL STAGING -------------------------------------------------------
// nn_grid_kernel answers a query per lane from global memory: 8 hash probes + two 16-B loads per candidate record, ~40 divergent
// loads per lane, which is what bounds it (the texture addresser retires about one distinct cache line per cycle).  But the 64
// queries of a wave are curve-adjacent source points: their searches touch the SAME few dozen cells.  Here the wave cooperates:
//   A. every lane names the occupied cells of its 2x2x2 home block (dense brick occupancy masks: no probes for empty cells; cells
//      that cannot beat last round's neighbour are dropped) and inserts them into a small hash SET in LDS (64-bit atomicCAS);
//      each DISTINCT cell is then looked up once per wave (brick entry -> cell table {start, count}) and its point run copied
//      into LDS; every lane scans its own cells from LDS in the reference's fp64 arithmetic, (d2, original index) total order;
//   B. the running best bounds the answer: the true neighbour lies in the ball B(q, sqrt(best)).  Lanes whose ball (plus the
//      margin rho * h that keeps next round's temporal-cache bound useful) reaches beyond the home block repeat the same three
//      steps for the cells of the ball outside the block.  The ball only shrinks afterwards, so two passes always suffice.
// A lane is resolved by construction: every target it did not scan lies outside its ball or in a cell whose lower bound exceeded
// the running best.  Lanes with no candidate in the home block, with a ball wider than CELL_SPAN cells, or that do not fit the
// workgroup's LDS budget join the far list with their best candidate so far and are finished by nn_far_kernel — exactness never
// depends on the budget.
constexpr int CELL_HS = 128;      // hash-set slots per wave
constexpr int CELL_POOL = 960;    // staged points per workgroup (3 x 8 B each), shared by its waves through a bump allocator
constexpr int CELL_SPAN = 6;      // widest per-lane ball box, cells per axis
constexpr unsigned int CELL_UNSET = 0xfffffffeu, CELL_NOFIT = 0xffffffffu;

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Squared distance from q to the cell (ix, iy, iz), shrunk by `tol` per axis: a point whose cell index was rounded across a
// boundary (the assignment floor((p - o) * inv_h) carries a relative 1e-16 error, i.e. < 1e-9 h) still satisfies d2 >= lb2.
__device__ __forceinline__ double cell_lb2(const GridView& g, int ix, int iy, int iz, double qx, double qy, double qz, double tol) {
  const double lx = g.ox + ix * g.h, ly = g.oy + iy * g.h, lz = g.oz + iz * g.h;
  const double gx = fmax(fmax(lx - qx, qx - (lx + g.h)) - tol, 0.0);
  const double gy = fmax(fmax(ly - qy, qy - (ly + g.h)) - tol, 0.0);
  const double gz = fmax(fmax(lz - qz, qz - (lz + g.h)) - tol, 0.0);
  return (gx * gx + gy * gy + gz * gz) * 0.999999;
}

__device__ __forceinline__ bool cell_insert(unsigned long long* __restrict__ hk, unsigned long long key) {
  unsigned int s = hash_slot(key, 64 - 7) & (CELL_HS - 1);
  for (int probe = 0; probe < 24; ++probe) {
    const unsigned long long old = atomicCAS(&hk[s], EMPTY, key);
    if (old == EMPTY || old == key) return true;
    s = (s + 1) & (CELL_HS - 1);
  }
  return false;   // set (nearly) full
}
__device__ __forceinline__ int cell_find(const unsigned long long* __restrict__ hk, unsigned long long key) {
  unsigned int s = hash_slot(key, 64 - 7) & (CELL_HS - 1);
  for (int probe = 0; probe < 24; ++probe) {
    const unsigned long long k = hk[s];
    if (k == key) return (int)s;
    if (k == EMPTY) return -1;
    s = (s + 1) & (CELL_HS - 1);
  }
  return -1;
}

struct CellLane {   // per-lane search state
  double qx, qy, qz, best, second;
  int bpos;          // SORTED position of the running best (-1: none)
  unsigned int n_cand;
  bool ovf;
};

// occupancy bit of cell (ix, iy, iz); `be` / `last_b` cache the brick entry across consecutive cells
__device__ __forceinline__ bool cell_occupied(const GridView& g, int ix, int iy, int iz, BrickEntry& be, long long& last_b) {
  const long long b = ((long long)(iz >> 2) * g.by + (iy >> 2)) * g.bx + (ix >> 2);
  if (b != last_b) { be = g.bricks[b]; last_b = b; }
  return (be.mask >> ((ix & 3) | ((iy & 3) << 2) | ((iz & 3) << 4))) & 1ull;
}

// stage every cell of the wave's set that has not been staged yet: one lookup + one copy per distinct cell
__device__ __forceinline__ unsigned int cell_stage(const GridView& g, const unsigned long long* __restrict__ hk, unsigned int* __restrict__ hv,
                                                   unsigned int* __restrict__ hs, double* __restrict__ sx, double* __restrict__ sy, double* __restrict__ sz,
                                                   unsigned int* __restrict__ pool, int lane, unsigned int* n_pts) {
  unsigned int staged = 0;
  for (int s = lane; s < CELL_HS; s += 64) {
    const unsigned long long key = hk[s];
    if (key == EMPTY || hv[s] != CELL_UNSET) continue;
    const int ix = (int)(key & 0x1fffffull), iy = (int)((key >> 21) & 0x1fffffull), iz = (int)((key >> 42) & 0x1fffffull);
    const long long b = ((long long)(iz >> 2) * g.by + (iy >> 2)) * g.bx + (ix >> 2);
    const unsigned int tab = g.bricks[b].tab;
    const uint2 run = g.celltab[(size_t)tab * 64 + ((ix & 3) | ((iy & 3) << 2) | ((iz & 3) << 4))];
    const unsigned int off = run.y <= (unsigned int)CELL_POOL ? atomicAdd(pool, run.y) : (unsigned int)CELL_POOL;
    if (off + run.y > (unsigned int)CELL_POOL) { hv[s] = CELL_NOFIT; continue; }
    hs[s] = run.x;
    const double* __restrict__ src = g.spts + 3 * (size_t)run.x;
    for (unsigned int k = 0; k < run.y; ++k) { sx[off + k] = src[3 * k]; sy[off + k] = src[3 * k + 1]; sz[off + k] = src[3 * k + 2]; }
    hv[s] = (off << 16) | run.y;
    ++staged; *n_pts += run.y;
  }
  return staged;
}

// scan one staged cell for the lane
__device__ __forceinline__ void cell_scan(const GridView& g, CellLane& L, const unsigned long long* __restrict__ hk, const unsigned int* __restrict__ hv,
                                          const unsigned int* __restrict__ hs, const double* __restrict__ sx, const double* __restrict__ sy,
                                          const double* __restrict__ sz, unsigned long long key) {
  const int s = cell_find(hk, key);
  if (s < 0) { L.ovf = true; return; }
  const unsigned int v = hv[s];
  if (v == CELL_NOFIT || v == CELL_UNSET) { L.ovf = true; return; }
  const unsigned int off = v >> 16, cnt = v & 0xffffu;
  const int start = (int)hs[s];
  for (unsigned int k = 0; k < cnt; ++k) {
    const int pos = start + (int)k;
    if (pos == L.bpos) continue;   // the running best met again (last round's neighbour is an ordinary target of its cell)
    const double d = dist2(L.qx, L.qy, L.qz, sx[off + k], sy[off + k], sz[off + k]);
    if (d < L.best) {
      if (L.bpos >= 0) L.second = fmin(L.second, L.best);
      L.best = d; L.bpos = pos;
    } else {
      L.second = fmin(L.second, d);
      if (d == L.best && L.bpos >= 0 && g.sidx[pos] < g.sidx[L.bpos]) L.bpos = pos;   // exact tie: lowest ORIGINAL index (rare: two global loads)
    }
  }
  L.n_cand += cnt;
}

__global__ __launch_bounds__(NT, 5) void nn_cell_kernel(const GridJob* __restrict__ jobs, double bound, double search_r2, unsigned long long* __restrict__ stats,
                                                        int2* __restrict__ far_list, unsigned int* __restrict__ far_count, double prune_rho) {
  __shared__ double sxf[kEdgeXf];
  __shared__ unsigned long long s_hkey[NT / 64][CELL_HS];
  __shared__ unsigned int s_hval[NT / 64][CELL_HS];      // (LDS offset << 16) | count; CELL_UNSET: not staged yet; CELL_NOFIT: pool exhausted
  __shared__ unsigned int s_hstart[NT / 64][CELL_HS];    // sorted position of the cell's first point
  __shared__ double s_x[CELL_POOL], s_y[CELL_POOL], s_z[CELL_POOL];
  __shared__ unsigned int s_pool;
  __shared__ unsigned int s_stat[NT / 64][3];
  const GridJob& job = jobs[blockIdx.y];
  if (blockIdx.x * NT >= job.n) return;
  if (threadIdx.x < kEdgeXf) sxf[threadIdx.x] = job.xf[threadIdx.x];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long* hk = s_hkey[wave];
  unsigned int* hv = s_hval[wave];
  unsigned int* hs = s_hstart[wave];
  hk[lane] = EMPTY; hk[lane + 64] = EMPTY;
  hv[lane] = CELL_UNSET; hv[lane + 64] = CELL_UNSET;
  if (threadIdx.x == 0) s_pool = 0u;
  if (lane < 3) s_stat[wave][lane] = 0u;
  __syncthreads();
  const int i = blockIdx.x * NT + threadIdx.x;
  const bool active = i < job.n;
  const GridView& g = job.dst;
  const double big = 1.7976931348623157e308;

  CellLane L;
  L.qx = L.qy = L.qz = 0.0;
  L.best = search_r2;    // nothing at or beyond the search radius needs resolving (frame.cpp:156 filters later)
  L.bpos = -1; L.second = big; L.n_cand = 0; L.ovf = false;
  double TA2 = big;      // pass A scans the home cells within this squared distance (seeded: old-neighbour distance + margin)
  double p0 = 0.0, p1 = 0.0, p2 = 0.0;
  if (active) {
    p0 = job.q[3 * (size_t)i]; p1 = job.q[3 * (size_t)i + 1]; p2 = job.q[3 * (size_t)i + 2];
    xf_point(sxf, p0, p1, p2, L.qx, L.qy, L.qz);
  }
  bool hit = false;
  const double tol = 1e-7 * g.h;
  if (active && job.seed) {
    const int pi = job.out_idx[i];   // sorted position of last round's neighbour
    if (pi >= 0 && pi < g.n) {
      const double* tp = g.spts + 3 * (size_t)pi;
      const double d = dist2(L.qx, L.qy, L.qz, tp[0], tp[1], tp[2]);
      const double slack = sxf[24];
      if (slack >= 0.0 && job.out_lb != nullptr) {
        // temporal cache (see nn_grid_kernel): how far THIS query moved since the last search is |dM p + dv|
        const double e0 = sxf[25] * p0 + sxf[28] * p1 + sxf[31] * p2 + sxf[34];
        const double e1 = sxf[26] * p0 + sxf[29] * p1 + sxf[32] * p2 + sxf[35];
        const double e2 = sxf[27] * p0 + sxf[30] * p1 + sxf[33] * p2 + sxf[36];
        const double eps = sqrt(e0 * e0 + e1 * e1 + e2 * e2) * (1.0 + 1e-9) + slack;
        const double nlb = (double)job.out_lb[i] - eps;
        if (sqrt(d) * (1.0 + 1e-12) < nlb) {
          if (eps != 0.0) {   // (eps == 0: bit-identical query transform, everything stored is already exact — see nn_grid_kernel)
            job.out_d2[i] = d;
            job.out_lb[i] = __double2float_rd(nlb);
            if (job.dirty) update_list(job, i, pi, d, bound, true);
          }
          hit = true;
        }
      }
      if (!hit && d < bound) {
        // the old neighbour is an ordinary candidate and bounds the search: the answer is within sqrt(d); cells within rho * h more
        // are scanned too, so that the lower bound left for next round's temporal cache keeps that margin
        L.best = d; L.bpos = pi;
        const double ra = (sqrt(d) * (1.0 + 1e-12) + prune_rho * g.h) * (1.0 + 1e-9) + tol;
        TA2 = ra * ra;
      }
    }
  }
  const bool search = active && !hit;
  bool resolved = false;
  double R1 = 0.0;
  unsigned int n_cells = 0, n_pts = 0;

  if (__ballot(search) != 0ull) {
    // ---- pass A: the 2x2x2 block of cells nearest to the query
    int hx = 0, hy = 0, hz = 0;
    if (search) {
      const double lim = 2.0e6;
      hx = (int)floor(fmin(fmax((L.qx - g.ox) * g.inv_h - 0.5, -lim), lim));
      hy = (int)floor(fmin(fmax((L.qy - g.oy) * g.inv_h - 0.5, -lim), lim));
      hz = (int)floor(fmin(fmax((L.qz - g.oz) * g.inv_h - 0.5, -lim), lim));
      long long last_b = -1;
      BrickEntry be; be.mask = 0ull; be.tab = 0u; be.pad = 0u;
      for (int c = 0; c < 8; ++c) {
        const int ix = hx + (c & 1), iy = hy + ((c >> 1) & 1), iz = hz + (c >> 2);
        if (ix < 0 || iy < 0 || iz < 0 || ix >= g.dx || iy >= g.dy || iz >= g.dz) continue;
        if (!cell_occupied(g, ix, iy, iz, be, last_b)) continue;
        if (cell_lb2(g, ix, iy, iz, L.qx, L.qy, L.qz, tol) > TA2) continue;   // beyond the old neighbour's ball (+ margin)
        if (!cell_insert(hk, cell_key(ix, iy, iz))) L.ovf = true;
      }
    }
    wave_sync();
    n_cells += cell_stage(g, hk, hv, hs, s_x, s_y, s_z, &s_pool, lane, &n_pts);
    wave_sync();
    bool passB = false;
    int cx0 = 0, cx1 = -1, cy0 = 0, cy1 = -1, cz0 = 0, cz1 = -1;
    double R2 = 0.0;
    if (search && !L.ovf) {
      long long last_b = -1;
      BrickEntry be; be.mask = 0ull; be.tab = 0u; be.pad = 0u;
      for (int c = 0; c < 8 && !L.ovf; ++c) {
        const int ix = hx + (c & 1), iy = hy + ((c >> 1) & 1), iz = hz + (c >> 2);
        if (ix < 0 || iy < 0 || iz < 0 || ix >= g.dx || iy >= g.dy || iz >= g.dz) continue;
        if (!cell_occupied(g, ix, iy, iz, be, last_b)) continue;
        if (cell_lb2(g, ix, iy, iz, L.qx, L.qy, L.qz, tol) > TA2) continue;   // everything in it is >= sqrt(TA2) >= R1 away
        cell_scan(g, L, hk, hv, hs, s_x, s_y, s_z, cell_key(ix, iy, iz));
      }
      if (!L.ovf && L.bpos >= 0) {
        // the answer lies in the ball of radius sqrt(best); rho * h more keeps a margin for next round's temporal cache
        R1 = sqrt(L.best) * (1.0 + 1e-12) + prune_rho * g.h;
        const double Rs = R1 * (1.0 + 1e-9) + tol;
        R2 = Rs * Rs;
        const double lim = 2.0e6;
        cx0 = (int)floor(fmin(fmax((L.qx - Rs - g.ox) * g.inv_h, -lim), lim)); cx1 = (int)floor(fmin(fmax((L.qx + Rs - g.ox) * g.inv_h, -lim), lim));
        cy0 = (int)floor(fmin(fmax((L.qy - Rs - g.oy) * g.inv_h, -lim), lim)); cy1 = (int)floor(fmin(fmax((L.qy + Rs - g.oy) * g.inv_h, -lim), lim));
        cz0 = (int)floor(fmin(fmax((L.qz - Rs - g.oz) * g.inv_h, -lim), lim)); cz1 = (int)floor(fmin(fmax((L.qz + Rs - g.oz) * g.inv_h, -lim), lim));
        if (cx0 >= hx && cx1 <= hx + 1 && cy0 >= hy && cy1 <= hy + 1 && cz0 >= hz && cz1 <= hz + 1) {
          resolved = true;   // the ball stays inside the block: done
        } else {
          // cells outside the grid hold nothing (the grid covers the cloud's bounding box)
          cx0 = max(cx0, 0); cy0 = max(cy0, 0); cz0 = max(cz0, 0);
          cx1 = min(cx1, g.dx - 1); cy1 = min(cy1, g.dy - 1); cz1 = min(cz1, g.dz - 1);
          passB = (cx1 - cx0 < CELL_SPAN) && (cy1 - cy0 < CELL_SPAN) && (cz1 - cz0 < CELL_SPAN);
        }
      }
    }
    // ---- pass B: the cells of the ball outside the home block
    if (__ballot(passB) != 0ull) {
      if (passB) {
        long long last_b = -1;
        BrickEntry be; be.mask = 0ull; be.tab = 0u; be.pad = 0u;
        for (int iz = cz0; iz <= cz1; ++iz)
          for (int iy = cy0; iy <= cy1; ++iy)
            for (int ix = cx0; ix <= cx1; ++ix) {
              if (ix >= hx && ix <= hx + 1 && iy >= hy && iy <= hy + 1 && iz >= hz && iz <= hz + 1) continue;
              if (!cell_occupied(g, ix, iy, iz, be, last_b)) continue;
              if (cell_lb2(g, ix, iy, iz, L.qx, L.qy, L.qz, tol) > R2) continue;      // outside the ball
              if (!cell_insert(hk, cell_key(ix, iy, iz))) L.ovf = true;
            }
      }
      wave_sync();
      n_cells += cell_stage(g, hk, hv, hs, s_x, s_y, s_z, &s_pool, lane, &n_pts);
      wave_sync();
      if (passB && !L.ovf) {
        long long last_b = -1;
        BrickEntry be; be.mask = 0ull; be.tab = 0u; be.pad = 0u;
        for (int iz = cz0; iz <= cz1 && !L.ovf; ++iz)
          for (int iy = cy0; iy <= cy1 && !L.ovf; ++iy)
            for (int ix = cx0; ix <= cx1 && !L.ovf; ++ix) {
              if (ix >= hx && ix <= hx + 1 && iy >= hy && iy <= hy + 1 && iz >= hz && iz <= hz + 1) continue;
              if (!cell_occupied(g, ix, iy, iz, be, last_b)) continue;
              if (cell_lb2(g, ix, iy, iz, L.qx, L.qy, L.qz, tol) > R2) continue;      // outside the ball: >= R1 away
              cell_scan(g, L, hk, hv, hs, s_x, s_y, s_z, cell_key(ix, iy, iz));
            }
        resolved = !L.ovf;
      }
    }
  }

  if (search) {
    job.out_idx[i] = L.bpos;
    job.out_d2[i] = L.best;
    // every target that was not scanned is at least R1 away: outside the ball, or in a home cell beyond sqrt(TA2) >= R1
    if (job.out_lb != nullptr) job.out_lb[i] = resolved ? __double2float_rd(sqrt(fmin(L.second, R1 * R1 * (1.0 - 1e-9))) * (1.0 - 1e-12)) : 0.f;
    if (resolved) {
      if (job.dirty) update_list(job, i, L.bpos, L.best, bound);
    } else {
      const unsigned long long mask = __ballot(1);
      const int leader = __ffsll((long long)mask) - 1;
      const int rank = __popcll(mask & ((1ull << lane) - 1ull));
      unsigned int base = 0;
      if (lane == leader) base = atomicAdd(far_count, (unsigned int)__popcll(mask));
      base = __shfl(base, leader, 64);
      far_list[base + rank] = make_int2((int)blockIdx.y, i);
    }
  }
  if (stats) {
    // census (profiling only): candidates scanned, distinct cells staged, far lanes, cache hits — per wave, one slot each
    if (L.n_cand) atomicAdd(&s_stat[wave][0], L.n_cand);
    if (n_cells) { atomicAdd(&s_stat[wave][1], n_cells); atomicAdd(&s_stat[wave][2], n_pts); }
    const unsigned long long fr = __ballot(search && !resolved), hh = __ballot(hit);
    wave_sync();
    if (lane == 0) {
      const size_t slot = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (NT / 64) + wave;
      atomicAdd(&stats[8 * slot], (unsigned long long)s_stat[wave][0]); atomicAdd(&stats[8 * slot + 5], (unsigned long long)s_stat[wave][1]);
      atomicAdd(&stats[8 * slot + 4], (unsigned long long)s_stat[wave][2]);
      atomicAdd(&stats[8 * slot + 2], (unsigned long long)__popcll(fr)); atomicAdd(&stats[8 * slot + 3], (unsigned long long)__popcll(hh));
    }
  }
}

// ---- phase 2: exact branch-and-bound for the compacted far list, EIGHT LANES PER QUERY over an implicit 8-ary
// box tree (children of heap node id are 8 id + 1 .. 8 id + 8; leaf j = sorted points [j n / 8^D, (j+1) n / 8^D)).
// A lane octet loads the 8 child boxes of a node in one coalesced 256-B access (one 32-B box per lane), ranks them by
// lower bound with shuffles and pushes the survivors nearest-on-top onto the octet's LDS stack; leaves are scanned 8
// points at a time with an octet arg-min.  Compared with one lane per query this divides the number of dependent
// memory round trips per query by ~3 (8-ary instead of binary), keeps the loads of an octet contiguous, and only the 8
// octets of a wave can diverge from each other.  Persistent grid-stride launch: the far count lives on the device.
constexpr int OCT_STACK = 56;  // >= 7 * max depth + 1 (build_grid keeps the depth <= 7)

__device__ __forceinline__ double oct_box_lb(double qx, double qy, double qz, const float4 a, const float4 b) {
  // box = {lo.xyz = a.xyz, hi.xyz = (a.w, b.x, b.y)}
  const double g0 = fmax(fmax(__dsub_rn((double)a.x, qx), __dsub_rn(qx, (double)a.w)), 0.0);
  const double g1 = fmax(fmax(__dsub_rn((double)a.y, qy), __dsub_rn(qy, (double)b.x)), 0.0);
  const double g2 = fmax(fmax(__dsub_rn((double)a.z, qz), __dsub_rn(qz, (double)b.y)), 0.0);
  return __dadd_rn(__dadd_rn(__dmul_rn(g0, g0), __dmul_rn(g1, g1)), __dmul_rn(g2, g2));
}

__global__ __launch_bounds__(NT) void nn_far_kernel(const GridJob* __restrict__ jobs, const int2* __restrict__ far_list, double bound,
                                                    const unsigned int* __restrict__ far_count, unsigned int* __restrict__ next_far_count,
                                                    unsigned long long* __restrict__ stats, size_t stats_slots, unsigned int* __restrict__ seen) {
  __shared__ int s_id[NT / 8][OCT_STACK];
  __shared__ double s_lb[NT / 8][OCT_STACK];
  const unsigned int nfar = *far_count;
  // two far-list counters alternate between launches: this launch consumes one, and leaves the OTHER one — which nothing touches
  // during this launch — zeroed for the next phase 1 (no memset and no clean-up kernel in the per-round sequence)
  if (blockIdx.x == 0 && threadIdx.x == 0) { *next_far_count = 0u; if (seen) *seen = nfar; }
  const int oct = threadIdx.x >> 3, l = threadIdx.x & 7;
  const int lane = threadIdx.x & 63, obase = lane & ~7;
  unsigned long long n_cand = 0, n_nodes = 0;
  for (unsigned int f = blockIdx.x * (NT / 8) + oct; f < ((nfar + (NT / 8) - 1) / (NT / 8)) * (NT / 8); f += gridDim.x * (NT / 8)) {
    const bool live = f < nfar;   // whole octet live or not (f is octet-uniform)
    if (!live) continue;
    const int2 item = far_list[f];
    const GridJob& job = jobs[item.x];
    const GridView& g = job.dst;
    const int i = item.y;
    double qx, qy, qz;
    {
      const double p0 = job.q[3 * (size_t)i], p1 = job.q[3 * (size_t)i + 1], p2 = job.q[3 * (size_t)i + 2];
      if (job.xf != nullptr) xf_point(job.xf, p0, p1, p2, qx, qy, qz);  // same rounded operations as phase 1
      else { qx = p0; qy = p1; qz = p2; }
    }
    const int out = i;
    double best = job.out_d2[out];
    int bi = job.out_idx[out];         // seed from phase 1: sorted position (or original index for raw queries)
    if (bi < 0) bi = 0x7fffffff;
    else if (job.inv) bi = (int)g.srec[bi].idx;
    double second = 1.7976931348623157e308;   // smallest d2 among scanned targets other than the running best
    double pruned = 1.7976931348623157e308;   // smallest lower bound among the boxes this lane skipped
    const long long first_leaf = g.oct_first_leaf;
    const int per_leaf = g.oct_leaf;
    int sp = 0;
    if (l == 0) { s_id[oct][0] = 0; s_lb[oct][0] = 0.0; }
    sp = 1;
    while (sp > 0) {
      --sp;
      const int id = s_id[oct][sp];
      const double lbp = s_lb[oct][sp];
      if (lbp > best) { pruned = fmin(pruned, lbp); continue; }
      if (id >= first_leaf) {
        const long long j = (long long)id - first_leaf;
        const int lo = (int)min(j * per_leaf, (long long)g.n), hi = min(lo + per_leaf, g.n);
        double d = 1.7976931348623157e308, ls = 1.7976931348623157e308;  // this lane's best and second best in the leaf
        int oi = 0x7fffffff;
        for (int k = lo + l; k < hi; k += 8) {
          const double2* p = reinterpret_cast<const double2*>(g.srec + k);
          const double2 a = p[0], b = p[1];
          const double dk = dist2(qx, qy, qz, a.x, a.y, b.x);
          const int ok = (int)__double_as_longlong(b.y);
          if (dk < d || (dk == d && ok < oi)) { ls = fmin(ls, d); d = dk; oi = ok; }
          else ls = fmin(ls, dk);
        }
        n_cand += (unsigned)(hi - lo);
        // octet arg-min with the (d2, index) total order
        double D = d; int OI = oi;
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {
          const double od = __shfl_xor(D, m, 64);
          const int oo = __shfl_xor(OI, m, 64);
          if (od < D || (od == D && oo < OI)) { D = od; OI = oo; }
        }
        // smallest d2 in the leaf among the points that are NOT the leaf winner
        double other = (oi == OI) ? ls : d;
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) other = fmin(other, __shfl_xor(other, m, 64));
        if (OI == bi) {                      // the running best itself was rescanned (seed from phase 1)
          second = fmin(second, other);
        } else if (D < best || (D == best && OI < bi)) {
          if (bi != 0x7fffffff) second = fmin(second, best);
          second = fmin(second, other);
          best = D; bi = OI;
        } else {
          second = fmin(second, D);
        }
        continue;
      }
      // internal node: one child box per lane (32 B each, 256 B contiguous per octet)
      const float4* bx = reinterpret_cast<const float4*>(g.oct + 8 * ((size_t)8 * id + 1 + l));
      const float4 a = bx[0], b = bx[1];
      const double lb = oct_box_lb(qx, qy, qz, a, b);
      n_nodes += 8;
      const bool pass = lb <= best;
      if (!pass) pruned = fmin(pruned, lb);
      // rank among the passing children by (lb, lane): nearest gets rank 0
      int rank = 0, npass = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const double lj = __shfl(lb, obase + j, 64);
        const bool pj = lj <= best;
        npass += pj ? 1 : 0;
        rank += (pj && (lj < lb || (lj == lb && j < l))) ? 1 : 0;
      }
      if (pass) {
        const int pos = sp + (npass - 1 - rank);  // nearest on top
        s_id[oct][pos] = 8 * id + 1 + l;
        s_lb[oct][pos] = lb;
      }
      sp += npass;
    }
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) pruned = fmin(pruned, __shfl_xor(pruned, m, 64));
    if (l == 0) {
      job.out_idx[out] = bi == 0x7fffffff ? -1 : (job.inv ? job.inv[bi] : bi);
      job.out_d2[out] = best;
      // every other target was scanned (>= second) or sits in a skipped box (>= its lower bound)
      // (-1: no target within the search radius — re-usable as it is while the query does not move at all, see the prologue of phase 1)
      if (job.out_lb != nullptr) job.out_lb[out] = bi == 0x7fffffff ? -1.f : __double2float_rd(sqrt(fmin(second, pruned)) * (1.0 - 1e-12));
      if (job.dirty) update_list(job, i, bi == 0x7fffffff ? -1 : job.inv[bi], best, bound);
      if (bi != 0x7fffffff && second == best) tie_report(job.tie, (unsigned int)i);
    }
  }
  if (stats) {
    // every lane of an octet counted the same work: keep one lane per octet
    unsigned long long c = l == 0 ? n_cand : 0, nd = l == 0 ? n_nodes : 0;
    c = __reduce_add_u64(c); nd = __reduce_add_u64(nd);
    const size_t slot = ((size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6)) % stats_slots;
    if (__lane0()) { atomicAdd(&stats[8 * slot], c); atomicAdd(&stats[8 * slot + 1], nd); }
  }
}

// sums the per-wave census slots (8 counters each) into out8 (zeroed by the caller); 64 workgroups, 8 atomics each
__global__ __launch_bounds__(256) void census_sum_kernel(const unsigned long long* __restrict__ stats, size_t slots, unsigned long long* __restrict__ out8) {
  __shared__ unsigned long long sh[8][256];
  unsigned long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < slots; i += (size_t)gridDim.x * 256)
    for (int k = 0; k < 8; ++k) v[k] += stats[8 * i + k];
  for (int k = 0; k < 8; ++k) sh[k][threadIdx.x] = v[k];
  __syncthreads();
  if (threadIdx.x < 8) {
    unsigned long long s = 0;
    for (int i = 0; i < 256; ++i) s += sh[threadIdx.x][i];
    atomicAdd(&out8[threadIdx.x], s);
  }
}

// ---------------------------------------------------------------------------------------- host build
inline unsigned long long morton3(unsigned int x, unsigned int y, unsigned int z) {
  auto spread = [](unsigned long long v) {
    v &= 0x1fffffull;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
  };
  return spread(x) | (spread(y) << 1) | (spread(z) << 2);
}

// 3-D Hilbert index of a cell (Skilling's transpose algorithm, `bits` per axis).  Consecutive runs of a Hilbert-sorted
// surface are compact patches without the long jumps of the Z-order curve at power-of-two boundaries, so the boxes of the
// 32-point tiles / 8-ary tree nodes built over the sorted array are tighter and fewer of them overlap a query patch.
inline unsigned long long hilbert3(unsigned int x, unsigned int y, unsigned int z, int bits) {
  unsigned int X[3] = {x, y, z};
  const unsigned int M = 1u << (bits - 1);
  for (unsigned int Q = M; Q > 1; Q >>= 1) {
    const unsigned int P = Q - 1;
    for (int i = 0; i < 3; ++i) {
      if (X[i] & Q) X[0] ^= P;
      else { const unsigned int t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
    }
  }
  for (int i = 1; i < 3; ++i) X[i] ^= X[i - 1];
  unsigned int t = 0;
  for (unsigned int Q = M; Q > 1; Q >>= 1) if (X[2] & Q) t ^= Q - 1;
  for (int i = 0; i < 3; ++i) X[i] ^= t;
  return morton3(X[2], X[1], X[0]);   // interleave, X[0] most significant in every bit triple
}

struct HostGrid {
  double o[3], h, inv_h;
  int d[3];
};

inline void cell_of(const HostGrid& g, const double* p, int* c) {
  for (int a = 0; a < 3; ++a) {
    int v = (int)std::floor((p[a] - g.o[a]) * g.inv_h);
    c[a] = std::min(std::max(v, 0), g.d[a] - 1);
  }
}

size_t occupied_cells(const HostGrid& g, const double* xyz, int n, int stride) {
  std::unordered_set<unsigned long long> s;
  s.reserve((size_t)n / stride + 16);
  int c[3];
  for (int i = 0; i < n; i += stride) { cell_of(g, xyz + 3 * (size_t)i, c); s.insert(cell_key(c[0], c[1], c[2])); }
  return s.size();
}

void make_grid(HostGrid& g, const double* lo, const double* hi, double h) {
  g.h = h; g.inv_h = 1.0 / h;
  for (int a = 0; a < 3; ++a) {
    g.o[a] = lo[a] - 0.01 * h;
    g.d[a] = std::max(1, (int)std::ceil((hi[a] - g.o[a]) * g.inv_h + 0.01) + 1);
    g.d[a] = std::min(g.d[a], (1 << 21) - 1);
  }
}


// Balanced k-d ORDER of the cloud (grid_curve 2): recursive split at a multiple of 32 points (the tile size of nn_tile.hip) along the
// widest axis of the range's bounding box, left part = the largest power-of-two number of tiles below the range's tile count, so
// every aligned run of 32 * 2^k points is one subtree.  Tiles of 32 consecutive points then have ~1.5x smaller boxes than runs of
// a space-filling curve over the hash cells, and a wave of 64 consecutive queries a tighter patch: ~40 % fewer tiles opened per
// wave in the moving rounds (profiles/r02_kd_order_sim.txt).  Ties on the split coordinate break on the original index and the
// recursion runs down to single points, so the order is a pure function of the cloud.
struct KdItem { double x, y, z; int idx; int pad; };
void kd_split(KdItem* a, long long lo, long long hi, int par) noexcept {   // par: levels that still fork a host thread
  for (;;) {
    const long long m = hi - lo;
    if (m <= 1) return;
    double bl[3] = {a[lo].x, a[lo].y, a[lo].z}, bh[3] = {a[lo].x, a[lo].y, a[lo].z};
    for (long long i = lo + 1; i < hi; ++i) {
      bl[0] = std::min(bl[0], a[i].x); bh[0] = std::max(bh[0], a[i].x);
      bl[1] = std::min(bl[1], a[i].y); bh[1] = std::max(bh[1], a[i].y);
      bl[2] = std::min(bl[2], a[i].z); bh[2] = std::max(bh[2], a[i].z);
    }
    int ax = 0;
    if (bh[1] - bl[1] > bh[ax] - bl[ax]) ax = 1;
    if (bh[2] - bl[2] > bh[ax] - bl[ax]) ax = 2;
    // above a tile: whole tiles on the left; inside a tile: keep halving down to single points, so that ANY short run of
    // consecutive points is compact (the leaves of the 8-ary box tree are runs of ~6) and the order is unique
    const long long units = m > 32 ? (m + 31) / 32 : m, unit = m > 32 ? 32 : 1;
    long long left = 1;
    while (left * 2 < units) left *= 2;
    const long long mid = lo + left * unit;
    auto less = [ax](const KdItem& p, const KdItem& q) {
      const double u = ax == 0 ? p.x : ax == 1 ? p.y : p.z, v = ax == 0 ? q.x : ax == 1 ? q.y : q.z;
      return u != v ? u < v : p.idx < q.idx;
    };
    std::nth_element(a + lo, a + mid, a + hi, less);
    if (par > 0 && m > (1 << 15)) {
      // kd_split itself never throws (nth_element on PODs with a noexcept comparator; every thread construction is guarded here), so
      // neither the child's entry point nor the parent can unwind past a joinable thread; the guard joins on every path regardless.
      struct Joiner { std::thread t; ~Joiner() { if (t.joinable()) t.join(); } } j;
      try { j.t = std::thread(kd_split, a, lo, mid, par - 1); } catch (...) {}   // no thread to be had (system_error, bad_alloc, ...): do both halves here
      if (j.t.joinable()) {
        kd_split(a, mid, hi, par - 1);
        return;   // ~Joiner joins
      }
    }
    // recurse into the smaller part, loop on the larger (bounded stack)
    if (mid - lo < hi - mid) { kd_split(a, lo, mid, 0); lo = mid; }
    else { kd_split(a, mid, hi, 0); hi = mid; }
  }
}
std::atomic<int> g_builds_running{0};   // builds of several clouds run side by side (api.cpp, mvicp_set_frame): the fewer threads each forks
void kd_order(const double* xyz, int n, std::vector<int>& order) {
  std::vector<KdItem> a((size_t)n);
  for (int i = 0; i < n; ++i) a[i] = KdItem{xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], i, 0};
  const int busy = g_builds_running.load();
  kd_split(a.data(), 0, n, busy > 4 ? 0 : busy > 1 ? 1 : 3);   // (the order does not depend on how the work was split)
  order.resize(n);
  for (int i = 0; i < n; ++i) order[i] = a[i].idx;
}

}  // namespace

int build_grid(mvicp_ctx* c, FrameDev& f, const double* xyz) {
  struct Running { Running() { ++g_builds_running; } ~Running() { --g_builds_running; } } running;
  const int n = f.n;
  GridDev& G = f.grid;
  double lo[3] = {xyz[0], xyz[1], xyz[2]}, hi[3] = {xyz[0], xyz[1], xyz[2]};
  for (int i = 1; i < n; ++i)
    for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], xyz[3 * (size_t)i + a]); hi[a] = std::max(hi[a], xyz[3 * (size_t)i + a]); }
  for (int i = 0; i < 3 * n; ++i)
    if (!std::isfinite(xyz[i])) { set_error("non-finite coordinate in cloud"); return MVICP_ERR_ARG; }
  double ext = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
  if (!(ext > 0.0)) ext = 1.0;
  // cell edge: aim at ~6 points per occupied cell.  Measure occupancy at two resolutions (the cloud is a surface,
  // so occupied(h) ~ h^-dim with dim ~ 2), extrapolate, verify once.
  const double target = c->grid_target;
  double h = ext / std::max(2.0, std::cbrt((double)n));
  HostGrid g;
  if (n > 64) {
    make_grid(g, lo, hi, h);
    const double occ1 = (double)occupied_cells(g, xyz, n, 1);
    make_grid(g, lo, hi, 2.0 * h);
    const double occ2 = (double)occupied_cells(g, xyz, n, 1);
    double dim = std::log(std::max(occ1, 1.0) / std::max(occ2, 1.0)) / std::log(2.0);
    dim = std::min(3.0, std::max(1.0, dim));
    const double want = (double)n / target;
    h = h * std::pow(std::max(occ1, 1.0) / want, 1.0 / dim);
    h = std::min(std::max(h, ext * 1e-6), ext);
    for (int it = 0; it < 3; ++it) {
      make_grid(g, lo, hi, h);
      const double per = (double)n / (double)occupied_cells(g, xyz, n, 1);
      if (per < 0.6 * target) h *= std::pow(target / per, 1.0 / dim);
      else if (per > 1.8 * target) h *= std::pow(target / per, 1.0 / dim);
      else break;
    }
  }
  make_grid(g, lo, hi, h);

  // sort by (space-filling-curve index of the cell, original index)
  std::vector<unsigned long long> mkey(n), ckey(n);
  int cc[3];
  int hbits = 1;
  while ((1 << hbits) < std::max(g.d[0], std::max(g.d[1], g.d[2]))) ++hbits;
  for (int i = 0; i < n; ++i) {
    cell_of(g, xyz + 3 * (size_t)i, cc);
    mkey[i] = c->grid_curve == 0 ? morton3((unsigned)cc[0], (unsigned)cc[1], (unsigned)cc[2])
                                 : hilbert3((unsigned)cc[0], (unsigned)cc[1], (unsigned)cc[2], hbits);
    ckey[i] = cell_key(cc[0], cc[1], cc[2]);
  }
  // CELL order: the hash table's {start, count} runs index the records in this order (G.crec)
  std::vector<int> corder(n);
  std::iota(corder.begin(), corder.end(), 0);
  std::sort(corder.begin(), corder.end(), [&](int a, int b) { return mkey[a] != mkey[b] ? mkey[a] < mkey[b] : a < b; });
  // CANONICAL order ("sorted positions": what every index of the pipeline means, and what the tile hierarchy and the box tree are
  // built over): the cell order itself for the two curves, the k-d order otherwise
  const bool split_orders = c->grid_curve >= 2 && n > 64;
  std::vector<int> order;
  if (split_orders) kd_order(xyz, n, order); else order = corder;
  std::vector<double> spts(3 * (size_t)n);
  for (int i = 0; i < n; ++i) std::memcpy(&spts[3 * (size_t)i], xyz + 3 * (size_t)order[i], 24);

  // hash table of cell runs
  std::vector<HashEntry> runs;
  for (int i = 0; i < n;) {
    int j = i + 1;
    while (j < n && ckey[corder[j]] == ckey[corder[i]]) ++j;
    runs.push_back(HashEntry{ckey[corder[i]], (unsigned)i, (unsigned)(j - i)});
    i = j;
  }
  int log2size = 4;
  while ((1ull << log2size) < 2 * runs.size() + 2) ++log2size;
  const unsigned int tsize = 1u << log2size, mask = tsize - 1;
  const int shift = 64 - log2size;
  std::vector<HashEntry> table(tsize, HashEntry{EMPTY, 0u, 0u});
  for (const HashEntry& r : runs) {
    unsigned int s = hash_slot(r.key, shift);
    while (table[s].key != EMPTY) s = (s + 1) & mask;
    table[s] = r;
  }

  // dense brick map (nn_cell_kernel): 4x4x4-cell bricks -> occupancy mask + row of the cell table; cell table = {start, count} of
  // every cell run.  Only built when the dense array stays small (a surface scan at a few points per cell: ~1e5..1e6 bricks).
  {
    const long long bd[3] = {(g.d[0] + 3) / 4, (g.d[1] + 3) / 4, (g.d[2] + 3) / 4};
    const long long nb = bd[0] * bd[1] * bd[2];
    const bool capped = g.d[0] >= (1 << 21) - 1 || g.d[1] >= (1 << 21) - 1 || g.d[2] >= (1 << 21) - 1;
    if (!split_orders && !capped && nb <= (1ll << 24) && n < (1 << 30)) {   // nn_cell_kernel stages runs of the CANONICAL arrays
      std::vector<BrickEntry> bricks((size_t)nb, BrickEntry{0ull, 0xffffffffu, 0u});
      std::vector<uint2> celltab;
      unsigned int n_tab = 0;
      for (const HashEntry& r : runs) {
        const int ix = (int)(r.key & 0x1fffffull), iy = (int)((r.key >> 21) & 0x1fffffull), iz = (int)((r.key >> 42) & 0x1fffffull);
        BrickEntry& be = bricks[(size_t)(((long long)(iz >> 2) * bd[1] + (iy >> 2)) * bd[0] + (ix >> 2))];
        if (be.tab == 0xffffffffu) { be.tab = n_tab++; celltab.resize((size_t)n_tab * 64, make_uint2(0u, 0u)); }
        const int bit = (ix & 3) | ((iy & 3) << 2) | ((iz & 3) << 4);
        be.mask |= 1ull << bit;
        celltab[(size_t)be.tab * 64 + bit] = make_uint2(r.start, r.count);
      }
      MV_HIP(hipMalloc((void**)&G.bricks, sizeof(BrickEntry) * (size_t)nb));
      MV_HIP(hipMemcpy(G.bricks, bricks.data(), sizeof(BrickEntry) * (size_t)nb, hipMemcpyHostToDevice));
      MV_HIP(hipMalloc((void**)&G.celltab, sizeof(uint2) * std::max<size_t>(celltab.size(), 1)));
      MV_HIP(hipMemcpy(G.celltab, celltab.data(), sizeof(uint2) * celltab.size(), hipMemcpyHostToDevice));
      G.bdims[0] = (int)bd[0]; G.bdims[1] = (int)bd[1]; G.bdims[2] = (int)bd[2];
    }
  }

  const float finf = std::numeric_limits<float>::infinity();
  auto down = [](double v) { float f = (float)v; if ((double)f > v) f = std::nextafterf(f, -std::numeric_limits<float>::infinity()); return f; };
  auto up = [](double v) { float f = (float)v; if ((double)f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity()); return f; };
  // implicit complete 8-ary box tree over the sorted array (phase 2 of the grid kernel): 32-B boxes {lo.xyz, hi.xyz, pad}
  // leaves are ALIGNED runs of 8, 16 or 32 sorted points (whole k-d subtrees in the default order, so are all their ancestors' runs);
  // of the (depth, leaf size) pairs that cover n the one with the fewest empty leaves is taken; empty leaves keep inverted boxes
  // (lower bound +inf: never opened)
  int D8 = 0, L8 = 8;
  {
    long long best_cap = -1;
    for (int d = 0; d <= 7; ++d)      // OCT_STACK holds 7 entries per level + 1
      for (int l8 = 8; l8 <= 32; l8 *= 2) {
        const long long cap = (1ll << (3 * d)) * l8;
        if (cap >= std::max(n, 1) && (best_cap < 0 || cap < best_cap)) { best_cap = cap; D8 = d; L8 = l8; }
      }
    if (best_cap < 0) { D8 = 7; L8 = 64; while ((1ll << 21) * L8 < n) L8 *= 2; }   // > 67 M points: longer leaves
  }
  const long long leaves8 = 1ll << (3 * D8);
  const long long first_leaf8 = (leaves8 - 1) / 7;
  const long long nodes8 = first_leaf8 + leaves8;
  std::vector<float> oct(8 * (size_t)nodes8, 0.f);
  for (long long j = 0; j < leaves8; ++j) {
    const int a = (int)std::min<long long>(j * L8, n), b = (int)std::min<long long>((long long)a + L8, n);
    float* bx = &oct[8 * (size_t)(first_leaf8 + j)];
    bx[0] = bx[1] = bx[2] = finf; bx[3] = bx[4] = bx[5] = -finf;
    for (int k = a; k < b; ++k)
      for (int ax = 0; ax < 3; ++ax) { bx[ax] = std::min(bx[ax], down(spts[3 * (size_t)k + ax])); bx[3 + ax] = std::max(bx[3 + ax], up(spts[3 * (size_t)k + ax])); }
  }
  for (long long id = first_leaf8 - 1; id >= 0; --id) {
    float* bx = &oct[8 * (size_t)id];
    bx[0] = bx[1] = bx[2] = finf; bx[3] = bx[4] = bx[5] = -finf;
    for (int ch = 1; ch <= 8; ++ch) {
      const float* cb = &oct[8 * (size_t)(8 * id + ch)];
      for (int ax = 0; ax < 3; ++ax) { bx[ax] = std::min(bx[ax], cb[ax]); bx[3 + ax] = std::max(bx[3 + ax], cb[3 + ax]); }
    }
  }
  G.oct_leaf = L8; G.oct_first_leaf = first_leaf8;
  MV_HIP(hipMalloc((void**)&G.oct, sizeof(float) * oct.size()));
  MV_HIP(hipMemcpy(G.oct, oct.data(), sizeof(float) * oct.size(), hipMemcpyHostToDevice));

  // upload
  G.dims[0] = g.d[0]; G.dims[1] = g.d[1]; G.dims[2] = g.d[2];
  G.origin[0] = g.o[0]; G.origin[1] = g.o[1]; G.origin[2] = g.o[2];
  G.cell = g.h; G.inv_cell = g.inv_h;
  G.n_cells = (int)runs.size();
  G.table_mask = mask; G.table_shift = shift;
  {
    std::vector<PointRec> rec(n);
    for (int i = 0; i < n; ++i) { rec[i].x = spts[3 * (size_t)i]; rec[i].y = spts[3 * (size_t)i + 1]; rec[i].z = spts[3 * (size_t)i + 2]; rec[i].idx = order[i]; }
    MV_HIP(hipMalloc((void**)&G.srec, sizeof(PointRec) * (size_t)std::max(n, 1)));
    MV_HIP(hipMemcpy(G.srec, rec.data(), sizeof(PointRec) * (size_t)n, hipMemcpyHostToDevice));
    G.crec = G.srec;
    if (split_orders) {
      for (int i = 0; i < n; ++i) { const double* p = xyz + 3 * (size_t)corder[i]; rec[i].x = p[0]; rec[i].y = p[1]; rec[i].z = p[2]; rec[i].idx = corder[i]; }
      MV_HIP(hipMalloc((void**)&G.crec, sizeof(PointRec) * (size_t)n));
      MV_HIP(hipMemcpy(G.crec, rec.data(), sizeof(PointRec) * (size_t)n, hipMemcpyHostToDevice));
    }
  }
  MV_HIP(hipMalloc((void**)&G.spts, sizeof(double) * 3 * (size_t)n));
  MV_HIP(hipMalloc((void**)&G.sidx, sizeof(int) * (size_t)n));
  MV_HIP(hipMalloc((void**)&G.table, sizeof(HashEntry) * (size_t)tsize));
  MV_HIP(hipMemcpy(G.spts, spts.data(), sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice));
  MV_HIP(hipMemcpy(G.sidx, order.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
  G.h_order = order;
  G.h_inv.assign(n, 0);
  for (int i = 0; i < n; ++i) G.h_inv[order[i]] = i;
  MV_HIP(hipMalloc((void**)&G.inv, sizeof(int) * (size_t)std::max(n, 1)));
  MV_HIP(hipMemcpy(G.inv, G.h_inv.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
  MV_HIP(hipMemcpy(G.table, table.data(), sizeof(HashEntry) * (size_t)tsize, hipMemcpyHostToDevice));
  G.struct_bytes = sizeof(HashEntry) * (double)tsize + sizeof(float) * 8.0 * nodes8;
  MV_CHECK(build_wide(f, spts.data()));
  MV_CHECK(build_mfma(f, spts.data()));
  f.has_grid = true;
  return MVICP_OK;
}

void free_grid(GridDev& g) {
  if (g.spts) (void)hipFree(g.spts);
  if (g.sidx) (void)hipFree(g.sidx);
  if (g.crec && g.crec != g.srec) (void)hipFree(g.crec);
  if (g.srec) (void)hipFree(g.srec);
  if (g.snor) (void)hipFree(g.snor);
  if (g.inv) (void)hipFree(g.inv);
  if (g.table) (void)hipFree(g.table);
  if (g.wide) (void)hipFree(g.wide);
  if (g.mf_ops) (void)hipFree(g.mf_ops);
  if (g.mf_blk) (void)hipFree(g.mf_blk);
  if (g.oct) (void)hipFree(g.oct);
  if (g.bricks) (void)hipFree(g.bricks);
  if (g.celltab) (void)hipFree(g.celltab);
  g = GridDev();
}

namespace {
GridView view_of(const FrameDev& f) {
  GridView v;
  const GridDev& g = f.grid;
  v.spts = g.spts; v.sidx = g.sidx; v.srec = (const PointRec*)g.srec; v.crec = (const PointRec*)g.crec; v.n = f.n;
  v.table = (const HashEntry*)g.table; v.mask = g.table_mask; v.shift = g.table_shift;
  v.ox = g.origin[0]; v.oy = g.origin[1]; v.oz = g.origin[2]; v.h = g.cell; v.inv_h = g.inv_cell;
  v.dx = g.dims[0]; v.dy = g.dims[1]; v.dz = g.dims[2];
  v.oct = g.oct; v.oct_leaf = g.oct_leaf; v.oct_first_leaf = g.oct_first_leaf;
  v.bricks = (const BrickEntry*)g.bricks; v.celltab = (const uint2*)g.celltab; v.bx = g.bdims[0]; v.by = g.bdims[1]; v.bz = g.bdims[2];
  return v;
}

int run(mvicp_ctx* c, std::vector<GridJob>& jobs, double bound, const FrameDev* const* dst_of) {
  if (jobs.empty()) return MVICP_OK;
  const double search = search_bound(c, bound);
  std::vector<TieJob> ties;
  if (!(c->tie_skip && jobs[0].dirty_slots != nullptr)) {
    double launch_q = 0;
    for (const GridJob& j : jobs) launch_q += j.n;
    const TieRef tref = tie_ref(c, (size_t)launch_q, 0u);
    for (size_t k = 0; k < jobs.size(); ++k) {
      GridJob& j = jobs[k];
      j.tie = tref; j.tie.job = (unsigned int)k;
      TieJob t;
      std::memset(&t, 0, sizeof(t));
      tie_job_fill(*dst_of[k], t);
      t.q = j.q; t.xf = j.xf; t.n = j.n; t.out_idx = j.out_idx; t.out_d2 = j.out_d2; t.inv = j.inv;
      t.list = ListRef{j.qpos, j.second, j.cd2, j.dirty, j.dirty_slots, j.stream, j.total_cap, j.dst_nor, j.dst.srec};
      ties.push_back(t);
    }
  }
  int max_n = 0;
  double nq = 0;
  for (const GridJob& j : jobs) { max_n = std::max(max_n, j.n); nq += j.n; }
  if (max_n == 0) return MVICP_OK;
  GridJob* d_jobs = nullptr;
  MV_CHECK(cached_upload(c, jobs[0].xf ? "grid_jobs" : "grid_jobs_raw", jobs.data(), sizeof(GridJob) * jobs.size(), (void**)&d_jobs));
  unsigned long long* d_stats = nullptr;
  const size_t slots = (size_t)((max_n + NT - 1) / NT) * jobs.size() * (NT / 64);
  if (c->profile && c->nn_census) {
    const size_t need = sizeof(unsigned long long) * 8 * (slots + 1);
    if (need > c->census_bytes) {
      if (c->d_census) MV_HIP(hipFree(c->d_census));
      MV_HIP(hipMalloc((void**)&c->d_census, need));
      c->census_bytes = need;
    }
    d_stats = (unsigned long long*)c->d_census;
    MV_HIP(hipMemsetAsync(d_stats, 0, need, c->stream));
  }
  // far list: (job, query) pairs, worst case every query
  const size_t total_q = (size_t)nq;
  if (total_q > c->far_cap) {
    if (c->d_far_list) MV_HIP(hipFree(c->d_far_list));
    MV_HIP(hipMalloc((void**)&c->d_far_list, sizeof(int2) * total_q));
    c->far_cap = total_q;
  }
  const bool edge_path = jobs[0].dirty_slots != nullptr;   // dirty_reduce_kernel re-zeroes the counter after phase 2
  if (!c->d_far_count) {
    MV_HIP(hipMalloc((void**)&c->d_far_count, 2 * sizeof(unsigned int)));
    MV_HIP(hipMemsetAsync(c->d_far_count, 0, 2 * sizeof(unsigned int), c->stream));
    c->far_parity = 0;
  }
  unsigned int* far_cnt = c->d_far_count + c->far_parity;         // this launch's counter (zero: left so by the previous launch)
  unsigned int* far_next = c->d_far_count + (c->far_parity ^ 1);
  c->far_parity ^= 1;
  bool use_cell = false;
  const dim3 grid((max_n + NT - 1) / NT, (unsigned)jobs.size());
  {
    ProfScope ps(c, "nn_grid", 36.0 * nq);  // the phase-1 kernel alone: query 24 B + result 12 B; the rest comes from the census below (0 if the census is off)
    // edge searches on clouds with a brick map: the wave-cooperative cell-staging kernel; otherwise (raw queries, profiling
    // switches, grids too large for a dense brick map) the per-lane hash kernel
    use_cell = c->nn_cell && !c->nn_tree_only && !c->nn_skip_far && edge_path;
    for (const GridJob& j : jobs) if (j.dst.bricks == nullptr || j.xf == nullptr) use_cell = false;
    if (c->nn_tree_only)
      hipLaunchKernelGGL((nn_grid_kernel<true>), grid, dim3(NT), 0, c->stream, d_jobs, bound, search, d_stats, 0, (int2*)c->d_far_list, far_cnt, 0.0);
    else if (use_cell)
      hipLaunchKernelGGL(nn_cell_kernel, grid, dim3(NT), 0, c->stream, d_jobs, bound, search, d_stats, (int2*)c->d_far_list, far_cnt, c->prune_rho);
    else
      hipLaunchKernelGGL((nn_grid_kernel<false>), grid, dim3(NT), 0, c->stream, d_jobs, bound, search, d_stats, c->nn_skip_far ? 1 : 0, (int2*)c->d_far_list, far_cnt,
                         c->prune_rho);
  }
  const bool launch_far = !(edge_path && c->far_skip), launch_dirty = edge_path && !c->skip_dirty_reduce;
  if (launch_far || launch_dirty) {
    ProfScope ps(c, "nn_far", 0.0);   // phase 2 + the list flags: their own scope, so that "nn_grid" times one kernel ("nn" = every nn_* scope together)
    // phase 2: persistent grid-stride launch (the far count is only known on the device)
    // With the temporal cache on, at most a fraction of a per cent of the queries ever reach the far list (0.1-0.2 % in the hand-over rounds,
    // none at the fixed point): a 2048-workgroup launch then costs 19-22 us to find an empty list — 128 workgroups walk the same list
    // (grid-stride) and cost ~4 us.  Only at the fixed point (api.cpp sets far_narrow: every transform bit-identical to last search's): with
    // 0.2 % far queries (the 96 %-hit round of a hand-over) the narrow launch measured 0.1 ms SLOWER than the wide one.
    const size_t far_wide = (edge_path && c->far_narrow) ? 128 : 256 * 8;
    const unsigned int far_blocks = (unsigned int)std::min<size_t>(far_wide, (total_q * 8 + NT - 1) / NT);
    if (!c->h_far_seen) {
      MV_HIP(hipHostMalloc((void**)&c->h_far_seen, sizeof(unsigned int), hipHostMallocMapped));
      *c->h_far_seen = 1u;   // unknown until a launch has written it
      MV_HIP(hipHostGetDevicePointer((void**)&c->d_far_seen, c->h_far_seen, 0));
    }
    // (at the fixed point — every query bit-identical to last round's — a round that follows one without far queries has none either:
    // its counter stays zero and phase 2 is not launched at all, api.cpp far_skip)
    if (launch_far)
      hipLaunchKernelGGL(nn_far_kernel, dim3(far_blocks), dim3(NT), 0, c->stream, d_jobs, (const int2*)c->d_far_list, bound, (const unsigned int*)far_cnt, far_next, d_stats, slots,
                         edge_path ? c->d_far_seen : nullptr);
    // per-edge OR of the "list changed" slots — not needed when the host already knows that nothing can change (api.cpp: every
    // transform bit-identical, every list valid): no query marks a slot then
    if (launch_dirty)
      hipLaunchKernelGGL(dirty_reduce_kernel, dim3(c->E), dim3(256), 0, c->stream, c->E, c->d_dslot_off, c->d_dirty_slots, c->d_dirty);
  }
  MV_HIP(hipGetLastError());
  if (d_stats) {
    if (!c->h_census) MV_HIP(hipHostMalloc((void**)&c->h_census, 8 * sizeof(unsigned long long), hipHostMallocDefault));
    // counters -> pinned memory, asynchronously; census_resolve() folds them in after the caller's own wait (no extra sync)
    hipLaunchKernelGGL(census_sum_kernel, dim3(64), dim3(256), 0, c->stream, d_stats, slots, d_stats + 8 * slots);
    MV_HIP(hipMemcpyAsync(c->h_census, d_stats + 8 * slots, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    c->census_pending = true; c->census_nq = nq; c->census_kind = c->nn_tree_only ? 1 : (use_cell ? 3 : 0); c->census_scope = "nn_grid";
  }
  MV_CHECK(launch_tie_fixup(c, ties, bound));   // exact distance ties: the reference's own descent decides (nn_tie.hip)
  return MVICP_OK;
}
}  // namespace

int launch_dirty_reduce(mvicp_ctx* c) {
  hipLaunchKernelGGL(dirty_reduce_kernel, dim3(c->E), dim3(256), 0, c->stream, c->E, c->d_dslot_off, c->d_dirty_slots, c->d_dirty);
  MV_HIP(hipGetLastError());
  return MVICP_OK;
}

namespace { __global__ void grid_warm_kernel() {} }

// Called once per graph (mvicp_set_graph): the runtime loads a translation unit's code object at the first launch of any of its
// kernels, and for this file that first launch used to sit in the middle of an ICP loop (the round AUTO hands over to the grid
// kernel: ~0.5 ms on top of that round).  An empty launch pays it at set-up time instead; the far-list counters are created here too.
int warm_nn_grid(mvicp_ctx* c) {
  if (!c->d_far_count) {
    MV_HIP(hipMalloc((void**)&c->d_far_count, 2 * sizeof(unsigned int)));
    MV_HIP(hipMemsetAsync(c->d_far_count, 0, 2 * sizeof(unsigned int), c->stream));
    c->far_parity = 0;
  }
  hipLaunchKernelGGL(grid_warm_kernel, dim3(1), dim3(64), 0, c->stream);
  MV_HIP(hipGetLastError());
  return MVICP_OK;
}

int launch_nn_grid_edges(mvicp_ctx* c, double d2_bound) {
  std::vector<GridJob> jobs;
  std::vector<const FrameDev*> dsts;
  for (int e = 0; e < c->E; ++e) {
    if (!c->active[e]) continue;
    const FrameDev& s = c->frames[c->esrc[e]];
    const FrameDev& d = c->frames[c->edst[e]];
    if (!s.has_grid || !d.has_grid) { set_error("grid NN needs the per-cloud structure on frames %d and %d", c->esrc[e], c->edst[e]); return MVICP_ERR_STATE; }
    GridJob j;
    std::memset(&j, 0, sizeof(j));  // padding too: the table is cached by content
    j.dst = view_of(d);
    j.q = s.grid.spts; j.qidx = nullptr; j.xf = c->d_xf + (size_t)e * kEdgeXf; j.n = s.n;
    j.out_idx = c->d_nn_idx + c->cap_off[e]; j.out_d2 = c->d_nn_d2 + c->cap_off[e];
    j.inv = d.grid.inv; j.out_lb = c->d_nn_lb + c->cap_off[e];
    j.qpos = c->d_qpos + c->cap_off[e]; j.second = c->d_second + c->cap_off[e]; j.cd2 = c->d_cd2 + c->cap_off[e]; j.dirty = c->d_dirty + e;
    j.stream = c->d_stream + c->cap_off[e]; j.total_cap = c->total_cap; j.dst_nor = d.grid.snor;
    j.dirty_slots = c->d_dirty_slots + c->dslot_off[e];
    j.seed = ((int)c->nn_cache_edge.size() == c->E && c->nn_cache_edge[e]) ? 1 : 0;
    jobs.push_back(j); dsts.push_back(&d);
  }
  return run(c, jobs, d2_bound, dsts.data());
}

int launch_nn_grid_queries(mvicp_ctx* c, const FrameDev& f, const double* d_q, int n, int* d_idx, double* d_d2) {
  if (!f.has_grid) { set_error("grid NN structure missing"); return MVICP_ERR_STATE; }
  std::vector<GridJob> jobs(1);
  std::memset(&jobs[0], 0, sizeof(GridJob));
  jobs[0].dst = view_of(f);
  jobs[0].q = d_q; jobs[0].qidx = nullptr; jobs[0].xf = nullptr; jobs[0].n = n;
  jobs[0].out_idx = d_idx; jobs[0].out_d2 = d_d2;
  jobs[0].inv = nullptr; jobs[0].out_lb = nullptr;
  jobs[0].qpos = nullptr; jobs[0].second = nullptr; jobs[0].cd2 = nullptr; jobs[0].dirty = nullptr; jobs[0].dirty_slots = nullptr;
  jobs[0].stream = nullptr; jobs[0].total_cap = 0; jobs[0].dst_nor = nullptr;
  const FrameDev* dst = &f;
  return run(c, jobs, 1.7976931348623157e308, &dst);
}

}  // namespace mvicp
