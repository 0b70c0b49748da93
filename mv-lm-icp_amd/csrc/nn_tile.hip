// K2t — wave-cooperative exact 1-NN: pruned brute force over LDS-staged target tiles.
//
// Same contract as K1/K2 (bit-exact indices and squared distances of the reference's
// Frame::getClosestPoint, src/internal/frame.cpp:187-206, metric include/frame.h:70-76, query transform
// frame.cpp:117-118,131,136; on an exact tie the lowest original index keeps the place and the query is reported to nn_tie.hip, which decides it
// the way nanoflann does).
//
// Both clouds are stored in a balanced k-d order whose aligned runs of 32 * 2^k points are subtrees (nn_grid.hip, kd_order; or
// sorted by the Hilbert index of their grid cell, grid_curve 1: ~1.7x more tiles opened per wave).  A WAVE owns 64
// consecutive sorted source points — a compact surface patch — and answers all 64 queries together:
//   * the target cloud is cut into leaves of LEAF = 32 consecutive sorted points; leaf boxes, boxes of 64 leaves,
//     boxes of 64 of those ... form a 64-wide hierarchy (float AABBs rounded outward, SoA per level) so one
//     coalesced load hands every lane ONE child box of the current node;
//   * a child is opened iff at least one lane still needs it: coarse cull = child box vs the patch's own AABB
//     against the largest per-lane threshold of the wave (all children tested in parallel, one per lane), then a
//     per-lane fp32 test of the box against the lane's own threshold (ballot).  Thresholds carry a rigorous guard
//     band for the fp32 conversions (see leaf_scan), so a box or point is only ever skipped when it is provably
//     farther than the lane's running best; whatever passes is re-evaluated in the reference's fp64 arithmetic;
//   * an opened leaf is staged once into LDS (coalesced) and every lane screens its 32 points from LDS (broadcast
//     reads, conflict-free, four candidates per step in packed fp32), confirming the few that pass in fp64 and
//     keeping its running (d2, index) minimum.  Children are opened nearest-first (wave arg-min on the DPP network)
//     so the running minima tighten before the siblings are tested;
//   * from the second round on every lane starts from last round's neighbour (an ordinary candidate);
//   * BND build (template flag; api.cpp): the guard band is widened by mu and the exact second-best distance is tracked, which yields
//     the per-query lower bound the temporal cache needs (leaf_scan).  Used for the AUTO round that hands over from the plain build, and
//     (round 3, "cache-aware" rounds) for every later round whose poses still move: there the kernel first runs the temporal-cache
//     check of nn_grid_kernel as its prologue — a lane whose neighbour provably did not change is finished and sits the traversal
//     out, the wave searches for its missed lanes only, a wave without a miss leaves at once.  The plain build carries none of that;
//   * whenever the host hands the edge's compacted list over (TileJob::list), every answered lane maintains its own entry in place
//     (nn_list.h: distance refresh, neighbour + operand patch, membership change -> edge dirty), so compaction + gather only run for
//     edges whose membership changed — in the seeded plain rounds too.
// The kernel is VALU-issue bound (not memory bound): its design minimises wave instructions per opened tile.
// No per-lane pointer chasing, no divergence between "near" and "far" queries: the first-round regime
// (centimetre misalignment) and the converged regime run the same code, the former just opens more leaves.
#include "nn_tile_common.h"

namespace mvicp {

namespace {

typedef float f2v __attribute__((ext_vector_type(2)));

#ifndef MVICP_TILE_THREADS
#define MVICP_TILE_THREADS 128
#endif
constexpr int TT = MVICP_TILE_THREADS;   // threads per workgroup of nn_tile_kernel

struct Lane {          // per-lane query state
  double qx, qy, qz, best;
  // fp32 copies for the screens, kept as splatted PAIRS (the packed-math operands of leaf_scan).  As four adjacent floats
  // {qxf, qyf, qzf, thr} the vectorizer loaded them as overlapping <2 x float> pairs, which pinned that quad in scratch memory:
  // three scratch loads and one scratch store per lane per opened tile — 1.4 GB of memory writes per launch on cfg4 (round-1 PMC).
  f2v qx2, qy2, qz2;
  double pad_;           // (keeps thr away from the pairs: no cross-field vector loads)
  float thr;             // fp32 screen threshold, always >= (sqrt(best) + slack + mu)^2 (see leaf_scan; mu = 0 unless BND)
  int bi;
  bool active;
  bool tie;              // another target met at exactly the running best's distance (nn_tie.h)
  double second;         // BND builds only: smallest exact d2 among the fp64-evaluated targets other than the running best
};
struct Group {         // wave-uniform patch description
  float lo[3], hi[3], c[3];   // the patch AABB in fp32, rounded OUTWARD, and its centre (the coarse cull runs in fp32)
  float slack;         // fp32 screening guard band (metres)
  float mu;            // BND builds: extra guard band (0 otherwise)
};

// fp32 squared distance from the lane's query to a box.  Same guard-band argument as the point screen in leaf_scan: the
// nearest point of the box has box coordinates (exact floats) or the query's own, so |sqrt(lb32) - sqrt(true)| <= slack and
// every box holding a point with d2 <= best satisfies lb32 <= L.thr.
__device__ __forceinline__ float box_lb32(const Lane& L, float b0, float b1, float b2, float b3, float b4, float b5) {
  const float qxf = L.qx2.x, qyf = L.qy2.x, qzf = L.qz2.x;
  const float g0 = fmaxf(fmaxf(b0 - qxf, qxf - b3), 0.f);
  const float g1 = fmaxf(fmaxf(b1 - qyf, qyf - b4), 0.f);
  const float g2 = fmaxf(fmaxf(b2 - qzf, qzf - b5), 0.f);
  return __builtin_fmaf(g2, g2, __builtin_fmaf(g1, g1, g0 * g0));
}


// Scan one leaf tile for the whole wave.  The tile is staged once in LDS as fp64 (exact evaluation) AND fp32
// (screening): a candidate is evaluated in the reference's fp64 arithmetic only if its fp32 distance is within
// a rigorous guard band of the lane's running best — |sqrt(d32) - sqrt(d)| <= slack, where `slack` bounds the two
// float conversions (2^-24 |coord| each, per axis) plus the fp32 rounding of the sum (relative 2^-22; the fused
// multiply-adds used here round less often than the separate operations the bound was derived for).  Everything the
// screen rejects is provably farther than `best`, so the result is unchanged bit for bit.
// The screen is the hot loop of the kernel (VALU-bound): four candidates per step, SoA in LDS so that one 16-B read
// brings four x (y, z) values, packed fp32 math (v_pk_add/mul/fma: two candidates per instruction), one branch per
// four candidates; the fp64 confirmation runs only for the lanes / candidates that pass.
struct TileLds {   // per wave
  double x[LEAF], y[LEAF], z[LEAF];
  float fx[LEAF], fy[LEAF], fz[LEAF];
  int id[LEAF];
};

// BND (the round that hands over to the grid kernel): the guard band is widened by mu, so everything the screen rejects — points
// here, boxes in visit() — is farther than sqrt(best) + mu from the query at that moment, hence farther than sqrt(final best) + mu;
// everything that passes is evaluated exactly and the smallest d2 among those that are NOT the running best is kept in L.second.
// min(sqrt(second), sqrt(best) + mu) is then a lower bound on the distance to every target other than the answer: the quantity the
// grid kernel's temporal cache needs (GridJob::out_lb).
template <bool BND>
__device__ __forceinline__ void leaf_scan(const TileView& g, int leaf, Lane& L, float slack, float mu, TileLds* __restrict__ T, unsigned int* n_cand) {
  const int lane = threadIdx.x & 63;
  const int lo = leaf * LEAF;
  const int cnt = min(LEAF, g.n - lo);
  __builtin_amdgcn_wave_barrier();
  if (lane < LEAF) {
    if (lane < cnt) {
      const double* p = g.spts + 3 * (size_t)(lo + lane);
      const double x = p[0], y = p[1], z = p[2];
      T->x[lane] = x; T->y[lane] = y; T->z[lane] = z;
      T->fx[lane] = (float)x; T->fy[lane] = (float)y; T->fz[lane] = (float)z;
      T->id[lane] = g.sidx[lo + lane];
    } else {
      const float inf = __int_as_float(0x7f800000);
      T->fx[lane] = inf; T->fy[lane] = inf; T->fz[lane] = inf;   // padded slots never pass a finite screen (and are re-checked below)
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const f2v qx2 = L.qx2, qy2 = L.qy2, qz2 = L.qz2;
  // L.thr >= (sqrt(best) + slack)^2 at all times.  When a candidate with screen value d32 becomes the best, sqrt(best) <=
  // sqrt(d32) + slack, so (sqrt(d32) + 2 slack)^2 is a valid new threshold: one fp32 sqrt instead of an fp64 one per update.
  float thr = L.thr;
  const float4* X4 = reinterpret_cast<const float4*>(T->fx);
  const float4* Y4 = reinterpret_cast<const float4*>(T->fy);
  const float4* Z4 = reinterpret_cast<const float4*>(T->fz);
#pragma unroll
  for (int k4 = 0; k4 < LEAF / 4; ++k4) {
    const float4 X = X4[k4], Y = Y4[k4], Z = Z4[k4];
    const f2v xa = {X.x, X.y}, xb = {X.z, X.w}, ya = {Y.x, Y.y}, yb = {Y.z, Y.w}, za = {Z.x, Z.y}, zb = {Z.z, Z.w};
    const f2v ea = qx2 - xa, eb = qx2 - xb, fa = qy2 - ya, fb = qy2 - yb, ga = qz2 - za, gb = qz2 - zb;
    const f2v da = __builtin_elementwise_fma(ga, ga, __builtin_elementwise_fma(fa, fa, ea * ea));
    const f2v db = __builtin_elementwise_fma(gb, gb, __builtin_elementwise_fma(fb, fb, eb * eb));
    // the four screen results as lane predicates (one v_cmp each, the masks live in SGPR pairs); an integer hit mask per lane is 8 more
    // VALU instructions per group in the listing but measures the same (SQ_INSTS_VALU identical, profiles/r03_tile_ab.txt)
    const bool h[4] = {da.x <= thr, da.y <= thr, db.x <= thr, db.y <= thr};
    if (h[0] || h[1] || h[2] || h[3]) {
      const float d32[4] = {da.x, da.y, db.x, db.y};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = 4 * k4 + j;
        if (h[j] && k < cnt) {
          const double d0 = __dsub_rn(L.qx, T->x[k]), d1 = __dsub_rn(L.qy, T->y[k]), d2 = __dsub_rn(L.qz, T->z[k]);
          const double d = __dadd_rn(__dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1)), __dmul_rn(d2, d2));
          // strictly nearer: the new best.  EXACTLY as near (rare): the lower original index keeps the place and the query is reported —
          // nn_tie.hip then lets the reference's own tree decide (nanoflann keeps the target it visits first)
          if (d < L.best) {
            if (BND) L.second = fmin(L.second, L.best);   // the old best (or the cutoff bound: only lowers the bound) is now "another target"
            L.best = d; L.bi = T->id[k];
            const float r = __builtin_amdgcn_sqrtf(d32[j]) * 1.000001f + 2.f * slack + (BND ? mu : 0.f);
            thr = fminf(thr, r * r * 1.000002f);
          } else if (d == L.best) {
            const int oi = T->id[k];
            if (oi != L.bi) {                        // (the running best itself comes by again when its tile is scanned after a seed)
              if (BND) L.second = fmin(L.second, d); else L.tie = true;   // (BND builds read the tie off second == best at the end)
              if (oi < L.bi) L.bi = oi;
            }
          } else if (BND) {
            L.second = fmin(L.second, d);
          }
        }
      }
    }
  }
  L.thr = thr;
  *n_cand += (unsigned)cnt;
}

// 64-bit wave broadcast (two v_readlane)
__device__ __forceinline__ double bcast_d(double v, int lane) {
  const long long b = __double_as_longlong(v);
  const unsigned int lo = (unsigned int)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
  const unsigned int hi = (unsigned int)__builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// MISS path (round 6; cache-aware rounds only).  After the temporal-cache prologue a wave is typically left with a handful of missed lanes
// (cfg4 rounds 6 / 7: 6 % / 1 % of the queries).  The lanes-over-queries scan above then stages a whole tile in LDS and runs the packed screen
// in all 64 lanes for the sake of one or two of them — ~440 VALU instructions and 900 B per opened tile whoever needs it.  Here the roles are
// swapped: the missed queries are taken ONE BY ONE (their state broadcast into SGPRs) and the 64 lanes are spread over the DATA —
//   * block level: lane t holds the box of the block's tile t and tests it against the query's ball (one ballot = the query's tile set);
//   * tile level: two tiles per step, lane l evaluates point l & 31 of tile l >> 5 in the reference's fp64 arithmetic straight from the 32-B sorted
//     records (one coalesced 2-KB load, no LDS staging, no fp32 screen); the few lanes that come within the running best update the scalar state
//     by leaf_scan's own rules, one after the other.
// BND's second-best is kept as a LOWER bound here (the fp32 floor of the points beyond the best, clamped at the best) — all the temporal cache needs;
// an exact tie still leaves second == best, the BND builds' tie signal.  Results (index, d2) are
// those of the scan above bit for bit: the same candidates — every point of every tile whose box reaches the query's ball — by the same rules.
template <bool BND>
__device__ void miss_block(const TileView& g, int first, int nchild, Lane& L, const Group& G, unsigned int* n_cand, unsigned int* n_box) {
  const int lane = threadIdx.x & 63;
  const float inf = __int_as_float(0x7f800000);
  float b0 = inf, b1 = inf, b2 = inf, b3 = -inf, b4 = -inf, b5 = -inf;
  if (lane < nchild) {
    const float* base = g.wide + g.off[0] + first + lane;
    const long long st = g.cnt[0];
    b0 = base[0]; b1 = base[st]; b2 = base[2 * st]; b3 = base[3 * st]; b4 = base[4 * st]; b5 = base[5 * st];
  }
  *n_box += (unsigned)nchild;
  unsigned long long am = __ballot(L.active);
  while (am != 0ull) {
    const int a = __ffsll((long long)am) - 1;
    am &= am - 1ull;
    const float qxf = bcast(L.qx2.x, a), qyf = bcast(L.qy2.x, a), qzf = bcast(L.qz2.x, a);
    float thr = bcast(L.thr, a);
    const float g0 = fmaxf(fmaxf(b0 - qxf, qxf - b3), 0.f);
    const float g1 = fmaxf(fmaxf(b1 - qyf, qyf - b4), 0.f);
    const float g2 = fmaxf(fmaxf(b2 - qzf, qzf - b5), 0.f);
    const float lb = __builtin_fmaf(g2, g2, __builtin_fmaf(g1, g1, g0 * g0));   // box_lb32 of the broadcast query (same guard-band argument)
    unsigned long long tm = __ballot(lane < nchild && lb <= thr);
    if (tm == 0ull) continue;
    const double qx = bcast_d(L.qx, a), qy = bcast_d(L.qy, a), qz = bcast_d(L.qz, a);
    double best = bcast_d(L.best, a), second = bcast_d(L.second, a);
    int bi = __builtin_amdgcn_readlane(L.bi, a);
    bool tie = __builtin_amdgcn_readlane((int)L.tie, a) != 0;
    bool changed = false;
    while (tm != 0ull) {
      const int t0 = __ffsll((long long)tm) - 1;
      tm &= tm - 1ull;
      int t1 = -1;
      if (tm != 0ull) { t1 = __ffsll((long long)tm) - 1; tm &= tm - 1ull; }
      const int t = lane < 32 ? t0 : t1;
      const int k = (first + t) * LEAF + (lane & 31);
      const bool valid = t >= 0 && k < g.n;
      double2 u = make_double2(0.0, 0.0), v = make_double2(0.0, 0.0);
      if (valid) {
        const double2* pr = reinterpret_cast<const double2*>(g.srec + k);
        u = pr[0]; v = pr[1];
      }
      const double d0 = __dsub_rn(qx, u.x), d1 = __dsub_rn(qy, u.y), d2 = __dsub_rn(qz, v.x);
      const double d = __dadd_rn(__dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1)), __dmul_rn(d2, d2));
      const int oi = (int)__double_as_longlong(v.y);
      *n_cand += (unsigned)(min(LEAF, g.n - (first + t0) * LEAF) + (t1 >= 0 ? min(LEAF, g.n - (first + t1) * LEAF) : 0));
      unsigned long long pm = __ballot(valid && d <= best);
      if (BND) {
        // everything beyond the running best only lowers `second`: its fp32 floor is a valid (and nearly tight) lower bound
        // (clamped from below at the smallest double ABOVE the running best: d > best means d >= that, so it is still a lower bound of every such d,
        // and it keeps `second == best` — the BND builds' tie signal — for exact ties only: an fp32 floor that rounds down onto the best would report one
        // false tie per ~1e7 queries, i.e. every launch)
        const float far32 = wave_min_f(valid && d > best ? __double2float_rd(d) : inf);
        second = fmin(second, fmax((double)far32, __longlong_as_double(__double_as_longlong(best) + 1ll)));
      }
      while (pm != 0ull) {
        const int j = __ffsll((long long)pm) - 1;
        pm &= pm - 1ull;
        const double dj = bcast_d(d, j);
        const int oj = __builtin_amdgcn_readlane(oi, j);
        if (dj < best) {
          if (BND) second = fmin(second, best);
          best = dj; bi = oj; changed = true;
        } else if (dj == best) {
          if (oj != bi) { if (BND) second = fmin(second, dj); else tie = true; if (oj < bi) bi = oj; }
        } else if (BND) {
          second = fmin(second, dj);
        }
      }
    }
    if (changed) thr = fminf(thr, thr_of(best, G.slack + G.mu));
    if (lane == a) { L.best = best; L.bi = bi; if (!BND) L.tie = tie; L.second = second; L.thr = thr; }
  }
}

// Test the `nchild` boxes [first, first + nchild) of level LEVEL (one per lane) and open what is needed.
// Register budget: the traversal recurses (level 2 -> 1 -> 0 -> tile scan) and everything a level keeps across the descent
// is live in all deeper levels.  Levels 1 and 2 therefore park their 64 child boxes in wave-private LDS (32 B each, read
// back with one uniform-address load per step) and keep only {cull distance, order key, pending} per lane; level 0, the
// hot one, keeps its boxes in registers and broadcasts them with v_readlane.
// `miss` (wave-uniform, BND builds only): below the block level the wave runs miss_block instead of the tile loop — a run-time flag, not a template
// parameter, so that the levels above exist once (two instantiations of the whole traversal cost the regular path 3-4 %: instruction cache)
template <int LEVEL, bool BND>
__device__ void visit(const TileView& g, int first, int nchild, Lane& L, const Group& G, TileLds* __restrict__ T,
                      float2* __restrict__ sbox, unsigned int* n_cand, unsigned int* n_box, bool miss) {
  if (BND && LEVEL == 0 && miss) { miss_block<BND>(g, first, nchild, L, G, n_cand, n_box); return; }
  {
  constexpr bool IN_LDS = LEVEL == 1 || LEVEL == 2;
  const int lane = threadIdx.x & 63;
  const float inf = __int_as_float(0x7f800000);
  float b0 = inf, b1 = inf, b2 = inf, b3 = -inf, b4 = -inf, b5 = -inf;
  if (lane < nchild) {
    const float* base = g.wide + g.off[LEVEL] + first + lane;
    const long long st = g.cnt[LEVEL];
    b0 = base[0]; b1 = base[st]; b2 = base[2 * st]; b3 = base[3 * st]; b4 = base[4 * st]; b5 = base[5 * st];
  }
  // coarse cull: child box vs the patch AABB; valid for every lane because lb_lane >= box-box distance
  float ddf, key;
  {
    // fp32 against the outward-rounded patch box: every operation rounds by <= 2^-24 relative, (1 - 1e-6) more than covers the five of
    // them, so ddf stays a lower bound of the box-to-patch distance (round 3: was fp64 — 12 conversions + ~20 fp64 operations per node)
    const float e0 = fmaxf(fmaxf(b0 - G.hi[0], G.lo[0] - b3), 0.f);
    const float e1 = fmaxf(fmaxf(b1 - G.hi[1], G.lo[1] - b4), 0.f);
    const float e2 = fmaxf(fmaxf(b2 - G.hi[2], G.lo[2] - b5), 0.f);
    ddf = (e0 * e0 + e1 * e1 + e2 * e2) * 0.999999f;
    const float k0 = fmaxf(fmaxf(b0 - G.c[0], G.c[0] - b3), 0.f);
    const float k1 = fmaxf(fmaxf(b1 - G.c[1], G.c[1] - b4), 0.f);
    const float k2 = fmaxf(fmaxf(b2 - G.c[2], G.c[2] - b5), 0.f);
    key = k0 * k0 + k1 * k1 + k2 * k2;   // visiting order only
  }
  float2* mybox = sbox + (IN_LDS ? (LEVEL - 1) * 3 * FAN : 0);   // [axis][child] -> (lo, hi)
  if (IN_LDS) {
    __builtin_amdgcn_wave_barrier();
    mybox[lane] = make_float2(b0, b3);
    mybox[FAN + lane] = make_float2(b1, b4);
    mybox[2 * FAN + lane] = make_float2(b2, b5);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  *n_box += (unsigned)nchild;
  bool pend = lane < nchild;
  // largest screen threshold of the wave (>= every lane's best): only shrinks, and only when a tile was scanned below ->
  // refreshed after descents
  float gmax = wave_max_f(L.active ? L.thr : 0.f);
  while (true) {
    pend = pend && ddf <= gmax;
    const unsigned long long mask = __ballot(pend);
    if (mask == 0ull) break;
    const float kmin = wave_min_f(pend ? key : inf);
    const unsigned long long pick = __ballot(pend && key == kmin);
    const int c = __builtin_amdgcn_readfirstlane(__ffsll((long long)pick) - 1);
    if (lane == c) pend = false;
    float c0, c1, c2, c3, c4, c5;
    if (IN_LDS) {
      const float2 u = mybox[c], v = mybox[FAN + c], w = mybox[2 * FAN + c];
      c0 = u.x; c3 = u.y; c1 = v.x; c4 = v.y; c2 = w.x; c5 = w.y;
    } else {
      c0 = bcast(b0, c); c1 = bcast(b1, c); c2 = bcast(b2, c); c3 = bcast(b3, c); c4 = bcast(b4, c); c5 = bcast(b5, c);
    }
    const float lb = box_lb32(L, c0, c1, c2, c3, c4, c5);
    if (__ballot(L.active && lb <= L.thr) == 0ull) continue;
    const int child = first + c;
    if (LEVEL == 0) {
      leaf_scan<BND>(g, child, L, G.slack, G.mu, T, n_cand);
    } else {
      const int cf = child * FAN;
      visit<(LEVEL > 0 ? LEVEL - 1 : 0), BND>(g, cf, min(FAN, g.cnt[LEVEL > 0 ? LEVEL - 1 : 0] - cf), L, G, T, sbox, n_cand, n_box, miss);
    }
    gmax = wave_max_f(L.active ? L.thr : 0.f);
  }
  }
}

// TOP >= 0: every target of the launch has exactly TOP + 1 hierarchy levels (the common case: clouds of similar size), so
// only that traversal is compiled in; TOP = -1: generic (per-job switch over the depth).
template <int WPE, int TOP, bool BND>
__global__ __launch_bounds__(TT, WPE) void nn_tile_kernel(const TileJob* __restrict__ jobs, double bound, double search, unsigned long long* __restrict__ stats) {
  __shared__ TileLds s_tile[TT / 64];
  __shared__ float2 s_box[TT / 64][2 * 3 * FAN];   // levels 1 and 2: 64 child boxes x 24 B each, per wave
  __shared__ double sxf[kEdgeXf];
  const TileJob& job = jobs[blockIdx.y];
  if (blockIdx.x * TT >= job.n) return;
  const bool has_xf = job.xf != nullptr;
  if (has_xf && threadIdx.x < kEdgeXf) sxf[threadIdx.x] = job.xf[threadIdx.x];
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  const int i = blockIdx.x * TT + threadIdx.x;
  if ((i & ~63) >= job.n) return;  // whole wave beyond the end
  const TileView& g = job.dst;

  Lane L;
  L.active = i < job.n;
  L.best = search; L.bi = 0x7fffffff; L.tie = false;
  L.second = 1.7976931348623157e308;
  L.qx = L.qy = L.qz = 0.0;
  double p0 = 0.0, p1 = 0.0, p2 = 0.0;
  if (L.active) {
    p0 = job.q[3 * (size_t)i]; p1 = job.q[3 * (size_t)i + 1]; p2 = job.q[3 * (size_t)i + 2];
    if (has_xf) xf_point(sxf, p0, p1, p2, L.qx, L.qy, L.qz);
    else { L.qx = p0; L.qy = p1; L.qz = p2; }
  }
  // Seed: last round's neighbour is an ordinary candidate (any target is), but starting from its distance instead of the
  // cutoff bound lets the traversal discard almost every tile that does not hold a true neighbour of some lane.
  int seed_pi = -1;
  double seed_d = 0.0;
  if (job.seed && L.active) {
    const int pi = job.out_idx[i];
    if (pi >= 0 && pi < g.n) {
      const double* p = g.spts + 3 * (size_t)pi;
      const double d0 = __dsub_rn(L.qx, p[0]), d1 = __dsub_rn(L.qy, p[1]), d2 = __dsub_rn(L.qz, p[2]);
      const double d = __dadd_rn(__dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1)), __dmul_rn(d2, d2));
      seed_pi = pi; seed_d = d;
      if (d <= L.best) { L.best = d; L.bi = g.sidx[pi]; }
    }
  }
  // Temporal cache (BND builds in cache-aware rounds; the same test as nn_grid_kernel's): last search left a lower bound on the distance
  // to every target OTHER than the neighbour; the query has moved by exactly eps = |dM p + dv| since, so if the re-evaluated distance to
  // the old neighbour is below (bound - eps) it is still the unique nearest neighbour and its exact d2 is the answer.  Such a lane is
  // finished here and sits the traversal out: the wave walks the hierarchy for its MISSED lanes only — a patch of a few queries opens
  // one or two tiles instead of six — and a wave without a miss leaves at once.
  unsigned int n_hit = 0;
  if (BND && job.cache && has_xf && job.seed && L.active && seed_pi < 0 && sxf[24] == 0.0 && job.out_lb[i] == -1.f) {
    // last search found NO target within the search radius and this edge's query transform is bit-identical to that search's (allowance 0:
    // dM = dv = 0): the query is the same, so is the answer — nothing to search, nothing to write
    L.active = false;
    n_hit = 1;
  }
  if (BND && job.cache && has_xf && seed_pi >= 0) {
    const double cslack = sxf[24];
    if (cslack >= 0.0) {
      const double e0 = sxf[25] * p0 + sxf[28] * p1 + sxf[31] * p2 + sxf[34];
      const double e1 = sxf[26] * p0 + sxf[29] * p1 + sxf[32] * p2 + sxf[35];
      const double e2 = sxf[27] * p0 + sxf[30] * p1 + sxf[33] * p2 + sxf[36];
      const double eps = sqrt(e0 * e0 + e1 * e1 + e2 * e2) * (1.0 + 1e-9) + cslack;
      const double nlb = (double)job.out_lb[i] - eps;
      // ... or (round 6) the query is provably still REJECTED: its old neighbour is beyond the cutoff now (exact) and every other target was at least
      // out_lb away, i.e. is at least nlb away now — if that is beyond the cutoff too, no target is inside it, which is all the reference's filter
      // (frame.cpp:156) asks; the exact neighbour of a rejected query is never output.  out_d2 then holds the distance to the OLD neighbour (>= bound:
      // the query stays rejected downstream), out_idx keeps it as a seed, the bound is carried on.  These are the lanes with the LARGEST balls (their
      // thresholds reach the search radius): taking them out of the traversal is what makes a partial-overlap round cheap.
      const bool still_rejected = eps != 0.0 && seed_d >= bound && nlb > sqrt(bound) * (1.0 + 1e-9);
      if (eps == 0.0 || sqrt(seed_d) * (1.0 + 1e-12) < nlb || (job.reject_cache && still_rejected)) {   // (eps == 0: the same query bit for bit keeps last search's exact answer)
        if (eps != 0.0) {   // (eps == 0: bit-identical query transform, everything stored is already exact)
          job.out_d2[i] = seed_d;
          job.out_lb[i] = __double2float_rd(nlb);
          if (job.list.dirty) update_list_entry(job.list, i, seed_pi, seed_d, bound, true);
        }
        L.active = false;
        n_hit = 1;
      }
    }
  }
  if (BND && job.cache && __ballot(L.active) == 0ull) {
    if (stats) {   // census (profiling only): all 64 lanes answered by the cache
      const size_t slot = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (TT / 64) + wave;
      const unsigned long long hits = __popcll(__ballot(n_hit != 0u));
      if ((threadIdx.x & 63) == 0) stats[8 * slot + 3] = hits;
    }
    return;
  }
  Group G;   // wave-uniform: lives in SGPRs
  {
    // patch box in fp32, rounded outward: min / max of the lanes' float copies, widened by more than the half ulp a conversion can have
    // moved a coordinate inwards (six DPP reductions of 7 instructions; round 2 reduced the fp64 coordinates: ~30 instructions each)
    const float inf = __int_as_float(0x7f800000);
    const float a = (float)L.qx, b = (float)L.qy, c2 = (float)L.qz;
    float lo[3] = {wave_min_any(L.active ? a : inf), wave_min_any(L.active ? b : inf), wave_min_any(L.active ? c2 : inf)};
    float hi[3] = {wave_max_any(L.active ? a : -inf), wave_max_any(L.active ? b : -inf), wave_max_any(L.active ? c2 : -inf)};
    float m = (float)g.maxabs * 1.000001f;   // largest coordinate magnitude either operand of a difference can have: the cloud's box and the patch
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      lo[ax] -= fabsf(lo[ax]) * 1.2e-7f + 1e-37f; hi[ax] += fabsf(hi[ax]) * 1.2e-7f + 1e-37f;
      G.lo[ax] = lo[ax]; G.hi[ax] = hi[ax]; G.c[ax] = 0.5f * (lo[ax] + hi[ax]);
      m = fmaxf(m, fmaxf(fabsf(lo[ax]), fabsf(hi[ax])));
    }
    // per axis: |fl32(q) - q| + |fl32(p) - p| + rounding of the fp32 subtraction <= 3 * 2^-24 * m; x sqrt(3) axes, x2 safety
    G.slack = m * (3.0f * 1.7320508f * 2.0f / 16777216.0f) * 1.00001f + 1e-30f;
    G.mu = BND ? job.mu : 0.f;
  }
  { const float a = (float)L.qx, b = (float)L.qy, c2 = (float)L.qz; L.qx2 = f2v{a, a}; L.qy2 = f2v{b, b}; L.qz2 = f2v{c2, c2}; L.pad_ = 0.0; }
  L.thr = L.active ? thr_of(L.best, G.slack + G.mu) : -1.f;   // a finished / padding lane screens nothing (d32 >= 0 > -1)
  unsigned int n_cand = 0, n_box = 0;
  const int top = g.levels - 1;
  TileLds* T = &s_tile[wave];
  float2* sbox = s_box[wave];
  // cache-aware rounds: a wave left with at most job.miss_max missed lanes takes them one by one with the lanes spread over the data (miss_block)
  const bool miss_path = BND && job.cache && job.miss_max > 0 && __popcll(__ballot(L.active)) <= job.miss_max;
  if (TOP >= 0) visit<(TOP >= 0 ? TOP : 0), BND>(g, 0, g.cnt[TOP >= 0 ? TOP : 0], L, G, T, sbox, &n_cand, &n_box, miss_path);
  else switch (top) {
    case 0: visit<0, BND>(g, 0, g.cnt[0], L, G, T, sbox, &n_cand, &n_box, miss_path); break;
    case 1: visit<1, BND>(g, 0, g.cnt[1], L, G, T, sbox, &n_cand, &n_box, miss_path); break;
    case 2: visit<2, BND>(g, 0, g.cnt[2], L, G, T, sbox, &n_cand, &n_box, miss_path); break;
    case 3: visit<3, BND>(g, 0, g.cnt[3], L, G, T, sbox, &n_cand, &n_box, miss_path); break;
    default: visit<4, BND>(g, 0, g.cnt[4], L, G, T, sbox, &n_cand, &n_box, miss_path); break;
  }
  const unsigned int n_exam = n_cand;
  if (L.active) {
    const int out = i;   // sorted order of the source cloud
    job.out_idx[out] = L.bi == 0x7fffffff ? -1 : (job.inv ? job.inv[L.bi] : L.bi);
    job.out_d2[out] = L.best;
    // every other target was evaluated exactly (>= second) or rejected by a screen (> sqrt(best) + mu away); 1e-9 relative covers the
    // fp64 roundings of this line.  No neighbour inside the cutoff: 0 forces a full search next round, like the grid kernel does.
    if (BND) job.out_lb[out] = L.bi == 0x7fffffff ? -1.f : __double2float_rd(fmin(sqrt(L.second), sqrt(L.best) + (double)G.mu) * (1.0 - 1e-9));
    // the edge's compacted list is maintained in place (nn_list.h) whenever the host handed it over (list.dirty != null): a query that
    // keeps its acceptance patches its own entry and operands, so compaction + gather only run for edges whose MEMBERSHIP changed —
    // also in the plain seeded rounds, where nearly every neighbour changes but hardly any acceptance does (round 3)
    if (job.list.dirty) update_list_entry(job.list, i, L.bi == 0x7fffffff ? -1 : job.inv[L.bi], L.best, bound, false);
    if ((BND ? L.second == L.best : L.tie) && L.bi != 0x7fffffff) tie_report(job.tie, (unsigned int)i);
  }
  if (stats && (threadIdx.x & 63) == 0) {
    // wave-uniform counters: candidates examined PER LANE x active lanes, boxes tested per wave
    const size_t slot = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (TT / 64) + wave;
    const unsigned long long act = (unsigned long long)min(64, job.n - (i & ~63));
    stats[8 * slot] = n_cand; stats[8 * slot + 1] = n_box; stats[8 * slot + 2] = miss_path ? (unsigned long long)n_exam : (unsigned long long)n_cand * act;
  }
  if (stats && BND && job.cache) {
    const unsigned long long hits = __popcll(__ballot(n_hit != 0u));
    if ((threadIdx.x & 63) == 0) stats[8 * (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * (TT / 64) + wave) + 3] = hits;
  }
}

}  // namespace

int build_wide(FrameDev& f, const double* spts) {
  GridDev& G = f.grid;
  const int n = f.n;
  const float finf = std::numeric_limits<float>::infinity();
  auto down = [](double v) { float x = (float)v; if ((double)x > v) x = std::nextafterf(x, -std::numeric_limits<float>::infinity()); return x; };
  auto up = [](double v) { float x = (float)v; if ((double)x < v) x = std::nextafterf(x, std::numeric_limits<float>::infinity()); return x; };
  std::vector<std::vector<float>> lv;  // per level: 6 x cnt SoA
  std::vector<int> cnts;
  int cnt = (n + LEAF - 1) / LEAF;
  {
    std::vector<float> b(6 * (size_t)cnt);
    for (int j = 0; j < cnt; ++j) {
      float lo[3] = {finf, finf, finf}, hi[3] = {-finf, -finf, -finf};
      for (int k = j * LEAF; k < std::min(n, (j + 1) * LEAF); ++k)
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], down(spts[3 * (size_t)k + a])); hi[a] = std::max(hi[a], up(spts[3 * (size_t)k + a])); }
      for (int a = 0; a < 3; ++a) { b[(size_t)a * cnt + j] = lo[a]; b[(size_t)(3 + a) * cnt + j] = hi[a]; }
    }
    lv.push_back(b); cnts.push_back(cnt);
  }
  double maxabs = 0.0;
  for (size_t k = 0; k < 3 * (size_t)n; ++k) maxabs = std::max(maxabs, std::fabs(spts[k]));
  G.maxabs = maxabs;
  while (cnt > FAN) {
    const int pc = cnt;
    cnt = (pc + FAN - 1) / FAN;
    if (lv.size() >= 5) { set_error("cloud too large for the 64-wide hierarchy"); return MVICP_ERR_ARG; }
    const std::vector<float>& p = lv.back();
    std::vector<float> b(6 * (size_t)cnt);
    for (int j = 0; j < cnt; ++j) {
      float lo[3] = {finf, finf, finf}, hi[3] = {-finf, -finf, -finf};
      for (int k = j * FAN; k < std::min(pc, (j + 1) * FAN); ++k)
        for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[(size_t)a * pc + k]); hi[a] = std::max(hi[a], p[(size_t)(3 + a) * pc + k]); }
      for (int a = 0; a < 3; ++a) { b[(size_t)a * cnt + j] = lo[a]; b[(size_t)(3 + a) * cnt + j] = hi[a]; }
    }
    lv.push_back(b); cnts.push_back(cnt);
  }
  size_t total = 0;
  for (size_t l = 0; l < lv.size(); ++l) { G.wide_off[l] = (long long)total; G.wide_cnt[l] = cnts[l]; total += lv[l].size(); }
  G.wide_levels = (int)lv.size();
  std::vector<float> flat(total);
  for (size_t l = 0; l < lv.size(); ++l) std::copy(lv[l].begin(), lv[l].end(), flat.begin() + G.wide_off[l]);
  MV_HIP(hipMalloc((void**)&G.wide, sizeof(float) * std::max<size_t>(total, 1)));
  MV_HIP(hipMemcpy(G.wide, flat.data(), sizeof(float) * total, hipMemcpyHostToDevice));
  return MVICP_OK;
}

namespace { __global__ void tile_warm_kernel() {} }

int warm_nn_tile(mvicp_ctx* c) {   // see warm_nn_grid (nn_grid.hip): loads this file's code object at set-up time
  hipLaunchKernelGGL(tile_warm_kernel, dim3(1), dim3(64), 0, c->stream);
  MV_HIP(hipGetLastError());
  return MVICP_OK;
}

int launch_nn_tile_edges(mvicp_ctx* c, double d2_bound, bool with_bounds, bool with_cache, bool with_list) {
  // the matrix-pipe build of the same search (nn_mfma.hip).  Cache-aware rounds (a few missed lanes per wave) stay here by default: with
  // most lanes finished by the cache the per-lane box tests prune nearly every tile, and the 7-wave occupancy of this build hides the
  // latency of the few that remain (cfg4 rounds 6 / 7: 0.57 / 0.38 ms here against 0.65 / 0.57 ms there)
  if (c->tile_mfma >= 2 || (c->tile_mfma == 1 && (!with_cache || c->cached_on_mfma))) return launch_nn_mfma_edges(c, d2_bound, with_bounds, with_cache, with_list);
  std::vector<TileJob> jobs;
  int max_n = 0;
  double nq = 0;
  std::vector<TieJob> ties;
  MV_CHECK(build_tile_jobs(c, with_bounds, with_cache, with_list, jobs, max_n, nq, ties));
  if (jobs.empty() || max_n == 0) return MVICP_OK;
  TileJob* d_jobs = nullptr;
  MV_CHECK(cached_upload(c, jobs[0].xf ? "tile_jobs" : "tile_jobs_raw", jobs.data(), sizeof(TileJob) * jobs.size(), (void**)&d_jobs));
  unsigned long long* d_stats = nullptr;
  const size_t slots = (size_t)((max_n + TT - 1) / TT) * jobs.size() * (TT / 64);
  MV_CHECK(census_scratch(c, slots, &d_stats));
  {
    ProfScope ps(c, "nn_tile", 36.0 * nq);  // query read 24 B + result write 12 B; candidate / box bytes come from the census
    const dim3 grid((max_n + TT - 1) / TT, (unsigned)jobs.size());
    int top = jobs[0].dst.levels - 1;   // same depth everywhere -> the traversal specialised for it
    for (const TileJob& j : jobs) if (j.dst.levels - 1 != top) top = -1;
#define MVICP_TILE_K(W, T, B) hipLaunchKernelGGL((nn_tile_kernel<W, T, B>), grid, dim3(TT), 0, c->stream, d_jobs, d2_bound, search_bound(c, d2_bound), d_stats)
    const int waves = c->tile_waves;   // 0 = pick: 7 waves per SIMD for the depth-3 build, 6 otherwise
    // depth-3 build: 7 waves per SIMD (no spills at 66 VGPRs); 6 and 8 measure 4-10 % slower (profiles/r03_tile_ab.txt)
    if (with_bounds) {   // hand-over round: one more fp64 register pair per lane, so one wave per SIMD fewer
      if (top == 2) MVICP_TILE_K(6, 2, true); else MVICP_TILE_K(6, -1, true);
    }
    else if (top == 2 && (waves == 0 || waves == 7)) MVICP_TILE_K(7, 2, false);
    else if (top == 2 && waves == 8) MVICP_TILE_K(8, 2, false);
    else if (top == 2 && waves == 6) MVICP_TILE_K(6, 2, false);
    else if (top == 1 && (waves == 0 || waves == 6)) MVICP_TILE_K(6, 1, false);
    else if (waves == 8) MVICP_TILE_K(8, -1, false);
    else MVICP_TILE_K(6, -1, false);
#undef MVICP_TILE_K
  }
  MV_HIP(hipGetLastError());
  MV_CHECK(launch_tie_fixup(c, ties, d2_bound));     // exact distance ties: the reference's own descent decides (nn_tie.hip); before the lists are read
  if (with_list) MV_CHECK(launch_dirty_reduce(c));   // per-edge OR of the "list membership changed" slots
  if (d_stats) MV_CHECK(census_collect(c, d_stats, slots, nq, "nn_tile"));
  return MVICP_OK;
}

}  // namespace mvicp
