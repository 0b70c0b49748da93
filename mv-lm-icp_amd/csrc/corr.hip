// K3/K4 — cutoff filter + stable compaction, packed-stream gather, exact median (compacting radix select / bracket select).
//
// Replaces the tail of Frame::computeClosestPointsToNeighbours (src/internal/frame.cpp:156-176):
//   if (sqrt(d2) < thresh) push {k, idx, dist}        -> flag / exclusive scan / scatter, ascending k
//   nth_element(dists, size/2); weight = 1.5 * median -> radix select on the fp64 bit pattern of d2
// The acceptance test is evaluated as `d2 < bound` where the host has computed `bound` = the smallest
// double whose correctly rounded sqrt is >= (double)thresh, so the decision is bit-identical to the
// reference's `sqrt(d2) < thresh` without a device sqrt.  sqrt is monotone, so the median of the
// distances is the sqrt of the median of d2; the host takes that one sqrt (IEEE, exact).
//
// The gather kernel also materialises, once per ICP round, the per-correspondence operand stream the
// LM evaluations re-read up to ~100 times: SoA p (3) | n (3) | c = n . q | q (3), 10 arrays, of which point-to-plane
// streams 7 (56 B per correspondence) and point-to-point 6 — instead of 2-3 random 24-B gathers per evaluation.
#include "common.h"

namespace mvicp {

namespace {

constexpr int NT = 256;
constexpr int IPT = kCompactBlock / NT;  // 4 consecutive queries per thread

__device__ __forceinline__ int find_edge(const int* __restrict__ off, int E, int b) {
  int lo = 0, hi = E;  // largest e with off[e] <= b
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= b) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ int block_exclusive_scan(int v, int* __restrict__ wave_tot, int* total) {
  // inclusive scan inside the wave (64 lanes) by shuffles, then across the 4 waves through LDS
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(inc, d, 64);
    if (lane >= d) inc += o;
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) {
    const int t = wave_tot[w];
    if (w < wave) base += t;
    tot += t;
  }
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(NT) void count_kernel(const int* __restrict__ cblock_off, int E, const int* __restrict__ nsrc,
                                                   const long long* __restrict__ cap_off, const int* __restrict__ nn_idx,
                                                   const double* __restrict__ nn_d2, double bound, int* __restrict__ cblock_cnt,
                                                   const int* __restrict__ dirty) {
  __shared__ int wave_tot[NT / 64];
  const int b = blockIdx.x;
  const int e = find_edge(cblock_off, E, b);
  if (dirty[e] == 0) return;  // list unchanged since last round
  const int lb = b - cblock_off[e];
  const int n = nsrc[e];
  const long long base = cap_off[e];
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int k = lb * kCompactBlock + threadIdx.x * IPT + i;
    if (k < n) cnt += (nn_idx[base + k] >= 0 && nn_d2[base + k] < bound) ? 1 : 0;
  }
  int total;
  block_exclusive_scan(cnt, wave_tot, &total);
  if (threadIdx.x == 0) cblock_cnt[b] = total;
}

// one workgroup per edge: exclusive scan of its block counts (<= ~1k blocks for 1M points), in place.
__global__ __launch_bounds__(NT) void scan_kernel(const int* __restrict__ cblock_off, int* __restrict__ cblock_cnt, int* __restrict__ count,
                                                  const int* __restrict__ dirty) {
  __shared__ int wave_tot[NT / 64];
  __shared__ int carry_s;
  const int e = blockIdx.x;
  if (dirty[e] == 0) return;
  const int b0 = cblock_off[e], b1 = cblock_off[e + 1];
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int s = b0; s < b1; s += NT) {
    const int b = s + threadIdx.x;
    const int v = (b < b1) ? cblock_cnt[b] : 0;
    int total;
    const int ex = block_exclusive_scan(v, wave_tot, &total);
    const int carry = carry_s;
    if (b < b1) cblock_cnt[b] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) count[e] = carry_s;
}

__global__ __launch_bounds__(NT) void scatter_kernel(const int* __restrict__ cblock_off, int E, const int* __restrict__ nsrc,
                                                     const long long* __restrict__ cap_off, const int* __restrict__ nn_idx,
                                                     const double* __restrict__ nn_d2, double bound, const int* __restrict__ cblock_cnt,
                                                     int* __restrict__ first, int* __restrict__ second, double* __restrict__ cd2,
                                                     int* __restrict__ qpos, const int* __restrict__ dirty) {
  __shared__ int wave_tot[NT / 64];
  const int b = blockIdx.x;
  const int e = find_edge(cblock_off, E, b);
  if (dirty[e] == 0) return;
  const int lb = b - cblock_off[e];
  const int n = nsrc[e];
  const long long base = cap_off[e];
  int idx[IPT];
  double d2[IPT];
  bool ok[IPT];
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int k = lb * kCompactBlock + threadIdx.x * IPT + i;
    ok[i] = false;
    if (k < n) {
      idx[i] = nn_idx[base + k];
      d2[i] = nn_d2[base + k];
      ok[i] = idx[i] >= 0 && d2[i] < bound;
    }
    cnt += ok[i] ? 1 : 0;
  }
  int total;
  int pos = cblock_cnt[b] + block_exclusive_scan(cnt, wave_tot, &total);
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int k = lb * kCompactBlock + threadIdx.x * IPT + i;
    if (k < n) qpos[base + k] = ok[i] ? pos : -1;
    if (!ok[i]) continue;
    first[base + pos] = k;
    second[base + pos] = idx[i];
    cd2[base + pos] = d2[i];
    ++pos;
  }
}

// packed operand stream for the LM evaluations
__global__ __launch_bounds__(NT) void gather_kernel(const int* __restrict__ cblock_off, int E, const int* __restrict__ count,
                                                    const long long* __restrict__ cap_off, long long total_cap, const int* __restrict__ first,
                                                    const int* __restrict__ second, const PointRec* const* __restrict__ src_rec,
                                                    const PointRec* const* __restrict__ dst_rec, const double* const* __restrict__ dst_nor,
                                                    double* __restrict__ stream, const int* __restrict__ dirty) {
  const int b = blockIdx.x;
  const int e = find_edge(cblock_off, E, b);
  if (dirty[e] == 0) return;  // same list -> same operands
  const int lb = b - cblock_off[e];
  const int cnt = count[e];
  if (lb * kCompactBlock >= cnt) return;
  const long long base = cap_off[e];
  const PointRec* __restrict__ sp = src_rec[e];   // first / second are SORTED positions: neighbouring correspondences
  const PointRec* __restrict__ dp = dst_rec[e];   // read neighbouring records (coherent gathers)
  const double* __restrict__ dn = dst_nor[e];
  for (int i = 0; i < IPT; ++i) {
    const int pos = lb * kCompactBlock + i * NT + threadIdx.x;
    if (pos >= cnt) break;
    const size_t f = (size_t)first[base + pos], s = (size_t)second[base + pos];
    const size_t o = (size_t)(base + pos);
    const double2* pa = reinterpret_cast<const double2*>(sp + f);
    const double2 a0 = pa[0], a1 = pa[1];
    const double2* pb = reinterpret_cast<const double2*>(dp + s);
    const double2 b0 = pb[0], b1 = pb[1];
    // arrays: 0-2 p | 3-5 n | 6 c = n . q | 7-9 q   (linearize.hip: point-to-plane reads 0..6, point-to-point 0-2 and 7-9)
    __builtin_nontemporal_store(a0.x, &stream[0 * total_cap + o]);
    __builtin_nontemporal_store(a0.y, &stream[1 * total_cap + o]);
    __builtin_nontemporal_store(a1.x, &stream[2 * total_cap + o]);
    __builtin_nontemporal_store(b0.x, &stream[7 * total_cap + o]);
    __builtin_nontemporal_store(b0.y, &stream[8 * total_cap + o]);
    __builtin_nontemporal_store(b1.x, &stream[9 * total_cap + o]);
    if (dn != nullptr) {
      const double n0 = dn[3 * s], n1 = dn[3 * s + 1], n2 = dn[3 * s + 2];
      __builtin_nontemporal_store(n0, &stream[3 * total_cap + o]);
      __builtin_nontemporal_store(n1, &stream[4 * total_cap + o]);
      __builtin_nontemporal_store(n2, &stream[5 * total_cap + o]);
      __builtin_nontemporal_store(n0 * b0.x + n1 * b0.y + n2 * b1.x, &stream[6 * total_cap + o]);
    }
  }
}

// ---- radix select over the fp64 patterns of the accepted d2: 3 streaming passes x 11 bits + an in-LDS finish ------------
// d2 >= 0, so bit 63 is clear and the unsigned pattern is monotone in the value.  Digits, MSB first:
//   A = bits 62..52 (the exponent), B = bits 51..41, C = bits 40..30, then 30 low bits resolved by one workgroup per edge.
// Pass A histograms every key.  Pass B histograms the keys whose exponent is the picked one AND appends them to a
// compact buffer (10-30 % of the list); pass C reads only that buffer, histograms the keys matching the 22-bit prefix and
// appends those (a few dozen per edge) to a second buffer, which the finishing workgroup resolves with three 10-bit
// passes in LDS.  Two full reads of the list instead of eight; every step is exact for any input (all-equal keys just
// make the compact buffers as long as the list).  The digit of a pass is picked by a one-workgroup-per-edge kernel between
// the passes (a "last workgroup picks" ticket scheme needs an agent-scope fence per workgroup, which on the 8-XCD part
// writes back / invalidates L2 and made the select 4x slower).
constexpr int kSelBins = 2048;
struct SelState { unsigned long long prefix; int k; int pad; };

// Workgroup-wide: which of `nbins` bins (nbins = NT * PER) holds rank k, and how many keys sit in the bins below it.
// `get(bin)` returns the count of a bin.  All NT threads must call; result valid in every thread.
template <int PER, typename F>
__device__ __forceinline__ void pick_bin(F get, unsigned int k, int* __restrict__ wave_tot, int* __restrict__ sh_pick, int& bin, unsigned int& below) {
  unsigned int h[PER];
  int mine = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) { h[i] = get(threadIdx.x * PER + i); mine += (int)h[i]; }
  int total;
  const int ex = block_exclusive_scan(mine, wave_tot, &total);
  if (threadIdx.x == 0) { sh_pick[0] = NT * PER - 1; sh_pick[1] = total; }   // unreachable default (k < total always)
  __syncthreads();
  if ((unsigned int)ex <= k && k < (unsigned int)(ex + mine)) {
    unsigned int cum = (unsigned int)ex;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      if (cum + h[i] > k) { sh_pick[0] = threadIdx.x * PER + i; sh_pick[1] = (int)cum; break; }
      cum += h[i];
    }
  }
  __syncthreads();
  bin = sh_pick[0]; below = (unsigned int)sh_pick[1];
  __syncthreads();
}

// PASS 0: keys = cd2 (count[e]), no filter, no output.   PASS 1: keys = cd2, filter on digit A, output -> out_keys/out_cnt.
// PASS 2: keys = in_keys (in_cnt[e]), filter on digits A,B, output -> out_keys/out_cnt.
template <int PASS>
__global__ __launch_bounds__(NT) void select_pass_kernel(const int* __restrict__ sblock_off, int E, const int* __restrict__ count,
                                                         const long long* __restrict__ cap_off, const double* __restrict__ in_keys,
                                                         const unsigned int* __restrict__ in_cnt, unsigned int* __restrict__ hist,
                                                         const SelState* __restrict__ state,
                                                         double* __restrict__ out_keys, unsigned int* __restrict__ out_cnt) {
  constexpr int SPT = kSelBlock / NT;                       // keys per thread
  constexpr int SHIFT = PASS == 0 ? 52 : PASS == 1 ? 41 : 30;
  __shared__ unsigned int lh[kSelBins];
  __shared__ int wave_tot[NT / 64];
  __shared__ unsigned int sh_base;
  const int b = blockIdx.x;
  const int e = find_edge(sblock_off, E, b);
  const int lb = b - sblock_off[e];
  const int cnt = count[e];
  const int n_in = PASS == 2 ? (cnt > 0 ? (int)in_cnt[e] : 0) : cnt;
  if ((long long)lb * kSelBlock >= n_in) return;
  SelState st;
  st.prefix = 0ull; st.k = cnt / 2; st.pad = 0;              // dists.begin() + size()/2  (frame.cpp:166)
  if (PASS > 0) st = state[(size_t)(PASS - 1) * E + e];
  for (int i = threadIdx.x; i < kSelBins; i += NT) lh[i] = 0u;
  __syncthreads();
  const long long base = cap_off[e];
  unsigned long long keys[SPT];
  int nmatch = 0;
#pragma unroll
  for (int i = 0; i < SPT; ++i) {
    const int pos = lb * kSelBlock + i * NT + threadIdx.x;
    bool match = pos < n_in;
    unsigned long long key = 0ull;
    if (match) {
      key = (unsigned long long)__double_as_longlong(__builtin_nontemporal_load(&in_keys[base + pos]));
      if (PASS > 0) match = (key >> (SHIFT + 11)) == (st.prefix >> (SHIFT + 11));
    }
    const unsigned int bin = (unsigned int)(key >> SHIFT) & (kSelBins - 1);
    if (PASS == 0) {
      // exponent digit: the 64 keys of a wave share a handful of exponents, so plain LDS atomics would serialise on a few
      // addresses.  Exponent fields 768..1023 (values in [2^-255, 2): every realistic squared distance) are counted in
      // EIGHT interleaved copies of a 256-bin window, picked by the lane number; anything else (exact zeros, huge cutoffs)
      // is rare and goes straight to the global histogram, one atomic per distinct value per wave.
      const int lane = threadIdx.x & 63;
      const bool inwin = match && bin >= 768u && bin < 1024u;
      if (inwin) atomicAdd(&lh[(bin & 255u) | ((unsigned int)(lane & 7) << 8)], 1u);
      unsigned long long todo = __ballot(match && !inwin);
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const unsigned int lbin = (unsigned int)__builtin_amdgcn_readlane((int)bin, leader);
        const unsigned long long same = __ballot(match && !inwin && bin == lbin) & todo;
        if (lane == leader) atomicAdd(&hist[((size_t)PASS * E + e) * kSelBins + lbin], (unsigned int)__popcll(same));
        todo &= ~same;
      }
    } else if (match) {
      atomicAdd(&lh[bin], 1u);
    }
    keys[i] = key;
    if (PASS > 0 && match) { keys[i] |= 1ull << 63; ++nmatch; }   // bit 63 is free (d2 >= 0): mark the keys to keep
  }
  if (PASS > 0) {
    int total;
    int off = block_exclusive_scan(nmatch, wave_tot, &total);
    if (threadIdx.x == 0) sh_base = total ? atomicAdd(&out_cnt[e], (unsigned int)total) : 0u;
    __syncthreads();
    off += (int)sh_base;
#pragma unroll
    for (int i = 0; i < SPT; ++i)
      if (keys[i] >> 63) out_keys[base + off++] = __longlong_as_double((long long)(keys[i] & ~(1ull << 63)));
  }
  __syncthreads();
  if (PASS == 0) {
    unsigned int v = 0;   // fold the 8 copies of the exponent window
#pragma unroll
    for (int cpy = 0; cpy < 8; ++cpy) v += lh[cpy * 256 + threadIdx.x];
    if (v) atomicAdd(&hist[((size_t)PASS * E + e) * kSelBins + 768 + threadIdx.x], v);
  } else {
    for (int i = threadIdx.x; i < kSelBins; i += NT) {
      const unsigned int v = lh[i];
      if (v) atomicAdd(&hist[((size_t)PASS * E + e) * kSelBins + i], v);
    }
  }
}

// digit pick of pass PASS (0 = A, 1 = B): one workgroup per edge over the finished histogram
template <int PASS>
__global__ __launch_bounds__(NT) void select_pick_kernel(int E, const int* __restrict__ count, const unsigned int* __restrict__ hist,
                                                         SelState* __restrict__ state) {
  constexpr int SHIFT = PASS == 0 ? 52 : 41;
  __shared__ int wave_tot[NT / 64];
  __shared__ int sh_pick[2];
  const int e = blockIdx.x;
  const int cnt = count[e];
  if (cnt <= 0) return;
  SelState st;
  st.prefix = 0ull; st.k = cnt / 2; st.pad = 0;              // dists.begin() + size()/2  (frame.cpp:166)
  if (PASS > 0) st = state[(size_t)(PASS - 1) * E + e];
  const unsigned int* hp = hist + ((size_t)PASS * E + e) * kSelBins;
  int bin; unsigned int below;
  pick_bin<kSelBins / NT>([&](int i) { return hp[i]; }, (unsigned int)st.k, wave_tot, sh_pick, bin, below);
  if (threadIdx.x == 0) {
    SelState o;
    o.prefix = st.prefix | ((unsigned long long)bin << SHIFT);
    o.k = st.k - (int)below; o.pad = 0;
    state[(size_t)PASS * E + e] = o;
  }
}

// SoftLOne scale of an edge on the device: a = (double)(float)(1.5 * sqrt(median d2)) — OutgoingEdge::weight (frame.cpp:168-176) widened
// the way Ceres reads it (icp-ceres.cpp:284,374,449).  Lets the first LM evaluation of the round be queued before the host has
// seen the median; the host recomputes the weight with its own IEEE sqrt and only trusts that evaluation if the two agree bit for bit.
__device__ __forceinline__ void write_a_scale(int cnt, double med, int e, double* __restrict__ a_dev, double* __restrict__ a_host) {
  if (a_dev == nullptr) return;
  const double a = cnt > 0 ? (double)(float)__dmul_rn(sqrt(med), 1.5) : 0.0;
  a_dev[e] = a;
  if (a_host) a_host[e] = a;
}

// one workgroup per edge: pick digit C, then the keys matching the 33-bit prefix (out of the second compact buffer) ->
// 3 x 10-bit passes in LDS.  Also hands (count, median d2) to the host through the mapped result buffer.
__global__ __launch_bounds__(NT) void select_final_kernel(int E, const int* __restrict__ count, const long long* __restrict__ cap_off,
                                                          const double* __restrict__ keys2, const unsigned int* __restrict__ cnt2,
                                                          const unsigned int* __restrict__ hist, const SelState* __restrict__ state,
                                                          double* __restrict__ median, double* __restrict__ host_res,
                                                          double* __restrict__ a_dev, double* __restrict__ a_host, double armed) {
  __shared__ unsigned int lh[1024];
  __shared__ int wave_tot[NT / 64];
  __shared__ int sh_pick[2];
  const int e = blockIdx.x;
  const int cnt = count[e];
  double med = 0.0;
  if (cnt > 0) {
    SelState st = state[(size_t)1 * E + e];
    {
      const unsigned int* hp = hist + ((size_t)2 * E + e) * kSelBins;
      int bin; unsigned int below;
      pick_bin<kSelBins / NT>([&](int i) { return hp[i]; }, (unsigned int)st.k, wave_tot, sh_pick, bin, below);
      st.prefix |= (unsigned long long)bin << 30;
      st.k -= (int)below;
    }
    const int n2 = (int)cnt2[e];
    const long long base = cap_off[e];
    for (int shift = 20; shift >= 0; shift -= 10) {
      for (int i = threadIdx.x; i < 1024; i += NT) lh[i] = 0u;
      __syncthreads();
      for (int pos = threadIdx.x; pos < n2; pos += NT) {
        const unsigned long long key = (unsigned long long)__double_as_longlong(keys2[base + pos]);
        if ((key >> (shift + 10)) == (st.prefix >> (shift + 10))) atomicAdd(&lh[(key >> shift) & 1023ull], 1u);
      }
      __syncthreads();
      int bin; unsigned int below;
      pick_bin<1024 / NT>([&](int i) { return lh[i]; }, (unsigned int)st.k, wave_tot, sh_pick, bin, below);
      st.prefix |= (unsigned long long)bin << shift;
      st.k -= (int)below;
    }
    med = __longlong_as_double((long long)st.prefix);
  }
  if (threadIdx.x == 0) {
    median[e] = med;
    if (host_res) {
      host_res[2 * e] = (double)(cnt > 0 ? cnt : 0); host_res[2 * e + 1] = med;
      if (e == 0) host_res[2 * E] = armed;   // slot behind the E pairs: 1 if this rank queued the speculative evaluation (summed over ranks)
    }
    write_a_scale(cnt, med, e, a_dev, a_host);
  }
}

// ---- bracket select: when the registration has settled, the median of an edge barely moves between rounds.  ONE pass
// over the keys counts those below a bracket [lo, hi] around last round's median and compacts the few per cent inside it;
// one workgroup per edge then selects the wanted rank among the compacted keys in LDS.  Exact whenever the rank falls inside
// the bracket; otherwise the edge's result slot is flagged (median = -1) and the host runs the full radix select.
__global__ __launch_bounds__(NT) void bracket_pass_kernel(const int* __restrict__ sblock_off, int E, const int* __restrict__ count,
                                                          const long long* __restrict__ cap_off, const double* __restrict__ keys,
                                                          const double* __restrict__ lohi, unsigned int* __restrict__ cnt_lt,
                                                          double* __restrict__ out_keys, unsigned int* __restrict__ out_cnt) {
  // One streaming read of the keys, 16 B per lane per load; no workgroup scans: the count below the bracket is a popcount per wave,
  // the rare keys inside the bracket (about 1 %) get their slot from an LDS counter, and each workgroup touches the two per-edge
  // global counters once (hot-address atomics per wave were measured 4x slower than the whole pass).
  constexpr int SPT = kSelBlock / NT;   // keys per thread, two per step
  __shared__ unsigned int s_lt[NT / 64];
  __shared__ unsigned int s_mid, s_base;
  const int b = blockIdx.x;
  const int e = find_edge(sblock_off, E, b);
  const int lb = b - sblock_off[e];
  const int cnt = count[e];
  if ((long long)lb * kSelBlock >= cnt) return;
  if (threadIdx.x == 0) s_mid = 0u;
  __syncthreads();
  const unsigned long long lo = (unsigned long long)__double_as_longlong(lohi[2 * e]), hi = (unsigned long long)__double_as_longlong(lohi[2 * e + 1]);
  const long long base = cap_off[e];   // multiple of 64 keys: 16-B aligned pairs
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned int nlt = 0, nmid = 0;
  unsigned long long kept[SPT];
#pragma unroll
  for (int i = 0; i < SPT / 2; ++i) {
    const int pos = lb * kSelBlock + 2 * (i * NT + threadIdx.x);
    unsigned long long k0 = ~0ull, k1 = ~0ull;   // ~0: not a key (d2 >= 0 has bit 63 clear)
    if (pos + 1 < cnt) {
      typedef double d2v __attribute__((ext_vector_type(2)));
      const d2v v = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(keys + base + pos));
      k0 = (unsigned long long)__double_as_longlong(v.x); k1 = (unsigned long long)__double_as_longlong(v.y);
    } else if (pos < cnt) {
      k0 = (unsigned long long)__double_as_longlong(keys[base + pos]);
    }
    nlt += (k0 < lo ? 1u : 0u) + (k1 < lo ? 1u : 0u);
    const bool m0 = k0 >= lo && k0 <= hi, m1 = k1 >= lo && k1 <= hi;   // (~0 > hi always)
    kept[2 * i] = m0 ? k0 : ~0ull; kept[2 * i + 1] = m1 ? k1 : ~0ull;
    nmid += (m0 ? 1u : 0u) + (m1 ? 1u : 0u);
  }
  unsigned int off = nmid ? atomicAdd(&s_mid, nmid) : 0u;   // LDS
  unsigned int w = nlt;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) w += __shfl_xor(w, d, 64);
  if (lane == 0) s_lt[wave] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = s_lt[0] + s_lt[1] + s_lt[2] + s_lt[3];
    if (t) atomicAdd(&cnt_lt[e], t);
    s_base = s_mid ? atomicAdd(&out_cnt[e], s_mid) : 0u;
  }
  __syncthreads();
  if (nmid) {
    off += s_base;
#pragma unroll
    for (int i = 0; i < SPT; ++i)
      if (kept[i] != ~0ull) out_keys[base + off++] = __longlong_as_double((long long)kept[i]);
  }
}

__global__ __launch_bounds__(NT) void bracket_final_kernel(int E, const int* __restrict__ count, const long long* __restrict__ cap_off,
                                                           const double* __restrict__ keys1, unsigned int* __restrict__ cnt_lt,
                                                           unsigned int* __restrict__ cnt_mid, double* __restrict__ median,
                                                           double* __restrict__ host_res, double* __restrict__ a_dev, double* __restrict__ a_host, double armed) {
  __shared__ unsigned int lh[kSelBins];
  __shared__ int wave_tot[NT / 64];
  __shared__ int sh_pick[2];
  const int e = blockIdx.x;
  const int cnt = count[e];
  double med = 0.0;
  bool ok = true;
  if (cnt > 0) {
    const unsigned int k = (unsigned int)(cnt / 2), nlt = cnt_lt[e], nmid = cnt_mid[e];
    ok = nlt <= k && k < nlt + nmid;
    if (ok) {
      unsigned long long prefix = 0ull;
      unsigned int rank = k - nlt;
      const long long base = cap_off[e];
      // the compacted keys are few (about 1 % of the list): keep them in registers across the 6 digit passes
      constexpr int KR = 8;
      const bool in_regs = nmid <= (unsigned int)(KR * NT);
      unsigned long long kreg[KR];
#pragma unroll
      for (int i = 0; i < KR; ++i) {
        const unsigned int pos = threadIdx.x + i * NT;
        kreg[i] = (in_regs && pos < nmid) ? (unsigned long long)__double_as_longlong(keys1[base + pos]) : ~0ull;
      }
      // 63 key bits, MSB first: 11 + 11 + 11 + 10 + 10 + 10
      int hi_bit = 63;
#pragma unroll 1
      for (int pass = 0; pass < 6; ++pass) {
        const int bits = pass < 3 ? 11 : 10;
        const int shift = hi_bit - bits;
        for (int i = threadIdx.x; i < kSelBins; i += NT) lh[i] = 0u;
        __syncthreads();
        if (in_regs) {
#pragma unroll
          for (int i = 0; i < KR; ++i) {   // (~0 never matches: bit 63 of the prefix is clear)
            const unsigned long long key = kreg[i];
            if ((key >> hi_bit) == (prefix >> hi_bit)) atomicAdd(&lh[(key >> shift) & ((1ull << bits) - 1ull)], 1u);
          }
        } else {
          for (unsigned int pos = threadIdx.x; pos < nmid; pos += NT) {
            const unsigned long long key = (unsigned long long)__double_as_longlong(keys1[base + pos]);
            if ((key >> hi_bit) == (prefix >> hi_bit)) atomicAdd(&lh[(key >> shift) & ((1ull << bits) - 1ull)], 1u);
          }
        }
        __syncthreads();
        int bin; unsigned int below;
        pick_bin<kSelBins / NT>([&](int i) { return lh[i]; }, rank, wave_tot, sh_pick, bin, below);
        prefix |= (unsigned long long)bin << shift;
        rank -= below;
        hi_bit = shift;
      }
      med = __longlong_as_double((long long)prefix);
    }
  }
  if (threadIdx.x == 0) {
    if (ok) median[e] = med;
    host_res[2 * e] = (double)(cnt > 0 ? cnt : 0);
    host_res[2 * e + 1] = ok ? med : -1.0;   // -1: rank outside the bracket -> the host falls back to the full select
    if (e == 0) host_res[2 * E] = armed;
    write_a_scale(ok ? cnt : 0, med, e, a_dev, a_host);
  }
  // leave the two counters zeroed for the next round's bracket pass (no memset in the steady-state launch sequence)
  __syncthreads();
  if (threadIdx.x == 0) { cnt_lt[e] = 0u; cnt_mid[e] = 0u; }
}

}  // namespace

int launch_compact(mvicp_ctx* c, double d2_bound) {
  if (c->n_cblocks == 0) return MVICP_OK;
  double bytes = 0;
  for (int e = 0; e < c->E; ++e) if (c->owned[e]) bytes += 2.0 * 12.0 * c->frames[c->esrc[e]].n;
  ProfScope ps(c, "compact", bytes);
  hipLaunchKernelGGL(count_kernel, dim3(c->n_cblocks), dim3(NT), 0, c->stream, c->d_cblock_off, c->E, c->d_nsrc, c->d_cap_off, c->d_nn_idx,
                     c->d_nn_d2, d2_bound, c->d_cblock_cnt, c->d_dirty);
  hipLaunchKernelGGL(scan_kernel, dim3(c->E), dim3(NT), 0, c->stream, c->d_cblock_off, c->d_cblock_cnt, c->d_count, c->d_dirty);
  hipLaunchKernelGGL(scatter_kernel, dim3(c->n_cblocks), dim3(NT), 0, c->stream, c->d_cblock_off, c->E, c->d_nsrc, c->d_cap_off, c->d_nn_idx,
                     c->d_nn_d2, d2_bound, c->d_cblock_cnt, c->d_first, c->d_second, c->d_cd2, c->d_qpos, c->d_dirty);
  MV_HIP(hipGetLastError());
  return MVICP_OK;
}

int launch_gather_stream(mvicp_ctx* c) {
  if (c->n_cblocks == 0) return MVICP_OK;
  // per-edge base pointers (device table lives in the pinned staging area's device twin: small, rebuilt per call)
  std::vector<const void*> tab(3 * (size_t)c->E, nullptr);
  for (int e = 0; e < c->E; ++e) {
    tab[e] = c->frames[c->esrc[e]].grid.srec;
    tab[c->E + e] = c->frames[c->edst[e]].grid.srec;
    tab[2 * c->E + e] = c->frames[c->edst[e]].grid.snor;
  }
  const void** d_tab = nullptr;
  MV_CHECK(cached_upload(c, "gather_tab", tab.data(), sizeof(void*) * tab.size(), (void**)&d_tab));
  {
    ProfScope ps(c, "gather", 0.0);
    hipLaunchKernelGGL(gather_kernel, dim3(c->n_cblocks), dim3(NT), 0, c->stream, c->d_cblock_off, c->E, c->d_count, c->d_cap_off, c->total_cap,
                       c->d_first, c->d_second, (const PointRec* const*)d_tab, (const PointRec* const*)(d_tab + c->E), (const double* const*)(d_tab + 2 * c->E), c->d_stream, c->d_dirty);
  }
  MV_HIP(hipGetLastError());
  return MVICP_OK;
}

int launch_select_bracket(mvicp_ctx* c) {
  if (c->E == 0) return MVICP_OK;
  double bytes = 0;
  for (int e = 0; e < c->E; ++e) if (c->owned[e]) bytes += 8.0 * c->h_count[e];   // one full read of the key list
  ProfScope ps(c, "select", bytes);
  const size_t E = (size_t)c->E;
  unsigned int* cnt_lt = c->d_sel_hist;   // reuses the histogram scratch: cnt_lt [E] | cnt_mid [E]
  unsigned int* cnt_mid = cnt_lt + E;
  if (c->n_sblocks) {
    if (!c->bracket_counters_clean) MV_HIP(hipMemsetAsync(cnt_lt, 0, sizeof(unsigned int) * 2 * E, c->stream));
    c->bracket_counters_clean = true;   // bracket_final_kernel re-zeroes what it used
    hipLaunchKernelGGL(bracket_pass_kernel, dim3(c->n_sblocks), dim3(NT), 0, c->stream, c->d_sblock_off, c->E, c->d_count, c->d_cap_off, c->d_cd2,
                       (const double*)c->d_sel_lohi, cnt_lt, c->d_sel_keys1, cnt_mid);
  }
  hipLaunchKernelGGL(bracket_final_kernel, dim3(c->E), dim3(NT), 0, c->stream, c->E, c->d_count, c->d_cap_off, c->d_sel_keys1, cnt_lt,
                     cnt_mid, c->d_median, c->d_res_target ? c->d_res_target : c->d_res_host, c->spec_arm ? c->d_a : (double*)nullptr,
                     c->spec_arm ? c->d_a_check : (double*)nullptr, c->spec_arm ? 1.0 : 0.0);
  MV_HIP(hipGetLastError());
  return MVICP_OK;
}

int launch_select_median(mvicp_ctx* c) {
  if (c->E == 0) return MVICP_OK;
  double bytes = 0;
  for (int e = 0; e < c->E; ++e) if (c->owned[e]) bytes += 16.0 * c->h_count[e];   // two full reads of the key list (last round's length)
  ProfScope ps(c, "select", bytes);
  const size_t E = (size_t)c->E;
  unsigned int* hist = c->d_sel_hist;                 // [3][E][2048] | cnt1 [E] | cnt2 [E]  (one memset)
  unsigned int* cnt1 = hist + 3 * E * kSelBins;
  unsigned int* cnt2 = cnt1 + E;
  SelState* st = (SelState*)c->d_sel_state;
  c->bracket_counters_clean = false;   // the radix passes use the same scratch
  if (c->n_sblocks) {
    MV_HIP(hipMemsetAsync(hist, 0, sizeof(unsigned int) * (3 * E * kSelBins + 2 * E), c->stream));
    const dim3 grid(c->n_sblocks), blk(NT);
    hipLaunchKernelGGL((select_pass_kernel<0>), grid, blk, 0, c->stream, c->d_sblock_off, c->E, c->d_count, c->d_cap_off, c->d_cd2, (const unsigned int*)nullptr,
                       hist, (const SelState*)st, (double*)nullptr, (unsigned int*)nullptr);
    hipLaunchKernelGGL((select_pick_kernel<0>), dim3(c->E), blk, 0, c->stream, c->E, c->d_count, (const unsigned int*)hist, st);
    hipLaunchKernelGGL((select_pass_kernel<1>), grid, blk, 0, c->stream, c->d_sblock_off, c->E, c->d_count, c->d_cap_off, c->d_cd2, (const unsigned int*)nullptr,
                       hist, (const SelState*)st, c->d_sel_keys1, cnt1);
    hipLaunchKernelGGL((select_pick_kernel<1>), dim3(c->E), blk, 0, c->stream, c->E, c->d_count, (const unsigned int*)hist, st);
    hipLaunchKernelGGL((select_pass_kernel<2>), grid, blk, 0, c->stream, c->d_sblock_off, c->E, c->d_count, c->d_cap_off, c->d_sel_keys1, (const unsigned int*)cnt1,
                       hist, (const SelState*)st, c->d_sel_keys2, cnt2);
  }
  hipLaunchKernelGGL(select_final_kernel, dim3(c->E), dim3(NT), 0, c->stream, c->E, c->d_count, c->d_cap_off, c->d_sel_keys2, (const unsigned int*)cnt2,
                     (const unsigned int*)hist, (const SelState*)st, c->d_median, c->d_res_target ? c->d_res_target : c->d_res_host,
                     c->spec_arm ? c->d_a : (double*)nullptr, c->spec_arm ? c->d_a_check : (double*)nullptr, c->spec_arm ? 1.0 : 0.0);
  MV_HIP(hipGetLastError());
  return MVICP_OK;
}

}  // namespace mvicp
