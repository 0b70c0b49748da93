// K3/K4 — cutoff filter + stable compaction, packed-stream gather, exact median (radix select).
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
// LM evaluations re-read up to ~100 times: SoA px py pz qx qy qz nx ny nz (72 B per correspondence,
// coalesced) instead of 2 random 24-B gathers per evaluation.
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
    stream[0 * total_cap + o] = a0.x;
    stream[1 * total_cap + o] = a0.y;
    stream[2 * total_cap + o] = a1.x;
    stream[3 * total_cap + o] = b0.x;
    stream[4 * total_cap + o] = b0.y;
    stream[5 * total_cap + o] = b1.x;
    if (dn != nullptr) {
      stream[6 * total_cap + o] = dn[3 * s];
      stream[7 * total_cap + o] = dn[3 * s + 1];
      stream[8 * total_cap + o] = dn[3 * s + 2];
    }
  }
}

// ---- radix select (8 passes x 8 bits, MSB first) over the fp64 patterns of the accepted d2 ------------------------
// hist is [8 passes][E][256], zeroed once per call.  The digit pick of pass p+1 is recomputed by EVERY workgroup of
// pass p from that pass's finished histogram (256 bins, trivial) instead of a separate 1-block-per-edge launch per
// pass: 8 + 1 launches instead of 17, and no latency-bound pick kernels between the streaming passes.
struct SelState { unsigned long long prefix; int k; };

// One pick: given the finished histogram of digit `p` (restricted to keys matching st.prefix above it) choose the bin
// holding rank st.k.  Wave 0 scans 4 bins per lane + a wave prefix sum; result shared through LDS.
__device__ __forceinline__ SelState select_pick(const unsigned int* __restrict__ hist_pe, int p, SelState st, SelState* sh_state) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const uint4 h = reinterpret_cast<const uint4*>(hist_pe)[lane];
    const unsigned int tot = h.x + h.y + h.z + h.w;
    unsigned int inc = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned int o = __shfl_up(inc, d, 64);
      if (lane >= d) inc += o;
    }
    const unsigned int before = inc - tot;  // keys in bins below this lane's 4
    const unsigned int k = (unsigned int)st.k;
    const bool mine = before <= k && k < inc;
    // if no lane owns rank k (k >= total: cannot happen for k < count) the last bin is taken
    const unsigned long long owners = __ballot(mine);
    const int owner = owners ? __ffsll((long long)owners) - 1 : 63;
    if (lane == owner) {
      unsigned int cum = before;
      int bin = 4 * lane + 3;
      const unsigned int hv[4] = {h.x, h.y, h.z, h.w};
      unsigned int cb = before;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (cum + hv[i] > k) { bin = 4 * lane + i; cb = cum; break; }
        cum += hv[i];
        cb = cum;
      }
      if (!owners) cb = before + tot - hv[3];
      SelState o;
      o.prefix = st.prefix | ((unsigned long long)bin << (8 * p));
      o.k = (int)(k - cb);
      *sh_state = o;
    }
  }
  __syncthreads();
  return *sh_state;
}

__global__ __launch_bounds__(NT) void select_hist_kernel(const int* __restrict__ cblock_off, int E, const int* __restrict__ count,
                                                         const long long* __restrict__ cap_off, const double* __restrict__ cd2, int pass,
                                                         unsigned int* __restrict__ hist, SelState* __restrict__ state) {
  __shared__ unsigned int lh[256];
  __shared__ SelState sh_state;
  const int b = blockIdx.x;
  const int e = find_edge(cblock_off, E, b);
  const int lb = b - cblock_off[e];
  const int cnt = count[e];
  if (lb * kCompactBlock >= cnt) return;
  // state after digit pass+1: start of the chain for pass 7, otherwise one pick on top of the state stored by the previous launch
  SelState st;
  st.prefix = 0ull; st.k = cnt / 2;   // dists.begin() + size()/2  (frame.cpp:166)
  if (pass < 7) {
    if (pass < 6) st = state[(size_t)(pass + 2) * E + e];
    st = select_pick(hist + ((size_t)(pass + 1) * E + e) * 256, pass + 1, st, &sh_state);
    if (lb == 0 && threadIdx.x == 0) state[(size_t)(pass + 1) * E + e] = st;
  }
  lh[threadIdx.x] = 0u;
  __syncthreads();
  const long long base = cap_off[e];
  const int shift = 8 * pass;
  for (int i = 0; i < IPT; ++i) {
    const int pos = lb * kCompactBlock + i * NT + threadIdx.x;
    if (pos < cnt) {
      const unsigned long long key = (unsigned long long)__double_as_longlong(cd2[base + pos]);
      const bool match = (pass == 7) || ((key >> (shift + 8)) == (st.prefix >> (shift + 8)));
      if (match) atomicAdd(&lh[(key >> shift) & 255ull], 1u);
    }
  }
  __syncthreads();
  const unsigned int v = lh[threadIdx.x];
  if (v) atomicAdd(&hist[((size_t)pass * E + e) * 256 + threadIdx.x], v);
}

__global__ __launch_bounds__(256) void select_final_kernel(const unsigned int* __restrict__ hist, int E, const int* __restrict__ count,
                                                           const SelState* __restrict__ state, double* __restrict__ median) {
  __shared__ SelState sh_state;
  const int e = blockIdx.x;
  const int cnt = count[e];
  if (cnt <= 0) { if (threadIdx.x == 0) median[e] = 0.0; return; }
  SelState st = state[(size_t)1 * E + e];
  st = select_pick(hist + ((size_t)0 * E + e) * 256, 0, st, &sh_state);
  if (threadIdx.x == 0) median[e] = __longlong_as_double((long long)st.prefix);
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

int launch_select_median(mvicp_ctx* c) {
  if (c->n_cblocks == 0 || c->E == 0) return MVICP_OK;
  double bytes = 0;
  for (int e = 0; e < c->E; ++e) if (c->owned[e]) bytes += 64.0 * c->h_count[e];
  ProfScope ps(c, "select", bytes);
  MV_HIP(hipMemsetAsync(c->d_sel_hist, 0, sizeof(unsigned int) * 8 * (size_t)c->E * 256, c->stream));
  for (int pass = 7; pass >= 0; --pass)
    hipLaunchKernelGGL(select_hist_kernel, dim3(c->n_cblocks), dim3(NT), 0, c->stream, c->d_cblock_off, c->E, c->d_count, c->d_cap_off, c->d_cd2,
                       pass, c->d_sel_hist, (SelState*)c->d_sel_state);
  hipLaunchKernelGGL(select_final_kernel, dim3(c->E), dim3(256), 0, c->stream, c->d_sel_hist, c->E, c->d_count, (const SelState*)c->d_sel_state,
                     c->d_median);
  MV_HIP(hipGetLastError());
  return MVICP_OK;
}

}  // namespace mvicp
