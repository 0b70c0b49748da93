// K5 — the correspondence lists in the REFERENCE's layout, for all edges at once.
//
// The reference fills Frame::neighbours[j].correspondances with {first = source index k, second = target index, dist = sqrt(d2)} in
// ascending k (src/internal/frame.cpp:129,156-160; struct Correspondance include/frame.h:18-22: int, int, double = 16 B).  The device
// lists (corr.hip) hold SORTED positions in the source cloud's sorted order — the layout every kernel downstream wants — so the drop-in
// copy-back needs an un-sort.  Rounds 1-4 did it on the host, per edge: three pageable hipMemcpy, an index remap and a std::sort per edge
// per round (62 x per cfg4 round).  Here it is one device pass over all owned edges and ONE asynchronous copy into pinned memory:
//
//   export_count_kernel    per 1024 ORIGINAL source indices k: how many are accepted  (qpos[inv_src[k]] >= 0)
//   export_scan_kernel     per edge: exclusive scan of its block counts (the edge's output offset is known to the host: the counts of
//                          the search are already there)
//   export_scatter_kernel  k ascending -> triple {k, sidx_dst[second[p]], sqrt(cd2[p])} at its rank: 16-B stores, consecutive per wave
//
// sqrt is the correctly rounded IEEE one (__dsqrt_rn), the same value the host's std::sqrt gave in rounds 1-4 (frame.cpp:139 pointDist =
// sqrt(pointDistSquared)); tests/test_gpu_parity.py compares both paths bit for bit.  Algorithmic bytes: 8 B per query (inv, qpos) + 32 B
// per correspondence (second, cd2, sidx_dst gather, 16-B triple) on the device, 16 B per correspondence over PCIe.
#include <algorithm>
#include <cstring>

#include "common.h"

namespace mvicp {

namespace {

constexpr int NT = 256;
constexpr int IPT = kCompactBlock / NT;

struct Triple { int first; int second; double dist; };
static_assert(sizeof(Triple) == 16 && sizeof(mvicp_corr) == 16, "Correspondance is 16 bytes (include/frame.h:18-22)");

__device__ __forceinline__ int find_edge(const int* __restrict__ off, int E, int b) {
  int lo = 0, hi = E;  // largest e with off[e] <= b
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= b) lo = mid; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ int block_exclusive_scan(int v, int* __restrict__ wave_tot, int* total) {
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

// nexp[e] = number of source points of an exportable edge (owned, searched, list built by the search), else 0
__global__ __launch_bounds__(NT) void export_count_kernel(const int* __restrict__ cblock_off, int E, const int* __restrict__ nexp,
                                                          const long long* __restrict__ cap_off, const int* const* __restrict__ src_inv,
                                                          const int* __restrict__ qpos, int* __restrict__ xblock_cnt) {
  __shared__ int wave_tot[NT / 64];
  const int b = blockIdx.x;
  const int e = find_edge(cblock_off, E, b);
  const int n = nexp[e];
  const int lb = b - cblock_off[e];
  if (lb * kCompactBlock >= n) { if (threadIdx.x == 0) xblock_cnt[b] = 0; return; }
  const long long base = cap_off[e];
  const int* __restrict__ inv = src_inv[e];
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int k = lb * kCompactBlock + i * NT + threadIdx.x;   // (strided: coalesced inv loads; only the block total matters here)
    if (k < n) cnt += qpos[base + inv[k]] >= 0 ? 1 : 0;
  }
  int total;
  block_exclusive_scan(cnt, wave_tot, &total);
  if (threadIdx.x == 0) xblock_cnt[b] = total;
}

__global__ __launch_bounds__(NT) void export_scan_kernel(const int* __restrict__ cblock_off, int* __restrict__ xblock_cnt) {
  __shared__ int wave_tot[NT / 64];
  __shared__ int carry_s;
  const int e = blockIdx.x;
  const int b0 = cblock_off[e], b1 = cblock_off[e + 1];
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int s = b0; s < b1; s += NT) {
    const int b = s + threadIdx.x;
    const int v = (b < b1) ? xblock_cnt[b] : 0;
    int total;
    const int ex = block_exclusive_scan(v, wave_tot, &total);
    const int carry = carry_s;
    if (b < b1) xblock_cnt[b] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = carry + total;
    __syncthreads();
  }
}

__global__ __launch_bounds__(NT) void export_scatter_kernel(const int* __restrict__ cblock_off, int E, const int* __restrict__ nexp,
                                                            const long long* __restrict__ cap_off, const int* const* __restrict__ src_inv,
                                                            const int* const* __restrict__ dst_sidx, const int* __restrict__ qpos,
                                                            const int* __restrict__ second, const double* __restrict__ cd2,
                                                            const int* __restrict__ xblock_cnt, const long long* __restrict__ out_off,
                                                            Triple* __restrict__ out) {
  __shared__ int wave_tot[NT / 64];
  const int b = blockIdx.x;
  const int e = find_edge(cblock_off, E, b);
  const int n = nexp[e];
  const int lb = b - cblock_off[e];
  if (lb * kCompactBlock >= n) return;
  const long long base = cap_off[e];
  const int* __restrict__ inv = src_inv[e];
  const int* __restrict__ sidx = dst_sidx[e];
  int p[IPT];
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int k = lb * kCompactBlock + threadIdx.x * IPT + i;   // (blocked: IPT consecutive k per thread keep the output ascending)
    p[i] = k < n ? qpos[base + inv[k]] : -1;
    cnt += p[i] >= 0 ? 1 : 0;
  }
  int total;
  long long pos = out_off[e] + xblock_cnt[b] + block_exclusive_scan(cnt, wave_tot, &total);
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    if (p[i] < 0) continue;
    Triple t;
    t.first = lb * kCompactBlock + threadIdx.x * IPT + i;
    t.second = sidx[second[base + p[i]]];
    t.dist = __dsqrt_rn(cd2[base + p[i]]);   // frame.cpp:139 (IEEE, correctly rounded)
    out[pos++] = t;
  }
}

}  // namespace

// Un-sorts every exportable edge's list into c->d_export and queues the copy into c->h_export (pinned); fills c->export_off (E + 1,
// in triples; edges that are not exportable have zero width).  The caller waits for the stream.
int launch_export(mvicp_ctx* c) {
  const int E = c->E;
  c->export_off.assign((size_t)E + 1, 0);
  std::vector<int> nexp(E, 0);
  long long total = 0;
  for (int e = 0; e < E; ++e) {
    const bool ok = c->owned[e] && !c->explicit_list[e] && c->qpos_valid[e] && c->h_count[e] > 0;   // (qpos_valid, not list_valid: mvicp_recompute_normals clears the latter — ADVICE r5)
    nexp[e] = ok ? c->frames[c->esrc[e]].n : 0;
    c->export_off[e + 1] = c->export_off[e] + (ok ? c->h_count[e] : 0);
  }
  total = c->export_off[E];
  c->export_chunks = 0; c->export_in_flight = false;
  if (total == 0 || c->n_cblocks == 0) return MVICP_OK;
  if ((size_t)total > c->export_cap) {
    MV_HIP(hipStreamSynchronize(c->stream));
    if (c->d_export) MV_HIP(hipFree(c->d_export));
    if (c->h_export) MV_HIP(hipHostFree(c->h_export));
    c->d_export = nullptr; c->h_export = nullptr; c->export_cap = 0;
    // sized for the worst case of this graph (every query of every owned edge accepted), so the buffers are allocated once per graph
    const size_t cap = std::max<size_t>((size_t)c->total_cap, (size_t)total);
    MV_HIP(hipMalloc(&c->d_export, cap * sizeof(Triple)));
    MV_HIP(hipHostMalloc(&c->h_export, cap * sizeof(Triple), hipHostMallocDefault));
    c->export_cap = cap;
  }
  if (!c->d_xblock_cnt) MV_HIP(hipMalloc((void**)&c->d_xblock_cnt, sizeof(int) * (size_t)std::max(c->n_cblocks, 1)));
  // small tables: per-edge point counts | output offsets | inv / sidx pointers (cached by content: re-uploaded only when they change)
  std::vector<char> tab(sizeof(int) * (size_t)E + sizeof(long long) * ((size_t)E + 1) + 2 * sizeof(void*) * (size_t)E + 16, 0);
  const size_t o_off = (sizeof(int) * (size_t)E + 7) & ~(size_t)7, o_ptr = o_off + sizeof(long long) * ((size_t)E + 1);
  std::memcpy(tab.data(), nexp.data(), sizeof(int) * (size_t)E);
  std::memcpy(tab.data() + o_off, c->export_off.data(), sizeof(long long) * ((size_t)E + 1));
  const void** ptrs = reinterpret_cast<const void**>(tab.data() + o_ptr);
  for (int e = 0; e < E; ++e) { ptrs[e] = c->frames[c->esrc[e]].grid.inv; ptrs[E + e] = c->frames[c->edst[e]].grid.sidx; }
  char* d_tab = nullptr;
  MV_CHECK(cached_upload(c, "export_tab", tab.data(), o_ptr + 2 * sizeof(void*) * (size_t)E, (void**)&d_tab));
  const int* d_nexp = reinterpret_cast<const int*>(d_tab);
  const long long* d_off = reinterpret_cast<const long long*>(d_tab + o_off);
  const int* const* d_inv = reinterpret_cast<const int* const*>(d_tab + o_ptr);
  const int* const* d_sidx = d_inv + E;
  {
    ProfScope ps(c, "export", 8.0 * (double)c->total_cap + 32.0 * (double)total);
    hipLaunchKernelGGL(export_count_kernel, dim3(c->n_cblocks), dim3(NT), 0, c->stream, c->d_cblock_off, E, d_nexp, c->d_cap_off, d_inv, c->d_qpos, c->d_xblock_cnt);
    hipLaunchKernelGGL(export_scan_kernel, dim3(E), dim3(NT), 0, c->stream, c->d_cblock_off, c->d_xblock_cnt);
    hipLaunchKernelGGL(export_scatter_kernel, dim3(c->n_cblocks), dim3(NT), 0, c->stream, c->d_cblock_off, E, d_nexp, c->d_cap_off, d_inv, d_sidx, c->d_qpos,
                       c->d_second, c->d_cd2, c->d_xblock_cnt, d_off, (Triple*)c->d_export);
  }
  MV_HIP(hipGetLastError());
  // device -> pinned host in chunks (runs of edges with one source frame), an event behind each: mvicp_wait_correspondences(edge) waits for the
  // chunk that holds the edge only.  The chunks are queued back to back on the library's stream, so the bus never idles between them.
  c->export_edge_chunk.assign(E, 0);
  int chunks = 0;
  for (int e = 0; e < E;) {
    int e1 = e + 1;
    while (e1 < E && c->esrc[e1] == c->esrc[e]) ++e1;
    const long long a = c->export_off[e], b = c->export_off[e1];
    for (int k = e; k < e1; ++k) c->export_edge_chunk[k] = chunks;
    if ((int)c->export_events.size() <= chunks) { hipEvent_t ev = nullptr; MV_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming)); c->export_events.push_back(ev); }
    if (b > a) MV_HIP(hipMemcpyAsync((Triple*)c->h_export + a, (const Triple*)c->d_export + a, (size_t)(b - a) * sizeof(Triple), hipMemcpyDeviceToHost, c->stream));
    MV_HIP(hipEventRecord(c->export_events[chunks], c->stream));
    ++chunks;
    e = e1;
  }
  c->export_chunks = chunks;
  c->export_in_flight = true;
  return MVICP_OK;
}

}  // namespace mvicp
