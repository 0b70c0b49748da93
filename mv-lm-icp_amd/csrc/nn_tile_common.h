// Shared pieces of the two wave-cooperative exact 1-NN kernels (nn_tile.hip: fp32 VALU screen of LDS-staged tiles; nn_mfma.hip: the
// same traversal with the screen of an opened tile on the matrix pipe): target / job views, the query transform (frame.cpp:117-118,
// 131,136), the DPP wave reductions, the census reduction and the per-edge job table.  Everything here has internal linkage (one copy
// per translation unit, like the kernels that use it).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "common.h"
#include "nn_list.h"
#include "nn_tie.h"

namespace mvicp {

namespace {

constexpr int NT = 256;
#ifndef MVICP_TILE_LEAF
#define MVICP_TILE_LEAF 32
#endif
constexpr int LEAF = MVICP_TILE_LEAF;   // points per leaf tile (tuning builds may override; 32 measured best)
constexpr int FAN = 64;    // children per node = one box per lane


// matrix-pipe operands of one cloud (nn_mfma.hip, build_mfma): per tile 64 x 16 B — lane l's A fragment of v_mfma_f32_32x32x16_f16 for
// the tile's point l & 31 — and per block (= 64 tiles = one level-0 node) the origin / scale / error terms those fragments refer to
struct MfBlock { double cx, cy, cz, scale; float db, en, pad0, pad1; };

struct TileView {
  const double* spts; const int* sidx; int n;
  const float* wide; int levels;  // number of box levels (>= 1)
  int cnt[6]; long long off[6];
  double maxabs;                  // largest |coordinate| in the cloud
  const PointRec* srec;           // sorted 32-B records {x, y, z, original index}
  const uint4* mf_ops; const MfBlock* mf_blk;
};

struct TileJob {
  TileView dst;
  const double* q; const int* qidx; const double* xf; int n;
  int* out_idx; double* out_d2;
  const int* inv;   // target original index -> sorted position
  int seed;         // out_idx still holds last round's neighbours (sorted positions, -1 = none): use them as starting candidates
  float* out_lb;    // BND builds (fp32, rounded down): per query, a lower bound on the distance to every target other than out_idx (the grid kernel's temporal cache)
  float mu;         // BND builds: width of the extra guard band (metres) that makes that bound useful
  // BND builds, cache-aware rounds (round 3): out_lb holds last search's bounds and the edge's query transform carries the temporal-cache
  // allowance (xf[24] >= 0): a lane whose neighbour provably did not change sits the traversal out, like in nn_grid_kernel.  `list`
  // (list.dirty != null) = the edge's compacted list is maintained in place by this launch (nn_list.h).
  int cache;
  int reject_cache; // cache-aware rounds: a query that is provably still rejected by the cutoff (old neighbour and every other target beyond it) is a hit too
  int miss_max;     // cache-aware rounds of nn_tile_kernel: a wave with at most this many missed lanes answers them one by one (miss_block); 0 = off
  ListRef list;
  float kacc; int trig;   // nn_mfma.hip tunables (ctx::mfma_kacc, mfma_trig)
  TieRef tie;             // where lanes whose best distance was met by more than one target report (nn_tie.h)
};

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

// Wave-wide reductions on the DPP network (row quad-perm / mirror steps, then row_bcast15 / row_bcast31; lane 63 ends up with the
// result, broadcast through readlane -> SGPRs).  __shfl_xor would go through ds_bpermute: ~12 LDS-crossbar round trips per reduction,
// and this kernel reduces once per traversal step.
// fp32 reductions of NON-NEGATIVE values (incl. +inf): their bit patterns order like unsigned integers, so the whole
// reduction is six v_min_u32 / v_max_u32 with the DPP permutation fused into the operand (the builtin form costs a
// mov + mov_dpp + op per step).  "s_nop 1": a VALU result needs two wait states before a DPP read of it.
#define MVICP_DPP_CHAIN(OP)                                                         \
  "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
  "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
  "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"     \
  "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"          \
  "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"        \
  "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"        \
  "s_nop 1"
__device__ __forceinline__ float wave_max_f(float nonneg) {
  unsigned int v = (unsigned int)__float_as_int(nonneg);
  asm volatile(MVICP_DPP_CHAIN("v_max_u32_dpp") : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane((int)v, 63));
}
__device__ __forceinline__ float wave_min_f(float nonneg) {
  unsigned int v = (unsigned int)__float_as_int(nonneg);
  asm volatile(MVICP_DPP_CHAIN("v_min_u32_dpp") : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane((int)v, 63));
}
// fp32 reductions of ARBITRARY floats (negative coordinates, +-inf): the key  b ^ ((b >> 31) & 0x7fffffff)  orders like the float as a
// signed integer and is its own inverse, so the wave minimum / maximum is again six DPP-fused v_min_i32 / v_max_i32.
__device__ __forceinline__ int fkey(float f) { const int b = __float_as_int(f); return b ^ ((b >> 31) & 0x7fffffff); }
__device__ __forceinline__ float wave_min_any(float f) {
  int v = fkey(f);
  asm volatile(MVICP_DPP_CHAIN("v_min_i32_dpp") : "+v"(v));
  return __int_as_float(fkey(__int_as_float(__builtin_amdgcn_readlane(v, 63))));
}
__device__ __forceinline__ float wave_max_any(float f) {
  int v = fkey(f);
  asm volatile(MVICP_DPP_CHAIN("v_max_i32_dpp") : "+v"(v));
  return __int_as_float(fkey(__int_as_float(__builtin_amdgcn_readlane(v, 63))));
}
__device__ __forceinline__ float bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}


// screen threshold for a running best: (sqrt(best) + slack)^2 with 2^-20 relative head-room
__device__ __forceinline__ float thr_of(double best, float slack) {
  const float rb = (float)sqrt(best) * 1.000001f + slack;
  return rb * rb * 1.000002f;
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

TileView view_of(const FrameDev& f) {
  TileView v;
  v.spts = f.grid.spts; v.sidx = f.grid.sidx; v.n = f.n;
  v.wide = f.grid.wide; v.levels = f.grid.wide_levels;
  for (int l = 0; l < 6; ++l) { v.cnt[l] = f.grid.wide_cnt[l]; v.off[l] = f.grid.wide_off[l]; }
  v.maxabs = f.grid.maxabs;
  v.srec = (const PointRec*)f.grid.srec;
  v.mf_ops = (const uint4*)f.grid.mf_ops; v.mf_blk = (const MfBlock*)f.grid.mf_blk;
  return v;
}

// the per-edge job table of a launch (one TileJob per active edge of this rank)
inline int build_tile_jobs(mvicp_ctx* c, bool with_bounds, bool with_cache, bool with_list, std::vector<TileJob>& jobs, int& max_n, double& nq, std::vector<TieJob>& ties) {
  max_n = 0; nq = 0;
  double launch_q = 0;
  for (int e = 0; e < c->E; ++e) if (c->active[e]) launch_q += c->frames[c->esrc[e]].n;
  const TieRef tref = tie_ref(c, (size_t)launch_q, 0u);
  for (int e = 0; e < c->E; ++e) {
    if (!c->active[e]) continue;
    const FrameDev& s = c->frames[c->esrc[e]];
    const FrameDev& d = c->frames[c->edst[e]];
    if (!s.has_grid || !d.has_grid) { set_error("tile NN needs the per-cloud structure on frames %d and %d", c->esrc[e], c->edst[e]); return MVICP_ERR_STATE; }
    TileJob j;
    std::memset(&j, 0, sizeof(j));  // padding too: the table is cached by content
    j.dst = view_of(d);
    j.q = s.grid.spts; j.qidx = nullptr; j.xf = c->d_xf + (size_t)e * kEdgeXf; j.n = s.n;
    j.out_idx = c->d_nn_idx + c->cap_off[e]; j.out_d2 = c->d_nn_d2 + c->cap_off[e];
    j.inv = d.grid.inv;
    j.seed = (c->tile_seed && (int)c->nn_cache_edge.size() == c->E && c->nn_cache_edge[e]) ? 1 : 0;
    if (with_bounds) { j.out_lb = c->d_nn_lb + c->cap_off[e]; j.mu = (float)(c->tile_mu * d.grid.cell); }
    if (with_bounds && with_cache) { j.cache = 1; j.miss_max = c->tile_miss; j.reject_cache = c->reject_cache ? 1 : 0; }
    if (with_list) {
      j.list = ListRef{c->d_qpos + c->cap_off[e], c->d_second + c->cap_off[e], c->d_cd2 + c->cap_off[e], c->d_dirty + e, c->d_dirty_slots + c->dslot_off[e],
                       c->d_stream + c->cap_off[e], c->total_cap, d.grid.snor, (const PointRec*)d.grid.srec};
    }
    j.kacc = (float)c->mfma_kacc; j.trig = c->mfma_trig;
    j.tie = tref; j.tie.job = (unsigned int)jobs.size();
    {
      TieJob t;
      std::memset(&t, 0, sizeof(t));
      tie_job_fill(d, t);
      t.q = j.q; t.xf = j.xf; t.n = j.n; t.out_idx = j.out_idx; t.out_d2 = j.out_d2; t.inv = j.inv; t.list = j.list;
      ties.push_back(t);
    }
    jobs.push_back(j);
    max_n = std::max(max_n, s.n);
    nq += s.n;
  }
  return MVICP_OK;
}

// census counters -> pinned memory, asynchronously; census_resolve() (api.cpp) folds them in after the caller's own wait (no extra sync)
inline int census_collect(mvicp_ctx* c, unsigned long long* d_stats, size_t slots, double nq, const char* scope) {
  if (!c->h_census) MV_HIP(hipHostMalloc((void**)&c->h_census, 8 * sizeof(unsigned long long), hipHostMallocDefault));
  hipLaunchKernelGGL(census_sum_kernel, dim3(64), dim3(256), 0, c->stream, d_stats, slots, d_stats + 8 * slots);
  MV_HIP(hipMemcpyAsync(c->h_census, d_stats + 8 * slots, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  c->census_pending = true; c->census_nq = nq; c->census_kind = 2; c->census_scope = scope;
  return MVICP_OK;
}

// scratch for the per-wave census slots (profiling with nn_census only); null when the census is off
inline int census_scratch(mvicp_ctx* c, size_t slots, unsigned long long** d_stats) {
  *d_stats = nullptr;
  if (c->profile && c->nn_census) {
    const size_t need = sizeof(unsigned long long) * 8 * (slots + 1);
    if (need > c->census_bytes) {
      if (c->d_census) MV_HIP(hipFree(c->d_census));
      MV_HIP(hipMalloc((void**)&c->d_census, need));
      c->census_bytes = need;
    }
    *d_stats = (unsigned long long*)c->d_census;
    MV_HIP(hipMemsetAsync(*d_stats, 0, need, c->stream));
  }
  return MVICP_OK;
}

}  // namespace

}  // namespace mvicp
