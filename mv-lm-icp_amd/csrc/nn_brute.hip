// K1 — exhaustive exact 1-NN (the parity kernel, and the small-N / no-structure path).
//
// Replaces the KD-tree descent of Frame::getClosestPoint (src/internal/frame.cpp:187-206 ->
// include/nanoflann.hpp:900-911,1199-1247) and the per-query transform of
// Frame::computeClosestPointsToNeighbours (frame.cpp:117-118,131,136).  Bit-exact contract: the
// distance is the expression of include/frame.h:70-76 evaluated in fp64, left to right, WITHOUT fma
// contraction (this TU is built with -ffp-contract=off and uses __dmul_rn/__dadd_rn/__dsub_rn), and
// the winner is the lowest index among equal distances (strict '<' over an ascending scan); a query whose best distance was met more
// than once is REPORTED (nn_tie.h) and re-answered the way the reference's tree decides such ties.
//
// Mapping (wave64, gfx950): one thread owns QPT queries in registers; the target cloud streams through
// LDS in SoA tiles of TS points (3 x TS x 8 B); every lane reads the same LDS address per candidate
// (broadcast, conflict-free), so one ds_read feeds 64 x QPT distance evaluations.  When queries x edges
// alone cannot fill 256 CUs the target range is split across gridDim.y and merged by a second kernel
// (ascending split order + strict '<' keeps the lowest-index rule).
#include "common.h"
#include <cstring>

#include "nn_tie.h"

namespace mvicp {

namespace {

constexpr int QPT = 4;
constexpr int TS = 512;
constexpr int NT = 256;

struct BruteJob {
  const double* q;      // queries (n x 3) in SOURCE coordinates if xf != null, else already local to the target
  const double* xf;     // kEdgeXf doubles or null
  const double* tgt;    // m x 3
  int n, m;
  int* out_idx; double* out_d2;  // final outputs (n)
  const int* inv;                // target original index -> sorted position (null: emit original indices)
  TieRef tie;
};

__device__ __forceinline__ void xf_point(const double* __restrict__ x, double p0, double p1, double p2, double& q0, double& q1, double& q2) {
  // g_i = ((R(i,0) p0 + R(i,1) p1) + R(i,2) p2) + t_i ; u = g - t_d ; q_i = (Ri(i,0) u0 + Ri(i,1) u1) + Ri(i,2) u2
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

__global__ __launch_bounds__(NT) void nn_brute_kernel(const BruteJob* __restrict__ jobs, int n_splits, int* __restrict__ split_idx,
                                                      double* __restrict__ split_d2, const long long* __restrict__ split_off) {
  const BruteJob job = jobs[blockIdx.z];
  const int qbase = blockIdx.x * (NT * QPT);
  if (qbase >= job.n) return;
  __shared__ double sx[TS], sy[TS], sz[TS];
  __shared__ double sxf[kEdgeXf];
  const int tid = threadIdx.x;
  if (job.xf != nullptr && tid < kEdgeXf) sxf[tid] = job.xf[tid];
  __syncthreads();

  double qx[QPT], qy[QPT], qz[QPT], best[QPT];
  int bi[QPT];
  bool tie[QPT];   // the running best distance was met by a second target
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    const int k = qbase + i * NT + tid;
    best[i] = 1.7976931348623157e308;
    bi[i] = -1; tie[i] = false;
    if (k < job.n) {
      const double p0 = job.q[3 * (size_t)k], p1 = job.q[3 * (size_t)k + 1], p2 = job.q[3 * (size_t)k + 2];
      if (job.xf != nullptr) xf_point(sxf, p0, p1, p2, qx[i], qy[i], qz[i]);
      else { qx[i] = p0; qy[i] = p1; qz[i] = p2; }
    } else {
      qx[i] = qy[i] = qz[i] = 0.0;
    }
  }

  // target range of this split, tile aligned
  const int tiles = (job.m + TS - 1) / TS;
  const int t0 = (int)(((long long)tiles * blockIdx.y) / n_splits);
  const int t1 = (int)(((long long)tiles * (blockIdx.y + 1)) / n_splits);
  const double inf = __longlong_as_double(0x7ff0000000000000LL);
  for (int t = t0; t < t1; ++t) {
    const int base = t * TS;
    __syncthreads();
    // stage: TS points = 3*TS contiguous doubles, coalesced; scatter to SoA
    for (int w = tid; w < 3 * TS; w += NT) {
      const size_t gidx = 3 * (size_t)base + w;
      const double v = (gidx < 3 * (size_t)job.m) ? job.tgt[gidx] : inf;
      const int pnt = w / 3, comp = w - 3 * pnt;
      if (comp == 0) sx[pnt] = v; else if (comp == 1) sy[pnt] = v; else sz[pnt] = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < TS; ++j) {
      const double x = sx[j], y = sy[j], z = sz[j];
#pragma unroll
      for (int i = 0; i < QPT; ++i) {
        const double d0 = __dsub_rn(qx[i], x), d1 = __dsub_rn(qy[i], y), d2 = __dsub_rn(qz[i], z);
        const double d = __dadd_rn(__dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1)), __dmul_rn(d2, d2));
        if (d < best[i]) { best[i] = d; bi[i] = base + j; tie[i] = false; }
        else if (d == best[i]) tie[i] = true;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < QPT; ++i) {
    const int k = qbase + i * NT + tid;
    if (k >= job.n) continue;
    if (n_splits == 1) {
      job.out_idx[k] = (bi[i] >= 0 && job.inv) ? job.inv[bi[i]] : bi[i];
      job.out_d2[k] = best[i];
      if (tie[i] && bi[i] >= 0) tie_report(job.tie, (unsigned int)k);
    } else {
      const size_t o = (size_t)split_off[blockIdx.z] + (size_t)blockIdx.y * job.n + k;
      split_idx[o] = bi[i] < 0 ? -1 : (bi[i] | (tie[i] ? 0x40000000 : 0));   // (bit 30: tie inside this split; clouds hold < 2^30 points)
      split_d2[o] = best[i];
    }
  }
}

__global__ __launch_bounds__(NT) void nn_brute_merge_kernel(const BruteJob* __restrict__ jobs, int n_splits, const int* __restrict__ split_idx,
                                                            const double* __restrict__ split_d2, const long long* __restrict__ split_off) {
  const BruteJob job = jobs[blockIdx.y];
  const int k = blockIdx.x * NT + threadIdx.x;
  if (k >= job.n) return;
  double best = 1.7976931348623157e308;
  int bi = -1;
  bool tie = false;
  for (int s = 0; s < n_splits; ++s) {
    const size_t o = (size_t)split_off[blockIdx.y] + (size_t)s * job.n + k;
    const double d = split_d2[o];
    const int i = split_idx[o];
    if (i < 0) continue;
    if (d < best) { best = d; bi = i & 0x3fffffff; tie = (i & 0x40000000) != 0; }
    else if (d == best) tie = true;
  }
  job.out_idx[k] = (bi >= 0 && job.inv) ? job.inv[bi] : bi;
  job.out_d2[k] = best;
  if (tie && bi >= 0) tie_report(job.tie, (unsigned int)k);
}

int run_jobs(mvicp_ctx* c, std::vector<BruteJob>& jobs, double* pairs_out, const FrameDev* const* dst_of) {
  if (jobs.empty()) return MVICP_OK;
  std::vector<TieJob> ties;
  {
    double launch_q = 0;
    for (const BruteJob& j : jobs) launch_q += j.n;
    const TieRef tref = tie_ref(c, (size_t)launch_q, 0u);
    for (size_t k = 0; k < jobs.size(); ++k) {
      jobs[k].tie = tref; jobs[k].tie.job = (unsigned int)k;
      TieJob t;
      std::memset(&t, 0, sizeof(t));
      tie_job_fill(*dst_of[k], t);
      t.q = jobs[k].q; t.xf = jobs[k].xf; t.n = jobs[k].n; t.out_idx = jobs[k].out_idx; t.out_d2 = jobs[k].out_d2; t.inv = jobs[k].inv;
      ties.push_back(t);
    }
  }
  int max_n = 0;
  double pairs = 0;
  for (const BruteJob& j : jobs) { max_n = std::max(max_n, j.n); pairs += (double)j.n * j.m; }
  if (max_n == 0) return MVICP_OK;
  const int qblocks = (max_n + NT * QPT - 1) / (NT * QPT);
  long long total_blocks = 0;
  for (const BruteJob& j : jobs) total_blocks += (j.n + NT * QPT - 1) / (NT * QPT);
  int splits = 1;
  if (total_blocks < 2048) splits = (int)std::min<long long>(16, (2048 + total_blocks - 1) / total_blocks);
  // staging for job table (+ split offsets)
  BruteJob* d_jobs = nullptr;
  long long* d_off = nullptr;
  scratch_reset(c);
  MV_CHECK(scratch_upload(c, jobs.data(), sizeof(BruteJob) * jobs.size(), (void**)&d_jobs));
  std::vector<long long> off(jobs.size() + 1, 0);
  if (splits > 1) {
    for (size_t i = 0; i < jobs.size(); ++i) off[i + 1] = off[i] + (long long)splits * jobs[i].n;
    const size_t need = (size_t)off.back();
    if (need > c->split_cap) {
      if (c->d_split_idx) { MV_HIP(hipFree(c->d_split_idx)); MV_HIP(hipFree(c->d_split_d2)); }
      MV_HIP(hipMalloc((void**)&c->d_split_idx, sizeof(int) * need));
      MV_HIP(hipMalloc((void**)&c->d_split_d2, sizeof(double) * need));
      c->split_cap = need;
    }
    MV_CHECK(scratch_upload(c, off.data(), sizeof(long long) * off.size(), (void**)&d_off));
  }
  {
    ProfScope ps(c, "nn_brute", pairs * 24.0);
    hipLaunchKernelGGL(nn_brute_kernel, dim3(qblocks, splits, (unsigned)jobs.size()), dim3(NT), 0, c->stream, d_jobs, splits,
                       c->d_split_idx, c->d_split_d2, d_off);
    if (splits > 1)
      hipLaunchKernelGGL(nn_brute_merge_kernel, dim3((max_n + NT - 1) / NT, (unsigned)jobs.size()), dim3(NT), 0, c->stream, d_jobs, splits,
                         c->d_split_idx, c->d_split_d2, d_off);
  }
  MV_HIP(hipGetLastError());
  if (pairs_out) *pairs_out = pairs;
  MV_CHECK(launch_tie_fixup(c, ties, 1.7976931348623157e308));   // (the brute path maintains no list)
  return MVICP_OK;
}

}  // namespace

int launch_nn_brute_edges(mvicp_ctx* c) {
  std::vector<BruteJob> jobs;
  std::vector<const FrameDev*> dsts;
  for (int e = 0; e < c->E; ++e) {
    if (!c->active[e]) continue;
    const FrameDev& s = c->frames[c->esrc[e]];
    const FrameDev& d = c->frames[c->edst[e]];
    BruteJob j;
    // queries in the source's SORTED order (the pipeline's order), targets scanned in original order (lowest original index wins ties)
    j.q = s.grid.spts; j.xf = c->d_xf + (size_t)e * kEdgeXf; j.tgt = d.pts; j.n = s.n; j.m = d.n;
    j.inv = d.grid.inv;
    j.out_idx = c->d_nn_idx + c->cap_off[e]; j.out_d2 = c->d_nn_d2 + c->cap_off[e];
    jobs.push_back(j); dsts.push_back(&d);
  }
  return run_jobs(c, jobs, nullptr, dsts.data());
}

int launch_nn_brute_queries(mvicp_ctx* c, const FrameDev& f, const double* d_q, int n, int* d_idx, double* d_d2) {
  std::vector<BruteJob> jobs(1);
  jobs[0].q = d_q; jobs[0].xf = nullptr; jobs[0].tgt = f.pts; jobs[0].n = n; jobs[0].m = f.n;
  jobs[0].out_idx = d_idx; jobs[0].out_d2 = d_d2; jobs[0].inv = nullptr;
  const FrameDev* dst = &f;
  return run_jobs(c, jobs, nullptr, &dst);
}

}  // namespace mvicp
