// f1 — Frame::recomputeNormals (src/internal/frame.cpp:244-255): for every point the k = 10 nearest points of its own
// cloud INCLUDING itself (Frame::getNeighbours, frame.cpp:208-231 -> nanoflann knnSearch), then pointSetPCA
// (include/common.h:331-346): centroid, covariance of the centred neighbours, eigenvector of the smallest eigenvalue,
// flipped so that n_z <= 0 (towards the camera).
//
// k-NN through the spatial hash built at upload: scan the (2r+1)^3 cell block around the point's own cell keeping the
// k best (d2, sorted position) in registers; the set is provably exact once the k-th distance is below the distance
// to the block faces (>= r h); r grows until that holds (r = 1 almost always: cells hold ~6 points).  Distances use
// the reference metric (include/frame.h:70-76, no fma).
// EXACT TIES.  The reference's clouds are range-image lattices (z quantised to 1 mm): 7 % of the Bunny points have a tie at the 10th
// place, and which of the tied points nanoflann returns is decided by the order in which its KD-tree visits them
// (KNNResultSet::addPoint keeps what came first, nanoflann.hpp:75-134; searchLevel visits the near child first, :1199-1247).  A
// "lowest index" rule picks a different neighbour set for 2-4 % of the points (normals up to 20 degrees apart, final Bunny poses 4.5e-5
// apart after 20 rounds).  So equal distances are ordered the way that tree would visit them: VisitTree below is the split structure
// nanoflann builds for this cloud (leaf size 1, frame.cpp:189), restated from the published algorithm, and two tied points are ordered by
// their lowest common ancestor — the child on the query's side of the split plane is visited first.  Result lists then equal
// knnSearch's element for element (order included: equal distances appear in visit order there too).
// The 3x3 symmetric eigenproblem is solved by cyclic Jacobi rotations in fp64 (Eigen's SelfAdjointEigenSolver is an
// iterative QR on the same matrix: the eigenvector agrees to rounding, not bit for bit).
#include <algorithm>
#include <vector>

#include "common.h"
#include "kdvisit.h"
#include "nn_tie.h"

namespace mvicp {

namespace {

constexpr int NT = 128;     // per-thread candidate lists live in LDS: KMAX x NT x 20 B = 40 KB per workgroup
constexpr int KMAX = 16;
constexpr unsigned long long EMPTY = ~0ull;

struct HashEntry { unsigned long long key; unsigned int start, count; };

__device__ __forceinline__ unsigned long long cell_key(int ix, int iy, int iz) {
  return (unsigned long long)ix | ((unsigned long long)iy << 21) | ((unsigned long long)iz << 42);
}
__device__ __forceinline__ unsigned int hash_slot(unsigned long long k, int shift) { return (unsigned int)((k * 0x9E3779B97F4A7C15ull) >> shift); }

struct NormJob {
  VisitTree tie;
  const PointRec* srec; int n;   // records in hash-CELL order (GridDev::crec)
  const int* inv;                // original index -> sorted (canonical) position
  const HashEntry* table; unsigned int mask; int shift;
  double ox, oy, oz, h, inv_h;
  int dx, dy, dz;
  int k;
  double* nor_out;   // n x 3, original order
  double* snor_out;  // n x 3, sorted order (what the gather kernel reads)
  int* knn_out;      // n x k original indices (optional, for tests), ascending distance, equal distances in the tree's visit order
};

__device__ __forceinline__ void jacobi_min_eigvec(double a00, double a01, double a02, double a11, double a12, double a22, double* v) {
  double A[3][3] = {{a00, a01, a02}, {a01, a11, a12}, {a02, a12, a22}};
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
    const double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
    if (off <= 1e-18 * diag || off == 0.0) break;
#pragma unroll
    for (int pq = 0; pq < 3; ++pq) {
      const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
      if (A[p][q] == 0.0) continue;
      const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
      const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
      for (int r = 0; r < 3; ++r) {  // A <- A J
        const double arp = A[r][p], arq = A[r][q];
        A[r][p] = c * arp - s * arq; A[r][q] = s * arp + c * arq;
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {  // A <- J^T A
        const double apr = A[p][r], aqr = A[q][r];
        A[p][r] = c * apr - s * aqr; A[q][r] = s * apr + c * aqr;
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const double vrp = V[r][p], vrq = V[r][q];
        V[r][p] = c * vrp - s * vrq; V[r][q] = s * vrp + c * vrq;
      }
    }
  }
  int m = 0;
  if (A[1][1] < A[m][m]) m = 1;
  if (A[2][2] < A[m][m]) m = 2;
  double nx = V[0][m], ny = V[1][m], nz = V[2][m];
  const double nn = sqrt(nx * nx + ny * ny + nz * nz);
  nx /= nn; ny /= nn; nz /= nn;
  if (nz > 0) { nx = -nx; ny = -ny; nz = -nz; }  // common.h:343 `if (normal(2) > 0) normal = -normal`
  v[0] = nx; v[1] = ny; v[2] = nz;
}

__global__ __launch_bounds__(NT) void normals_kernel(NormJob job) {
  const int i = blockIdx.x * NT + threadIdx.x;  // position in cell order
  if (i >= job.n) return;
  const PointRec me = job.srec[i];
  const int K = job.k;
  // the thread's current k best, ascending: distance, cell-order position, original index.  In LDS (one column per thread, no bank
  // conflicts) rather than in register arrays: the insertion below indexes them at run time, and this kernel runs once per cloud
  __shared__ double s_bd[KMAX][NT];
  __shared__ int s_bj[KMAX][NT];
  __shared__ long long s_bo[KMAX][NT];
#define bd(t) s_bd[t][threadIdx.x]
#define bj(t) s_bj[t][threadIdx.x]
#define bo(t) s_bo[t][threadIdx.x]
  const int cx = min(max((int)floor((me.x - job.ox) * job.inv_h), 0), job.dx - 1);
  const int cy = min(max((int)floor((me.y - job.oy) * job.inv_h), 0), job.dy - 1);
  const int cz = min(max((int)floor((me.z - job.oz) * job.inv_h), 0), job.dz - 1);
  const int rmax = max(job.dx, max(job.dy, job.dz));
  for (int r = 1;; ++r) {
#pragma unroll
    for (int t = 0; t < KMAX; ++t) { bd(t) = 1.7976931348623157e308; bj(t) = -1; bo(t) = 0x7fffffffffffffffLL; }
    for (int iz = max(cz - r, 0); iz <= min(cz + r, job.dz - 1); ++iz)
      for (int iy = max(cy - r, 0); iy <= min(cy + r, job.dy - 1); ++iy)
        for (int ix = max(cx - r, 0); ix <= min(cx + r, job.dx - 1); ++ix) {
          const unsigned long long key = cell_key(ix, iy, iz);
          unsigned int slot = hash_slot(key, job.shift) & job.mask;
          HashEntry e = job.table[slot];
          while (e.key != key && e.key != EMPTY) { slot = (slot + 1) & job.mask; e = job.table[slot]; }
          if (e.key == EMPTY) continue;
          for (unsigned int j = e.start; j < e.start + e.count; ++j) {
            const PointRec p = job.srec[j];
            const double d0 = __dsub_rn(me.x, p.x), d1 = __dsub_rn(me.y, p.y), d2 = __dsub_rn(me.z, p.z);
            const double d = __dadd_rn(__dadd_rn(__dmul_rn(d0, d0), __dmul_rn(d1, d1)), __dmul_rn(d2, d2));
            // sorted insertion (ascending distance; equal distances in the tree's visit order).  One comparator call site, plain indexed
            // arrays: this kernel runs once per cloud, simplicity beats register residency here.
            int pos = K;
            while (pos > 0) {
              const double ed = bd(pos - 1);
              bool before = d < ed;
              if (!before && d == ed) before = bj(pos - 1) < 0 || visited_before(job.tie, me.x, me.y, me.z, p.idx, bo(pos - 1));
              if (!before) break;
              --pos;
            }
            if (pos < K) {
              for (int t = K - 1; t > pos; --t) { bd(t) = bd(t - 1); bj(t) = bj(t - 1); bo(t) = bo(t - 1); }
              bd(pos) = d; bj(pos) = (int)j; bo(pos) = p.idx;
            }
          }
        }
    // exact iff the K-th distance is inside the scanned block: every unscanned point is >= m away along some axis
    const double fx = job.ox + (cx - r) * job.h, fy = job.oy + (cy - r) * job.h, fz = job.oz + (cz - r) * job.h;
    const double w = (2 * r + 1) * job.h;
    double m = fmin(fmin(me.x - fx, fx + w - me.x), fmin(fmin(me.y - fy, fy + w - me.y), fmin(me.z - fz, fz + w - me.z)));
    m *= 0.999;
    const bool full = bj(K - 1) >= 0;
    if ((full && m > 0.0 && bd(K - 1) < m * m) || r >= rmax) break;
  }
  // PCA of the neighbours (common.h:331-346)
  double mx = 0, my = 0, mz = 0;
  int kk = 0;
#pragma unroll
  for (int t = 0; t < KMAX; ++t)
    if (t < K && bj(t) >= 0) { const PointRec p = job.srec[bj(t)]; mx += p.x; my += p.y; mz += p.z; ++kk; }
  mx /= kk; my /= kk; mz /= kk;
  double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
#pragma unroll
  for (int t = 0; t < KMAX; ++t)
    if (t < K && bj(t) >= 0) {
      const PointRec p = job.srec[bj(t)];
      const double x = p.x - mx, y = p.y - my, z = p.z - mz;
      c00 += x * x; c01 += x * y; c02 += x * z; c11 += y * y; c12 += y * z; c22 += z * z;
    }
  double nv[3];
  jacobi_min_eigvec(c00, c01, c02, c11, c12, c22, nv);
  double* o = job.nor_out + 3 * (size_t)me.idx;
  o[0] = nv[0]; o[1] = nv[1]; o[2] = nv[2];
  double* os = job.snor_out + 3 * (size_t)job.inv[me.idx];
  os[0] = nv[0]; os[1] = nv[1]; os[2] = nv[2];
  if (job.knn_out) {
#pragma unroll
    for (int t = 0; t < KMAX; ++t)
      if (t < K) job.knn_out[(size_t)me.idx * K + t] = bj(t) >= 0 ? (int)bo(t) : -1;   // nanoflann's result order
  }
}

#undef bd
#undef bj
#undef bo

}  // namespace

int launch_normals(mvicp_ctx* c, FrameDev& f, int k, int* d_knn) {
  if (!f.has_grid) { set_error("normals need the per-cloud hash structure"); return MVICP_ERR_STATE; }
  if (k < 3 || k > KMAX) { set_error("k = %d outside [3, %d]", k, KMAX); return MVICP_ERR_ARG; }
  // tie order: nanoflann's tree for this cloud — built once per cloud upload and kept with the frame (nn_tie.hip; the 1-NN kernels'
  // tie rule uses the same tree), like the reference's own lazy index (frame.cpp:209-214)
  {
    int fi = -1;
    for (int t = 0; t < c->n_frames; ++t) if (&c->frames[t] == &f) fi = t;
    if (fi < 0) { set_error("normals: frame is not part of the context"); return MVICP_ERR_ARG; }
    MV_CHECK(ensure_tie_trees(c, std::vector<int>(1, fi)));
  }
  NormJob j;
  const GridDev& g = f.grid;
  j.tie.nodes = static_cast<const VisitNode*>(f.tie_nodes); j.tie.slot = f.tie_slot;
  j.srec = (const PointRec*)g.crec; j.inv = g.inv; j.n = f.n;
  j.table = (const HashEntry*)g.table; j.mask = g.table_mask; j.shift = g.table_shift;
  j.ox = g.origin[0]; j.oy = g.origin[1]; j.oz = g.origin[2]; j.h = g.cell; j.inv_h = g.inv_cell;
  j.dx = g.dims[0]; j.dy = g.dims[1]; j.dz = g.dims[2];
  j.k = k; j.nor_out = f.nor; j.snor_out = f.grid.snor; j.knn_out = d_knn;
  {
    ProfScope ps(c, "normals", 0.0);
    hipLaunchKernelGGL(normals_kernel, dim3((f.n + NT - 1) / NT), dim3(NT), 0, c->stream, j);
  }
  MV_HIP(hipGetLastError());
  MV_HIP(hipStreamSynchronize(c->stream));
  return MVICP_OK;
}

}  // namespace mvicp
