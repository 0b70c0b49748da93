// In-place maintenance of an edge's compacted correspondence list by the NN kernels (nn_grid.hip, nn_tile.hip).
//
// The edge's list survives a round when every query keeps its acceptance (cutoff test, frame.cpp:156): squared distances are refreshed in
// place, a changed neighbour is patched in place (list entry + its operands in the packed stream), and compaction + operand gather are
// skipped for that edge.  Only a change of acceptance (the list's membership) marks the edge dirty.  No hot-address atomics: a query that
// invalidates the list stores 1 into its workgroup's slot (plain store, benign same-value race; one slot per 256 queries);
// dirty_reduce_kernel ORs the slots per edge afterwards.
#pragma once
#include "common.h"

namespace mvicp {

struct ListRef {
  const int* qpos; int* second; double* cd2; const int* dirty; int* dirty_slots;   // dirty: host-forced flag of the edge
  double* stream; long long total_cap; const double* dst_nor;                      // the edge's slice of the operand stream (linearize.hip) + sorted dst normals
  const PointRec* dst_srec;                                                         // the target's sorted records
};

// same_neighbour: the caller KNOWS idx_new is last round's neighbour (temporal-cache hit): a valid list then holds exactly that index at
// the query's position (the list was built from out_idx and every round since kept every neighbour), so the load is skipped.
__device__ __forceinline__ void update_list_entry(const ListRef& R, int i, int idx_new, double d2_new, double bound, bool same_neighbour) {
  if (*R.dirty != 0) return;   // forced dirty by the host (no valid list yet): nothing to check, qpos may be uninitialised
  const int pos = R.qpos[i];
  const bool acc = idx_new >= 0 && d2_new < bound;
  bool clean;
  if (pos < 0) clean = !acc;
  else {
    clean = acc;
    if (clean) {
      if (!same_neighbour && R.second[pos] != idx_new) {
        // The query keeps its place in the list (still accepted) but has a NEW neighbour: patch the entry and its operands in place —
        // n, c = n . q, q of the operand stream, exactly as gather_kernel writes them (corr.hip; same expression, no contraction) —
        // instead of declaring the whole edge dirty (which re-compacts and re-gathers all of its ~N_src entries).
        R.second[pos] = idx_new;
        const double2* pb = reinterpret_cast<const double2*>(R.dst_srec + idx_new);
        const double2 b0 = pb[0], b1 = pb[1];
        double* st = R.stream + pos;
        st[7 * R.total_cap] = b0.x; st[8 * R.total_cap] = b0.y; st[9 * R.total_cap] = b1.x;
        if (R.dst_nor != nullptr) {
          const double n0 = R.dst_nor[3 * (size_t)idx_new], n1 = R.dst_nor[3 * (size_t)idx_new + 1], n2 = R.dst_nor[3 * (size_t)idx_new + 2];
          st[3 * R.total_cap] = n0; st[4 * R.total_cap] = n1; st[5 * R.total_cap] = n2;
          st[6 * R.total_cap] = n0 * b0.x + n1 * b0.y + n2 * b1.x;
        }
      }
      R.cd2[pos] = d2_new;
    }
  }
  if (!clean) R.dirty_slots[i / 256] = 1;
}

// per-edge OR of the "list changed" slots -> d_dirty (nn_grid.hip); also leaves the slots zeroed for the next round
int launch_dirty_reduce(mvicp_ctx* c);

}  // namespace mvicp
