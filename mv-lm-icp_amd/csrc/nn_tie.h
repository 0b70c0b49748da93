// nanoflann's tie rule for the 1-NN kernels (include/nanoflann.hpp:1205-1212: a leaf point replaces the running best only if it is
// STRICTLY nearer, so among targets at exactly the same distance the one the tree VISITS first wins — and the visit order depends on
// the query, nanoflann.hpp:1222-1233).  The search kernels keep their own cheap rule (lowest original index) and only REPORT the queries
// whose best distance was met by more than one target; nn_tie.hip re-answers exactly those queries by running the reference's own
// descent on the reference's own tree (kdvisit.h).  Exact ties are a measure-zero event for transformed queries, so the list is empty
// except on clouds with duplicated points — where the answer would otherwise differ from the reference's.
#pragma once
#include <vector>

#include "common.h"
#include "nn_list.h"

namespace mvicp {

// where a kernel reports: entry = (job index of the launch) << 32 | query index.  count may exceed cap (entries beyond cap are dropped):
// the fix-up then re-answers every query of the launch instead.
struct TieRef { unsigned long long* list; unsigned int* count; unsigned int cap; unsigned int job; };

__device__ __forceinline__ void tie_report(const TieRef& T, unsigned int i) {
  if (T.list == nullptr) return;
  const unsigned int k = atomicAdd(T.count, 1u);   // (a rare event: no aggregation needed)
  if (k < T.cap) T.list[k] = ((unsigned long long)T.job << 32) | i;
}

// One query set of a launch, as the fix-up sees it (the same q / xf / outputs the search kernel used).
struct TieJob {
  const void* nodes; const int* ord; double box[6];   // the target's tree: kdvisit.h VisitNode[], slot -> original index, root bounding box
  const double* tpts;                                  // target points, ORIGINAL order
  const double* q; const double* xf; int n;
  int* out_idx; const double* out_d2; const int* inv;  // inv: target original index -> what out_idx holds (sorted position); null: original indices
  ListRef list;                                        // list.dirty == null: no list
  long long q_begin;                                   // prefix of n over the launch's jobs (the re-answer-everything mode)
};

// host (nn_tie.hip)
int ensure_tie_trees(mvicp_ctx* c, const std::vector<int>& frames);
TieRef tie_ref(mvicp_ctx* c, size_t launch_queries, unsigned int job);   // the list the NEXT launch reports to (allocates / grows it)
void tie_job_fill(const FrameDev& dst, TieJob& j);                       // target part of a TieJob
int launch_tie_fixup(mvicp_ctx* c, const std::vector<TieJob>& jobs, double d2_bound);

}  // namespace mvicp
