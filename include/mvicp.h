/* mvicp.h — C ABI of libmvicp_hip.so: the MI355X (gfx950) multiview LM-ICP hot path.
 *
 * This is the drop-in boundary.  The reference (adrelino/mv-lm-icp) has no FFI: its
 * seams are plain C++ members/free functions.  Each entry point below names the reference interface
 * it replaces (file:line relative to the reference tree).  Plain pointers and sizes only; every call
 * returns 0 on success or a negative mvicp_status (one exception, stated at its declaration: mvicp_get_correspondences
 * returns the number of triples it wrote, >= 0), and mvicp_last_error() holds the message.  Test `< 0` for failure.  The
 * library owns all device memory; the caller owns every host buffer it passes.  One context drives
 * one GPU (one process per GPU; see mvicp_set_shard / mvicp_comm_init for the multi-GPU path).
 *
 * Conventions
 *   pose    : 16 doubles, 4x4 COLUMN-major = Eigen::Isometry3d::data()  (include/frame.h:42)
 *   points  : n x 3 doubles AoS = &std::vector<Eigen::Vector3d>[0]      (include/frame.h:38-39)
 *   edge e  : directed src -> dst = Frame::neighbours[j] of frame src    (include/frame.h:24-29,46)
 *   indices : int32, local to their frame
 */
#ifndef MVICP_H
#define MVICP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mvicp_ctx mvicp_ctx;

enum mvicp_status {
  MVICP_OK = 0,
  MVICP_ERR_ARG = -1,      /* bad argument / call order */
  MVICP_ERR_HIP = -2,      /* HIP runtime error (no GPU, OOM, launch failure) */
  MVICP_ERR_STATE = -3,    /* frames/graph/correspondences not set */
  MVICP_ERR_COMM = -4,     /* RCCL failure */
  MVICP_ERR_NUMERIC = -5,  /* LM solve failed (non-finite / not positive definite) */
  MVICP_ERR_INTERNAL = -6  /* a host-side C++ exception (out of memory, thread creation, ...) was caught at the boundary */
};

/* Rotation parameterization of the LM solve = which reference optimizer is mirrored. */
enum mvicp_param {
  MVICP_PARAM_EIGEN_QUATERNION = 0, /* ICP_Ceres::ceresOptimizer            src/internal/icp-ceres.cpp:220-323 */
  MVICP_PARAM_ANGLE_AXIS = 1,       /* ICP_Ceres::ceresOptimizer_ceresAngleAxis              :325-395 */
  MVICP_PARAM_SOPHUS_SE3 = 2        /* ICP_Ceres::ceresOptimizer_sophusSE3                   :398-475 */
};

/* Nearest-neighbour kernels (all exact; identical results by construction). */
enum mvicp_nn_method {
  MVICP_NN_AUTO = 0,
  MVICP_NN_BRUTE = 1, /* LDS-tiled exhaustive scan */
  MVICP_NN_GRID = 2,  /* per-query: spatial-hash (uniform grid) lookup with exact AABB-tree fallback */
  MVICP_NN_TILE = 3   /* per-wave: 64 curve-adjacent queries share a pruned brute force over LDS-staged 32-point tiles */
};

const char* mvicp_last_error(void);
/* "name major.minor" of the build, and the gfx arch the kernels were compiled for. */
const char* mvicp_version(void);

/* ---- lifetime -------------------------------------------------------------------------------- */
int mvicp_create(int device, mvicp_ctx** out);
int mvicp_destroy(mvicp_ctx* ctx);

/* ---- data upload (once; clouds are static in their local frame) ------------------------------- */
/* Replaces Frame::pts / Frame::nor as the kernel-visible copy (include/frame.h:38-39) and the lazy
 * KD-tree build of Frame::getClosestPoint (src/internal/frame.cpp:188-193): the per-cloud NN
 * structure is built here, once.  nrm may be NULL (point-to-point only).  The call returns when the cloud is on the device; the
 * structures themselves (k-d order, box hierarchy, matrix-pipe operands, hash) are built behind it on a host thread from a private copy
 * of the cloud, so that the clouds of an upload loop are built side by side (option "async_build", default 1); the first entry point that
 * needs a structure (mvicp_set_graph, mvicp_nn_query, mvicp_recompute_normals) waits for the pending builds and reports a failed one. */
int mvicp_set_num_frames(mvicp_ctx* ctx, int n_frames);
int mvicp_set_frame(mvicp_ctx* ctx, int frame, const double* xyz, const double* nrm, int n);

/* Replaces Frame::recomputeNormals() (include/frame.h:49, src/internal/frame.cpp:244-255; on by default in the reference,
 * main_multiview.cpp:49,68-70): normal of every point = eigenvector of the smallest eigenvalue of the covariance of its
 * k nearest points INCLUDING itself (reference k = 10), flipped so n_z <= 0 (include/common.h:331-346).  Overwrites the
 * frame's device normals; nrm_out (n x 3) and knn_out (n x k original indices, nearest first) may be NULL.  May be called
 * at any time: correspondence lists that point into this frame are re-gathered with the new normals (the reference reads
 * dstCloud.nor when it builds the problem, icp-ceres.cpp:270-292). */
int mvicp_recompute_normals(mvicp_ctx* ctx, int frame, int k, double* nrm_out, int* knn_out);

/* Pose graph = all Frame::neighbours[j].neighbourIdx (frame.cpp:67-89 builds it; main_multiview.cpp:
 * 104-117).  Edge order is the reference's loop order: src ascending, then neighbour order. */
int mvicp_set_graph(mvicp_ctx* ctx, int n_edges, const int* src, const int* dst);

/* Multi-GPU: this rank owns a contiguous chunk of the edge list (balanced by N_src).  Call before
 * mvicp_set_graph.  Default rank 0 of 1. */
int mvicp_set_shard(mvicp_ctx* ctx, int rank, int world);
/* The partition rule itself (pure host function, no context): owner[e] in [0, world) for edges with n_src[e]
 * source points each.  Contiguous chunks, balanced by source points. */
int mvicp_edge_owner(int n_edges, const int* n_src, int world, int* owner);
/* RCCL communicator for the per-edge normal-equation all-reduce.  librccl_path: the librccl.so to
 * dlopen (pass the one torch already loaded, or NULL for "librccl.so.1").  unique_id: 128 bytes
 * produced by mvicp_comm_unique_id on rank 0 and broadcast by the launcher. */
int mvicp_comm_unique_id(const char* librccl_path, void* unique_id_128);
int mvicp_comm_init(mvicp_ctx* ctx, const char* librccl_path, const void* unique_id_128, int rank, int world);
/* Ranks of the communicator as RCCL itself reports them (ncclCommCount): 0 = no communicator, -1 = this librccl does not say. */
int mvicp_comm_nranks(mvicp_ctx* ctx);

/* Alternative exchange without RCCL: the launcher supplies an in-place sum all-reduce over HOST doubles (e.g. MPI or
 * torch.distributed/gloo); the library stages the per-edge blocks through host memory.  Same sharding, same exactness;
 * meant for bring-up and for testing the N > 1 path where RCCL cannot run (several ranks on one GPU). */
typedef int (*mvicp_allreduce_fn)(void* user, double* host_buf, size_t n);
int mvicp_comm_set_callback(mvicp_ctx* ctx, mvicp_allreduce_fn fn, void* user);

/* ---- S1: correspondence search ----------------------------------------------------------------
 * Replaces, for ALL non-fixed frames at once, Frame::computeClosestPointsToNeighbours(frames, thresh)
 * (include/frame.h:54, src/internal/frame.cpp:91-185; caller main_multiview.cpp:119-127).
 *   poses  : n_frames x 16;  fixed : n_frames bytes (edges whose src is fixed are skipped, frame.cpp:93)
 *   thresh : the float cutoff (frame.h:54)
 *   counts / weights (n_edges, may be NULL): |correspondances| and OutgoingEdge::weight =
 *            (float)(1.5 * upper median distance) (frame.cpp:166-176).  weight of an empty edge = 0. */
int mvicp_correspond(mvicp_ctx* ctx, const double* poses, const unsigned char* fixed, float thresh, int nn_method,
                     int* counts, float* weights);
/* Start a new registration on the same clouds and graph: forget everything earlier searches left behind (temporal NN cache, seeds,
 * reusable lists, settled medians, MVICP_NN_AUTO state, the queued evaluation).  The next mvicp_correspond behaves like the first one
 * after mvicp_set_graph — the state of a fresh run of the reference program (main_multiview.cpp:130-148).  Host-only bookkeeping. */
int mvicp_reset_history(mvicp_ctx* ctx);
/* Copy edge e's list back as Frame::neighbours[j].correspondances (frame.h:18-22): ascending `first`.
 * RETURNS THE NUMBER OF TRIPLES WRITTEN (>= 0, = counts[e] of the last mvicp_correspond) or a negative mvicp_status;
 * cap is the capacity of the three output arrays (each may be NULL to skip that field). */
int mvicp_get_correspondences(mvicp_ctx* ctx, int edge, int cap, int* first, int* second, double* dist);
/* ALL lists of the last mvicp_correspond at once, as the reference lays them out: `struct Correspondance {int first; int second; double
 * dist;}` (include/frame.h:18-22), ascending `first` within an edge (frame.cpp:129,156-160).  One device pass un-sorts every edge this
 * rank owns, ONE asynchronous copy brings the triples into pinned host memory OWNED BY THE LIBRARY:
 *   *triples            -> the buffer;   *offsets -> n_edges + 1 positions: edge e = (*triples)[(*offsets)[e] .. (*offsets)[e + 1])
 * (zero width for edges of other ranks, edges whose source is fixed and edges that hold an explicit list).  Both pointers stay valid until
 * the next mvicp_correspond that changes a list (see mvicp_correspondence_epochs) / mvicp_set_correspondences / mvicp_set_graph /
 * mvicp_reset_history / mvicp_destroy on this context.  The
 * first call after a search does the work, later calls (and mvicp_get_correspondences, which slices the same buffer) are free. */
typedef struct mvicp_corr { int first; int second; double dist; } mvicp_corr;
int mvicp_map_correspondences(mvicp_ctx* ctx, const mvicp_corr** triples, const long long** offsets);
/* The same export WITHOUT waiting for it: the device pass is queued and the triples travel to the pinned buffer in chunks (one per run of edges with
 * the same source frame, in edge order).  *triples / *offsets are valid at once (the offsets are known from the search's counts); the BYTES of edge e
 * may be read after mvicp_wait_correspondences(ctx, e) returned — which waits for e's chunk only, so a caller that fills Frame::neighbours frame by
 * frame (frame.cpp:91-185 is called once per frame, main_multiview.cpp:119-127) copies frame i while the lists of frames i + 1 .. are still on the
 * bus.  Any later library call that waits for the stream (mvicp_map_correspondences, mvicp_get_correspondences, mvicp_optimize ...) completes it too. */
int mvicp_map_correspondences_async(mvicp_ctx* ctx, const mvicp_corr** triples, const long long** offsets);
int mvicp_wait_correspondences(mvicp_ctx* ctx, int edge);
/* Per-edge change counters of the lists, for callers that keep their own copy (the Frame mirror's `neighbours[j].correspondances`,
 * frame.cpp:110,156-160: the reference clears and refills every list every round; a caller that holds edge e's list with epoch x may skip the
 * refill while (*epochs)[e] == x).  An edge keeps its epoch across an mvicp_correspond iff its list is PROVABLY last search's bit for bit — same
 * cutoff, both poses bit-identical, searched then and now (a search is a pure function of these) — e.g. every edge in the rounds after the
 * registration has converged; then mvicp_map_correspondences returns the buffer it already holds without any device work.  Every other event
 * (a search with different inputs, mvicp_set_correspondences, mvicp_reset_history, a failed search) gives the edge a new, never repeated epoch.
 * *epochs -> n_edges counters owned by the library, valid until mvicp_set_graph / mvicp_destroy. */
int mvicp_correspondence_epochs(mvicp_ctx* ctx, const unsigned long long** epochs);
/* Install an explicit list (pairwise known-correspondence case, main_pairwise.cpp:60-61; tests). */
int mvicp_set_correspondences(mvicp_ctx* ctx, int edge, int n, const int* first, const int* second, float weight);

/* S1': batch form of Frame::getClosestPoint (frame.h:55, frame.cpp:187-206): queries are already in
 * the frame's local coordinates; returns index and SQUARED distance per query. */
int mvicp_nn_query(mvicp_ctx* ctx, int frame, const double* queries, int n, int nn_method, int* idx, double* d2);

/* ---- normal equations ---------------------------------------------------------------------------
 * Per edge: the 12x12 Gauss-Newton block of  sum rho(||r||^2)/2  in canonical right-perturbation
 * coordinates [upsilon_s, omega_s, upsilon_d, omega_d]  (T <- T exp(delta)), i.e. what
 * Ceres accumulates from the residual blocks of include/icp-ceres.h:49-316 + SoftLOneLoss(edge.weight)
 * (icp-ceres.cpp:284,374,449).  out: n_edges x 91 = [78 upper-triangular row-major H | 12 g | cost]. */
#define MVICP_EDGE_BLOCK 91
int mvicp_linearize(mvicp_ctx* ctx, const double* poses, int point_to_plane, int robust, double* out);

/* ---- S2: the LM solve ----------------------------------------------------------------------------
 * Replaces ICP_Ceres::ceresOptimizer / _ceresAngleAxis / _sophusSE3 (frames, pointToPlane, robust)
 * (include/icp-ceres.h:40-42; caller main_multiview.cpp:158-161).  poses in/out (frames[i]->pose).
 * fixed[0] is forced to 1 like icp-ceres.cpp:244,341,417.  Edges whose SOURCE frame is fixed contribute nothing (no cost, no
 * normal-equation terms), whatever correspondences they hold: the reference adds no residual blocks for them
 * (`if(srcCloud.fixed) continue;` icp-ceres.cpp:255,351,426).  An edge with a fixed DESTINATION keeps its residuals. */
typedef struct mvicp_summary {
  double initial_cost, final_cost;
  int iterations;        /* LM iterations after the initial evaluation (<= max_iterations) */
  int successful_steps;
  int termination;       /* 0 max-iterations, 1 gradient tol, 2 parameter tol, 3 function tol, 4 radius, -1 failure */
  int evaluations;       /* device linearize launches */
} mvicp_summary;
/* One deliberate deviation from the reference's write-back (icp-ceres.cpp:312-321,386-394,472-474 always converts the parameter
 * blocks back to poses, which re-orthonormalises R): a solve that ends WITHOUT taking a step (successful_steps == 0) returns the
 * caller's poses bit for bit.  pose -> parameters -> pose is a last-bit 2-cycle for some rotations, and the round trip would keep a
 * converged registration from being a fixed point of the round.  Costs in the summary are those of the round-tripped poses the
 * solver evaluates at (they differ from the returned ones by that last bit at most).  Callers that hand in a NON-orthonormal
 * rotation and rely on the write-back to repair it must orthonormalise it themselves. */
int mvicp_optimize(mvicp_ctx* ctx, double* poses, unsigned char* fixed, int param, int point_to_plane, int robust,
                   int max_iterations /* reference: 50, icp-ceres.cpp:81 */, mvicp_summary* summary);

/* Host-only form of the same solver over a caller-supplied evaluator (no GPU touched by this call):
 * eval(user, poses[n_frames x 16], blocks[n_edges x 91]) must fill the per-edge canonical blocks exactly
 * as mvicp_linearize does.  mvicp_optimize is this with the device evaluator plugged in. */
typedef int (*mvicp_eval_fn)(void* user, const double* poses, double* blocks);
int mvicp_lm_solve(int n_frames, int n_edges, const int* src, const int* dst, double* poses, unsigned char* fixed, int param,
                   int max_iterations, mvicp_eval_fn eval, void* user, mvicp_summary* summary);

/* ---- closed-form pairwise solvers (host only; no GPU touched) ----------------------------------------
 * Replace ICP_Closedform::pointToPoint / pointToPlane (include/icp-closedform.h:10-11, src/internal/icp-closedform.cpp:9-26,
 * 30-54): the reference's comparison baselines in main_pairwise.cpp:74-76,93-95 and an independent check of the
 * point-to-point / point-to-plane normal equations.  Index-aligned pairs (src[i], dst[i]) (and nor[i], the normal at dst[i]);
 * pose_out = the src -> dst transform, 16 doubles column-major.  point_to_point: the least-squares rigid transform (exact);
 * point_to_plane: one linearised step from identity with R = Rx Ry Rz of the three solved angles (icp-closedform.cpp:47-51). */
int mvicp_closedform_point_to_point(const double* src, const double* dst, int n, double* pose_out);
int mvicp_closedform_point_to_plane(const double* src, const double* dst, const double* nor, int n, double* pose_out);

/* Tuning / test switches.  "nn_tree_only" (0/1): skip the hash-grid fast path and answer every query with the
 * exact AABB-tree descent (same results; used by the parity tests to exercise the fallback on every query).
 * "grid_target" (points per occupied hash cell the cell-edge heuristic aims at; set before mvicp_set_frame).
 * "nn_cache" (0/1, default 1): temporal cache of the grid kernel — a query whose previous neighbour is provably still
 * nearest after the pose update skips the search (results are bit-identical either way).
 * "nn_census" (0/1): count candidates / boxes / cache hits per launch while profiling (feeds the algorithmic-byte model).
 * "spin_wait" (0/1, default 0): poll the stream for up to 2 ms before blocking on the per-evaluation / per-round waits.
 * "tile_seed" (0/1, default 1): the tile kernel starts from last round's neighbours; "tile_waves" (0 = auto, 4..8):
 * occupancy variant of the tile kernel; "tile_mfma" (0/1/2, default 1): the tile method screens an opened tile on the matrix pipe
 * (nn_mfma_kernel) — 1: except in cache-aware rounds, 2: always, 0: never (the fp32 VALU screen of nn_tile_kernel); "mfma_kacc" (default
 * 34): allowance for the fp32 accumulation inside one matrix instruction, in units of 2^-24 x sum |terms| (34 = seventeen truncating
 * additions; lower values are a measured, not a proven, bound); "mfma_trig" (default 2): a lane with more screen hits than this in one
 * tile makes the wave confirm nearest-first and screen the tile again; "mfma_lbt" (0/1, default 1): launches without any seed test a tile's box per
 * lane before screening it; "nn_search_factor" (default 4; 0 = unbounded): the kernels look for a neighbour within this many cutoffs (a query the
 * cutoff rejects keeps a seed and a temporal-cache bound; results are filtered by the cutoff afterwards, like the reference); "tie_rule" (0/1,
 * default 1): exact distance ties are decided the way nanoflann decides them (first visited target; 0 = lowest original index); "tie_lazy" (0/1, default 1;
 * single rank only, read at mvicp_set_graph): the reference-equivalent trees that decide ties are built when a search first reports a tie on a target
 * without one — that search is then repeated once — like the reference's own lazily built index (frame.cpp:188-193); 0 = built for every target at mvicp_set_graph; "sel_bracket" (0/1, default 1): one-pass median select around last round's median
 * once it has settled; "spec_eval" (0/1, default 1): mvicp_correspond queues the first linearization of the following
 * mvicp_optimize (same poses, previous solve's flags) behind its own kernels so the round waits once, not twice; "spec2_eval" (0/1, default 1; single rank): when a
 * search's poses are bit-identical to the last search's (a converged registration), the candidate evaluation of the last solve is queued as well — the fixed-point
 * round's solve then needs no further device launch and no second wait (used only if the solve asks for exactly those poses); "lin_share_p"
 * (0/1, default 1): the linearization reads the source points of an all-accepted edge from the shared sorted cloud; "lin_interleave" (0/1, default 1; read at
 * mvicp_set_graph): the linearization's workgroups of the edges that share a source cloud are launched interleaved in groups of 8, so that the second reader of a
 * piece of the cloud runs on the XCD whose L2 still holds it; "nn_cell"
 * (0/1, default 0): wave-cooperative cell-staging variant of the grid kernel; "tile_bounds" (default 1): the MVICP_NN_AUTO round
 * that hands over from the tile kernel to the grid kernel runs a build of the tile kernel that also leaves the temporal-cache
 * bounds (2: every tile round does, 0: off), "tile_mu" (default 0.02): its guard band in hash-cell edges; "tile_cache" (0/1, default 1):
 * after that hand-over, rounds whose poses still move run the same build with the temporal-cache check as its prologue (lanes whose
 * neighbour provably did not change sit out; the wave searches for its missed lanes only) and the grid kernel only re-verifies the fixed point (2: the
 * prologue in EVERY tile round that follows a bounds-leaving one — an experiment, profiles/r06_tile_ab.txt); "reject_cache" (0/1, default 1): in those rounds a query whose old neighbour
 * AND every other target are provably beyond the cutoff after the pose update is a cache hit too (it stays rejected, frame.cpp:156; its exact neighbour is never
 * output) — the lanes with the largest search balls leave the traversal; "cache_mfma_ratio" (default 3; 0 = never): a cache-aware round runs on the
 * matrix-pipe build instead when the median displacement bound of the queries since the last search exceeds this many guard bands (low expected hit rate);
 * "tile_miss" (default 8; 0 = off): in those
 * cache-aware rounds a wave left with at most this many missed lanes answers them one by one with its 64 lanes spread over the tile boxes / the points
 * (nn_tile.hip miss_block); "mfma_entry" (0/1, default 0): seeded matrix-pipe launches enter at the blocks their seeds lie in and prove completeness with one
 * flat sweep over the block boxes instead of the top-down walk (an experiment: faster only once the poses have settled); "prune_rho", "auto_settle", "auto_switch",
 * see DESIGN.md; "grid_curve" (default 2): device order of the clouds at the next mvicp_set_frame, 2 = balanced k-d order,
 * 1 / 0 = Hilbert / Morton index of the hash cell (nn_cell needs 0 or 1).  Tuning knobs: correspondences are
 * bit-identical for every setting. */
int mvicp_set_option(mvicp_ctx* ctx, const char* name, double value);
/* NN census accumulated while profiling and the "nn_census" option are on: counters in this order — queries, candidate points
 * examined, tree boxes / grid cells looked up, queries that needed the tree fallback, queries answered by the temporal cache,
 * candidate points fetched from memory (a wave-cooperative kernel fetches a point once and examines it from LDS many times). */
int mvicp_nn_census(mvicp_ctx* ctx, double* out5);              /* the first five counters (the 0.1 contract) */
/* All counters the build has, at most `cap` of them; returns how many were written (10 since version 0.4) or a negative status. */
int mvicp_nn_census_ex(mvicp_ctx* ctx, double* out, int cap);

/* ---- profiling (HIP events on the library's own stream) ------------------------------------------ */
/* on = 0: off; 1: every scope below; 2: only the NN kernels, "linearize" and "comm" (fewer event packets between the kernels of
 * a timed run). */
int mvicp_profile_enable(mvicp_ctx* ctx, int on);
int mvicp_profile_reset(mvicp_ctx* ctx);
/* kernel in {"nn_brute","nn_grid","nn_tile","nn_mfma" (one scope per NN kernel; "nn" = all of them together),"compact","gather","select","linearize",
 * "reduce","comm"} (HIP-event scopes) or a host timer ("host.correspond", "host.optimize", "host.evaluate", ...) or "spec.hit" (first evaluations
 * served by the queued launch: launches only): total ms, launches, bytes of the library's own model. */
int mvicp_profile_get(mvicp_ctx* ctx, const char* kernel, double* total_ms, long long* launches, double* alg_bytes);
/* The same entry as numbers: out[0..4] = total ms, launches, model bytes, SURVEY 8(d) algorithmic bytes (NN scopes: 36 B per query + 24 B per
 * candidate point fetched + 8 B per box or cell looked up; the last two need the "nn_census" option), queries answered.  Returns how many were written. */
int mvicp_profile_get_ex(mvicp_ctx* ctx, const char* kernel, double* out, int cap);
/* Opaque hipStream_t the library launches on (so a harness can bracket it with its own events). */
void* mvicp_stream(mvicp_ctx* ctx);
int mvicp_sync(mvicp_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* MVICP_H */
