// Internal declarations shared by the HIP translation units of libmvicp_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <future>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mvicp.h"
#include "abi_guard.h"

namespace mvicp {

void set_error(const char* fmt, ...);

#define MV_HIP(expr)                                                                          \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      mvicp::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));  \
      return MVICP_ERR_HIP;                                                                   \
    }                                                                                         \
  } while (0)

#define MV_CHECK(call)            \
  do {                            \
    int _s = (call);              \
    if (_s != MVICP_OK) return _s; \
  } while (0)

constexpr int kLinPartial = 32;      // doubles per linearize workgroup partial (28 plane / 29 point, padded)
constexpr int kLinThreads = 256;
constexpr int kCompactBlock = 1024;  // queries per compaction workgroup
constexpr int kSelBlock = 8192;      // keys per radix-select workgroup
constexpr int kEdgeXf = 40;          // Rs(9) ts(3) Rdinv(9) td(3), column-major; [24] = temporal-cache switch (>= 0: on, value =
                                     // rounding allowance in metres), [25..36] = dM (9, col-major) dv (3): change of the query map
                                     // q = M p + v since the last search, so a query moved by exactly |dM p + dv|
constexpr int kEdgeRel = 12;         // R_ds(9, column-major) t_ds(3)

// Uniform grid (spatial hash) over one cloud; see nn_grid.hip.
struct GridDev {
  int dims[3] = {0, 0, 0};
  double origin[3] = {0, 0, 0};
  double cell = 0.0, inv_cell = 0.0;
  int n_cells = 0;             // occupied cells
  double* spts = nullptr;      // n x 3 in the cloud's sorted order (balanced k-d order by default, ctx::grid_curve)
  int* sidx = nullptr;         // n: original index of sorted point
  void* srec = nullptr;        // n x 32 B {x, y, z, idx}: the same, packed for the per-lane candidate scans
  void* crec = nullptr;        // the records in hash-CELL order ({start, count} runs of `table`); aliases srec unless grid_curve = 2
  double* snor = nullptr;      // n x 3 normals in sorted order (null without normals)
  int* inv = nullptr;          // n: original index -> sorted position
  std::vector<int> h_order, h_inv;  // host copies: sorted -> original, original -> sorted
  void* table = nullptr;       // open-addressing hash: cell -> (start, count), 16-B entries
  unsigned int table_mask = 0; int table_shift = 0;
  float* oct = nullptr;        // implicit complete 8-ary box tree, 8 floats per node {lo.xyz, hi.xyz, pad2}
  int oct_leaf = 0; long long oct_first_leaf = 0;
  // 64-wide box hierarchy over the same sorted array (nn_tile.hip): level 0 = boxes of 64-point leaves,
  // level l+1 = boxes of 64 consecutive level-l boxes; SoA per level: 6 arrays of `wide_cnt[l]` floats.
  float* wide = nullptr;
  double maxabs = 0.0;         // largest |coordinate|
  int wide_levels = 0;
  int wide_cnt[6] = {0, 0, 0, 0, 0, 0};
  long long wide_off[6] = {0, 0, 0, 0, 0, 0};
  // matrix-pipe operands (nn_mfma.hip, build_mfma): per 32-point tile 64 x 16 B of ready-made v_mfma_f32_32x32x16_f16 A fragments, per block of
  // 64 tiles the origin / scale / error terms they refer to
  void* mf_ops = nullptr; void* mf_blk = nullptr;
  double struct_bytes = 0.0;
  // dense brick map over the cells (nn_grid.hip, nn_cell_kernel): brick = 4x4x4 cells -> {occupancy mask, row of the cell table};
  // cell table row = 64 x {start, count} of the cell runs in the sorted array.  Null when the grid is too large for a dense map.
  void* bricks = nullptr; void* celltab = nullptr; int bdims[3] = {0, 0, 0};
};

struct PointRec { double x, y, z; long long idx; };  // 32-B aligned sorted point + original index

// The per-cloud structure build of one mvicp_set_frame, running on a host thread of its own (api.cpp): the call returns once the cloud is
// on the device, the k-d order / box hierarchy / matrix-pipe operands / hash are built behind it, one cloud beside the other, and every
// entry point that needs a structure waits for the pending builds first (finish_builds).
struct BuildJob { std::future<int> fut; std::string err; std::vector<double> xyz, nrm; };

struct FrameDev {
  std::shared_ptr<BuildJob> job;   // pending structure build (null: none)
  std::string build_error;         // sticky: this cloud's structure build failed (message); cleared only by a new mvicp_set_frame of this slot.  While it
                                   // is set, every entry point that needs the structures fails — never a silent O(N^2) brute-force fallback
  int n = 0;
  double* pts = nullptr;  // n x 3 AoS, original order
  double* nor = nullptr;  // n x 3 or null
  GridDev grid;
  bool has_grid = false;
  double max_norm = 0.0;   // max |p| over the cloud (bounds how far a pose change can move a query)
  // the reference's own tree over this cloud (kdvisit.h: split structure of nanoflann's buildIndex, leaf_max_size 1): decides exact distance
  // ties the way the reference does (nn_tie.hip) and orders the k-NN lists of the normals (normals.hip); built on first use
  void* tie_nodes = nullptr; int* tie_ord = nullptr; int* tie_slot = nullptr; double tie_box[6] = {0, 0, 0, 0, 0, 0}; bool has_tie = false;
};

struct ProfEntry {
  double ms = 0.0;
  long long launches = 0;
  double bytes = 0.0;          // the library's own byte model of the scope's launches
  double survey_bytes = 0.0;   // NN scopes: SURVEY.md 8(d) algorithmic bytes (36 B/query + 24 B/candidate point fetched + 8 B/box or cell looked up)
  double queries = 0.0;        // NN scopes: queries answered
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
  std::vector<hipEvent_t> pool;
};

struct RcclApi;  // comm.cpp

}  // namespace mvicp

struct mvicp_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int n_frames = 0;
  std::vector<mvicp::FrameDev> frames;

  // graph + sharding
  int rank = 0, world = 1;
  int E = 0;
  std::vector<int> esrc, edst;
  std::vector<char> owned;         // E
  std::vector<long long> cap_off;  // E+1: prefix of N_src over OWNED edges (0-width for others)
  long long total_cap = 0;
  std::vector<int> h_count;        // E, valid after correspond / set_correspondences (owned edges)
  std::vector<float> h_weight;     // E
  bool have_corr = false;

  // device per-edge tables.  d_xf / d_nsrc / d_dirty and d_rel / d_a are views into ONE control block (d_ctl) mirrored in
  // pinned host memory, so a round costs one upload before the NN stage and one per LM evaluation:
  //   region 1 (correspond): xf [E x kEdgeXf] | nsrc [E int] | dirty [E int]        region 2 (evaluate): rel [E x kEdgeRel] | a [E]
  double* d_ctl = nullptr; size_t ctl_r1 = 0, ctl_r2_off = 0, ctl_r2 = 0;   // sizes / offset in doubles
  int* d_esrc = nullptr; int* d_edst = nullptr;
  long long* d_cap_off = nullptr;   // E+1
  int* d_nsrc = nullptr;            // E: N_src if owned and active else 0
  int* d_count = nullptr;           // E
  double* d_a = nullptr;            // E: SoftLOne scale (double)(float)weight
  double* d_xf = nullptr;           // E x kEdgeXf
  double* d_rel = nullptr;          // E x kEdgeRel
  // per-query (total_cap)
  int* d_nn_idx = nullptr; double* d_nn_d2 = nullptr;
  float* d_nn_lb = nullptr;         // fp32, rounded down: lower bound on the DISTANCE from the query to every target other than nn_idx (temporal cache)
  bool nn_cache_valid = false; bool nn_cache_enable = true; float nn_cache_thresh = -1.f;
  std::vector<char> nn_cache_edge;  // edges searched (active) in the last grid search
  std::vector<double> prev_q;       // E x 12: query map M = Rd^-1 Rs (9, col-major) and v = Rd^-1 (ts - td) of the last search
  std::vector<double> prev_xf;      // E x 24: the raw query transform (Rs ts Rd^-1 td) of the last search: bit-identical -> queries bit-identical
  // per-correspondence (total_cap)
  int* d_first = nullptr; int* d_second = nullptr; double* d_cd2 = nullptr;
  int* d_qpos = nullptr;            // per query: its position in the edge's compacted list, or -1 (rejected by the cutoff)
  int* d_dirty = nullptr;           // E: != 0 -> the edge's list (membership or a neighbour) changed this round: re-compact + re-gather
  int* d_dirty_slots = nullptr; int* d_dslot_off = nullptr; std::vector<int> dslot_off; int n_dslots = 0;  // one slot per 256 queries
  std::vector<char> list_valid;     // E: d_qpos / lists describe last round's result of this edge
  std::vector<char> explicit_list;  // E: the edge's list was installed by mvicp_set_correspondences (arbitrary order / repeats: no d_qpos for it)
  // the lists in the reference's layout (export.hip): {first, second, dist} triples, ascending first, all exportable edges of the last search
  void* d_export = nullptr; void* h_export = nullptr; size_t export_cap = 0;   // device staging / pinned host copy (capacity in triples)
  int* d_xblock_cnt = nullptr;
  std::vector<long long> export_off; // E+1: edge e's triples = h_export[export_off[e] .. export_off[e + 1])
  bool export_valid = false;         // h_export holds the lists as they are on the device now
  // the copy of the export into pinned memory goes in CHUNKS (one per run of edges with the same source frame), each followed by an event, so that a
  // caller that fills its lists frame by frame (host/frame.cpp) slices frame i while frames i+1.. are still on the bus (mvicp_map_correspondences_async)
  std::vector<hipEvent_t> export_events; std::vector<int> export_edge_chunk; int export_chunks = 0; bool export_in_flight = false;
  std::vector<char> qpos_valid;      // E: d_qpos / d_second / d_cd2 describe the LAST SEARCH's result of this edge (what the export reads).  Unlike list_valid
                                     // (= the list may be maintained in place next round) it survives mvicp_recompute_normals, which only re-gathers operands
  std::vector<unsigned long long> corr_epoch;   // E: changes whenever the edge's list (count, triples, weight) may differ from what it was (mvicp_correspondence_epochs)
  unsigned long long epoch_counter = 0;
  double* d_stream = nullptr;       // 10 x total_cap SoA: p (3) | n (3) | c = n . q | q (3)   (linearize.hip)
  // compaction scratch
  int n_cblocks = 0;                // total compaction blocks over owned edges
  std::vector<int> cblock_off;      // E+1
  int* d_cblock_off = nullptr; int* d_cblock_cnt = nullptr;
  // select scratch
  void* d_sel_state = nullptr; unsigned int* d_sel_hist = nullptr; double* d_median = nullptr;
  int n_sblocks = 0; std::vector<int> sblock_off; int* d_sblock_off = nullptr;   // kSelBlock keys per workgroup
  double* d_sel_lohi = nullptr;      // view into the control block: per edge the bracket [lo, hi] (d2) of the one-pass select
  std::vector<double> sel_med1, sel_med2;   // per owned edge: median d2 of the last / the one-before-last round (< 0: unknown)
  bool sel_bracket = true;           // option: use the one-pass bracket select once the medians have settled
  bool bracket_counters_clean = false;   // the bracket pass's two per-edge counters are zero on the device (left so by bracket_final_kernel)
  double* d_sel_keys1 = nullptr; double* d_sel_keys2 = nullptr;                  // compact key buffers of passes B and C (total_cap each)
  // linearize chunks
  bool lin_interleave = true;       // launch order of the linearize workgroups: chunks of the edges that share a source cloud interleaved in groups of 8 (api.cpp mvicp_set_graph)
  bool lin_share_p = true;          // linearize reads p from the sorted source cloud when an edge's list is the identity (option "lin_share_p")
  int lin_chunk_override = 0;
  int lin_chunk = 4096;             // correspondences per linearize workgroup (chosen from the GLOBAL problem size)
  int n_chunks = 0;
  std::vector<int> chunk_first;     // E+1
  int* d_chunk_edge = nullptr; int* d_chunk_start = nullptr; int* d_chunk_first = nullptr;
  double* d_partials = nullptr;     // n_chunks x kLinPartial
  double* d_out = nullptr;          // E x 91 blocks | E x 2 (count, median d2) | 1 "armed" slot: ONE buffer, so that with N > 1 ranks a round's
                                    // counts / medians / use-the-queued-evaluation decision travel in the same all-reduce as the queued blocks
  double* d_res_target = nullptr;   // where the select kernels put (count, median d2): null = mapped host memory (single rank), else d_out's tail
  // pinned (device-mapped) host staging: [control-block mirror | blocks E x 91 | results E x 2 | misc]
  double* h_pin = nullptr; size_t h_pin_doubles = 0;
  size_t pin_blocks_off = 0, pin_res_off = 0, pin_misc_off = 0;
  double* d_res_host = nullptr;     // device view of the results region: (count, median d2) per edge, written by select_final_kernel
  double* d_blocks_host = nullptr;  // device view of the blocks region: single-rank evaluations write the E x 91 blocks straight to the host
  double* lin_out = nullptr;        // where the next launch_linearize puts the E x 91 blocks (d_out or d_blocks_host)
  // Speculative first evaluation: mvicp_correspond queues the linearization the NEXT mvicp_optimize will ask for first (same poses,
  // the parameterization / cost flags of the previous solve) right behind the select kernels, so the round waits once for
  // (counts, medians, first blocks) instead of twice.  Used only if the solve really asks for exactly that evaluation.
  bool spec_enable = true; bool spec_flags_valid = false; int spec_param = 0, spec_plane = 0, spec_robust = 0;
  int spec_q_plane = 0, spec_q_robust = 0;   // flags the QUEUED evaluation was launched with (spec_plane / spec_robust may have moved on since)
  bool spec_arm = false;            // this correspond call queues one (select kernels write the SoftLOne scales on the device)
  bool spec_ready = false;          // blocks of spec_poses are in the pinned spec region
  std::vector<double> spec_poses;   // n_frames x 16: poses the speculative evaluation was made at (after the parameterization round trip)
  size_t pin_spec_off = 0, pin_adev_off = 0; double* d_spec_host = nullptr; double* d_adev_host = nullptr;
  // Second queued evaluation (round 6, single rank).  At the fixed point of a registration every round's solve is: first evaluation (queued, above) -> one LM
  // iteration -> candidate evaluation -> function-tolerance stop, and the candidate poses are last round's bit for bit (same inputs).  When this search's poses
  // are bit-identical to the last search's, the candidate evaluation of the LAST solve is queued right behind the first one (relative transforms from a second
  // pinned slice, copied into d_rel in stream order between the two), so the round waits ONCE for both.  Used only if the solve really asks for exactly those
  // poses; nothing is skipped — the work is queued earlier.
  bool spec2_enable = true, spec2_armed = false, spec2_ready = false; int spec2_plane = 0, spec2_robust = 0;
  std::vector<double> spec2_poses;       // poses the second queued evaluation was made at
  std::vector<double> last_cand_poses;   // poses of the last evaluation a solve asked for beyond its first (the prediction for the next solve's candidate)
  int last_cand_plane = -1, last_cand_robust = -1;
  size_t pin_spec2_off = 0, pin_rel2_off = 0; double* d_spec2_host = nullptr;
  double* d_a_check = nullptr;      // where this search's select kernels copy the SoftLOne scales they derive: mapped host memory (single rank) or d_out's tail
  bool spin_wait = false;           // poll the stream instead of a blocking wait (measured: no gain, HIP's own wait already spins)
  unsigned long long* h_census = nullptr;   // pinned: 8 counters of the last NN launch, resolved after the round's own sync
  bool census_pending = false; double census_nq = 0; int census_kind = 0; const char* census_scope = "nn_grid";   // kind: 0 grid (per-lane), 1 tree-only grid, 2 tile, 3 grid (cell staging)
  // brute-force split scratch
  int* d_split_idx = nullptr; double* d_split_d2 = nullptr; size_t split_cap = 0;

  // cached small tables
  struct CachedTable { std::vector<char> bytes; void* d = nullptr; size_t cap = 0; };
  std::map<std::string, CachedTable> tables;
  // table scratch
  char* d_scratch = nullptr; size_t scratch_bytes = 0, scratch_used = 0;
  std::vector<char> active;        // E: owned && src not fixed (set by correspond)

  // comm
  mvicp::RcclApi* rccl = nullptr;
  void* comm = nullptr;
  mvicp_allreduce_fn ar_fn = nullptr; void* ar_user = nullptr;   // host-staged all-reduce supplied by the launcher (no RCCL)
  std::vector<double> ar_host;

  bool async_build = true;          // option "async_build": mvicp_set_frame builds the per-cloud structures on a background host thread
  std::atomic<int> fault_inject_build{0};   // tests only: the n-th structure build from now fails (builds run on background threads)
  int fault_inject = 0, fault_inject_eval = 0;   // tests only: make the n-th search / exchanged evaluation from now fail locally before its collective
  // options / NN census (profiling only)
  bool list_reuse = true;          // skip compaction + gather for edges whose list did not change
  bool nn_tree_only = false;
  bool nn_census = false;          // count candidates / tree nodes per launch while profiling (small extra cost)
  void* d_census = nullptr; size_t census_bytes = 0;
  void* d_far_list = nullptr; size_t far_cap = 0; unsigned int* d_far_count = nullptr;  // nn_grid far-query list
  int far_parity = 0;              // d_far_count holds TWO counters used alternately; a launch zeroes the one the next launch will use
  unsigned int* h_far_seen = nullptr; unsigned int* d_far_seen = nullptr;   // mapped host word: far-list length of the last nn_far_kernel launch
  bool prev_grid_kernel = false;   // the previous search ran nn_grid_kernel + nn_far_kernel (so h_far_seen describes it)
  bool far_skip = false;           // set by mvicp_correspond: bit-identical queries after a grid round without far queries -> no far query, no phase-2 launch
  bool far_narrow = false;         // this grid launch expects (almost) no far queries: narrow far-kernel launch (set by mvicp_correspond)
  bool skip_dirty_reduce = false;  // set by mvicp_correspond for a search in which no list can change (see api.cpp)
  bool nn_skip_far = false;        // PROFILING ONLY: leave unresolved queries unresolved (wrong results)
  double nn_search_factor = 4.0;   // the kernels look for a neighbour within this many cutoffs (0 = unbounded, like the reference's findNeighbors): a query the
                                   // cutoff rejects then still has a neighbour to seed next round's search with and a temporal-cache bound, instead of being
                                   // searched from scratch every round (partial overlap: a third of the queries); the filter of frame.cpp:156 is applied after
  bool tie_lazy = true;            // single rank: the reference-equivalent trees (kdvisit.h) are built when a search first REPORTS a tie on a target without one (the
                                   // search is then repeated once), like the reference's own lazily built index (frame.cpp:188-193) — synthetic clouds never tie, and
                                   // the eager build was most of cfg5's set-up time.  N > 1 ranks build them at mvicp_set_graph (a repeated search is a collective)
  bool tie_rule = true;            // exact distance ties are decided as nanoflann decides them (first visited; nn_tie.hip); false: lowest original index
  unsigned long long* d_tie_list = nullptr; size_t tie_cap = 0; unsigned int* d_tie_count = nullptr; int tie_parity = 0;   // queries reported by the NN kernels
  unsigned int* h_tie_seen = nullptr; unsigned int* d_tie_seen = nullptr;   // mapped host word: reports of the last fix-up launch (read after the round's wait)
  unsigned int corr_tie_seen = 1u, corr_far_seen = 1u;   // what the LAST mvicp_correspond's own fix-up / far launch reported (read after its wait; 1 = unknown).
                                   // mvicp_nn_query runs the same launches and overwrites the mapped words, so the skip decisions below use these copies
  bool tie_skip = false;           // set by mvicp_correspond: a search that reproduces last round's queries bit for bit after a round without any report cannot report
  bool tile_seed = true;           // tile kernel starts from last round's neighbours when there are any
  int tile_bounds = 1;             // 1: the AUTO round that would hand over to the grid kernel runs the tile kernel's BND build instead (it leaves the
                                   // temporal-cache bounds, so the grid kernel starts with cache hits one round later and the uncached grid round — the
                                   // slowest of a registration — never runs); 2: every tile round leaves bounds (tests); 0: off
  int tile_cache = 1;              // AUTO, after the hand-over: rounds whose transforms still move run the tile kernel's bounds-leaving build WITH the
                                   // temporal-cache check as its prologue (missed lanes are searched wave-cooperatively) instead of the grid kernel
  double cache_mfma_ratio = 3.0;   // cache-aware rounds run on the matrix-pipe build when the median displacement bound of the queries exceeds this many guard bands
                                   // (eps / mu; 0 = never: always nn_tile_kernel).  Measured (profiles/r06_tile_ab.txt): eps / mu <= 2.2 <-> hit rates >= 0.86 (nn_tile_kernel +
                                   // miss_block faster), eps / mu >= 3.4 <-> hit rates <= 0.77 (nn_mfma_kernel 6-12 % faster)
  bool cached_on_mfma = false;     // this launch's decision (api.cpp -> launch_nn_tile_edges)
  bool reject_cache = true;        // cache-aware rounds: "provably still rejected by the cutoff" counts as a temporal-cache hit (nn_tile.hip / nn_mfma.hip prologue)
  int tile_miss = 8;               // cache-aware rounds: waves with at most this many missed lanes use nn_tile.hip's miss_block (0 = off)
  double tile_mu = 0.02;           // BND guard band as a fraction of the target's hash-cell edge (same role as prune_rho in the grid kernel); round 3 sweep on cfg4
                                   // (hand-over round + the two cache-aware rounds after it): 0.02 -> 2.06 ms, 0.05 -> 2.11, 0.1 -> 2.23, 0.2 -> 2.45
  int tile_mfma = 1;               // tile method: 1 = the screen of an opened tile runs on the matrix pipe (nn_mfma.hip) except in cache-aware rounds, 2 = always,
                                   // 0 = never (the fp32 VALU screen of nn_tile.hip)
  double mfma_kacc = 34.0;         // nn_mfma.hip: allowance for the fp32 accumulation inside one matrix instruction, in units of 2^-24 x sum |terms| (see tau_pieces)
  int mfma_lbt = 1;                // nn_mfma.hip: unseeded launches test every tile's box per lane before screening it (scan_block LBT)
  int mfma_entry = 0;              // nn_mfma.hip experiment (round 6): seeded launches enter at the seeds' blocks + one flat sweep over the block boxes instead of the top-down walk
  int mfma_trig = 2;               // nn_mfma.hip: a lane with more than this many screen hits in a tile triggers the nearest-first second screen
  int tile_waves = 0;              // nn_tile_kernel variant: waves per SIMD it is compiled for (0 = the measured best for the depth)
  double auto_prev_dist = 0.0; int auto_last_method = -1; double auto_settle = 0.5;   // MVICP_NN_AUTO policy state (api.cpp)
  double last_rms = -1.0;          // RMS residual at the end of the last mvicp_optimize since the last search (< 0: none): predicts the next NN distances
  bool nn_cell = false;            // grid method: wave-cooperative cell-staging kernel (nn_cell_kernel) instead of the per-lane hash kernel.  Exact and
                                   // tested at full size, but measured SLOWER on MI355X (cfg4 rounds 3-7: 5.4 / 3.2 / 2.3 / 1.7 / 1.0 ms vs 2.3 / 1.8 / 2.6 / 0.9 / 0.6 ms,
                                   // profiles/r02_nn_cell_experiment.txt): off by default
  double auto_switch = 1.0;        // AUTO: hand over from the tile kernel to the grid method once the median distance is below this many hash cells
  double prune_rho = 0.05;         // grid kernel: with a seed, skip block cells farther than seed distance + prune_rho * cell edge; 0 = off
  int grid_curve = 2;              // order of the sorted clouds: 0 Morton (Z-order) of the cells, 1 Hilbert of the cells, 2 balanced k-d order
  double grid_target = 5.0;        // points per occupied cell the cell-edge heuristic aims at (4-6 measure the same within 2 %)
  double nn_candidates = 0, nn_nodes = 0, nn_far = 0, nn_queries = 0, nn_hits = 0, nn_fetched = 0;
  double nn_dbg[4] = {0, 0, 0, 0};   // nn_mfma_kernel census: second-screen passes, confirmation rounds, blocks scanned (all per wave), fp64 confirmations (per lane)

  // profiling
  bool profile = false; int profile_level = 0;   // 1: every scope, 2: only "nn" and "linearize"
  std::map<std::string, mvicp::ProfEntry> prof;
};

namespace mvicp {

// kernels (each defined in its own TU)
int launch_nn_brute_edges(mvicp_ctx* c);                                             // nn_brute.hip
int launch_nn_brute_queries(mvicp_ctx* c, const FrameDev& f, const double* d_q, int n, int* d_idx, double* d_d2);
int launch_nn_grid_edges(mvicp_ctx* c, double d2_bound);                              // nn_grid.hip (d2_bound = the cutoff's; the search radius is search_bound())
double search_bound(const mvicp_ctx* c, double d2_bound);                               // api.cpp                              // nn_grid.hip
int launch_nn_tile_edges(mvicp_ctx* c, double d2_bound, bool with_bounds, bool with_cache, bool with_list);
int launch_nn_mfma_edges(mvicp_ctx* c, double d2_bound, bool with_bounds, bool with_cache, bool with_list);   // nn_mfma.hip
int build_mfma(FrameDev& f, const double* sorted_pts);                                 // nn_mfma.hip (host, called by build_grid)
int warm_nn_mfma(mvicp_ctx* c);
int warm_nn_tile(mvicp_ctx* c); int warm_nn_grid(mvicp_ctx* c);                                    // code-object load at set-up time (mvicp_set_graph)                              // nn_tile.hip
int build_wide(FrameDev& f, const double* sorted_pts);                                 // nn_tile.hip (host, called by build_grid)
int launch_nn_grid_queries(mvicp_ctx* c, const FrameDev& f, const double* d_q, int n, int* d_idx, double* d_d2);
int build_grid(mvicp_ctx* c, FrameDev& f, const double* h_xyz);
int finish_builds(mvicp_ctx* c);   // api.cpp: wait for every pending structure build of this context; first failure wins
void free_grid(GridDev& g);
void free_tie(FrameDev& f);                                                            // nn_tie.hip
int launch_compact(mvicp_ctx* c, double d2_bound);                                    // corr.hip
int launch_gather_stream(mvicp_ctx* c);
int launch_select_median(mvicp_ctx* c);
int launch_export(mvicp_ctx* c);           // export.hip: every exportable edge's list -> reference-order triples in pinned memory (async; caller waits)
int launch_select_bracket(mvicp_ctx* c);   // one-pass select around last round's medians (d_sel_lohi); flags edges it cannot answer
int launch_linearize(mvicp_ctx* c, int plane, int robust);                            // linearize.hip
int launch_normals(mvicp_ctx* c, FrameDev& f, int k, int* d_knn);                      // normals.hip
int stream_wait(mvicp_ctx* c);      // api.cpp: wait for the context's stream (spin-polls first)
void census_resolve(mvicp_ctx* c);  // api.cpp: fold the counters of the last NN launch into the profile (after a sync)

// small host->device table uploads through a persistent bump-allocated scratch buffer; the copy is a
// synchronous hipMemcpy (tables are tiny) so the pageable source may die right after the call.
// device copy of a small host table, re-uploaded only when its bytes change (job / pointer tables are the same every round)
int cached_upload(mvicp_ctx* c, const char* key, const void* src, size_t bytes, void** dptr);
void scratch_reset(mvicp_ctx* c);
int scratch_upload(mvicp_ctx* c, const void* src, size_t bytes, void** dptr);

// profiling helpers
struct ProfScope {
  mvicp_ctx* c; const char* name; hipEvent_t a = nullptr, b = nullptr; bool on;
  ProfScope(mvicp_ctx* c, const char* name, double bytes);
  ~ProfScope();
};
void prof_collect(mvicp_ctx* c);
void prof_collect_lazy(mvicp_ctx* c);   // the hot calls: only once a scope holds > 256 unresolved event pairs
// host wall-clock sections (same table, names prefixed "host."), only while profiling
struct HostScope {
  mvicp_ctx* c; const char* name; double t0; bool on;
  HostScope(mvicp_ctx* c, const char* name);
  ~HostScope();
};

}  // namespace mvicp
