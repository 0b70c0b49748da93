// C-ABI of libmvicp_hip.so (include/mvicp.h): context, uploads, the correspondence pipeline
// (S1: NN -> cutoff/compaction -> median), per-edge normal equations, profiling.
// The LM solve (S2) lives in host/lm.cpp; the RCCL glue in comm.cpp.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <ctime>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <new>
#include <thread>

#include "common.h"
#include "comm.h"
#include "nn_tie.h"
#include "../host/se3.h"

namespace mvicp {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int abi_exception() noexcept {
  try { throw; }
  catch (const std::bad_alloc&) { set_error("out of host memory"); }
  catch (const std::exception& e) { set_error("internal error: %s", e.what()); }
  catch (...) { set_error("internal error (unknown exception)"); }
  return MVICP_ERR_INTERNAL;
}

// squared search radius of the NN kernels for a cutoff whose squared bound is d2_bound (ctx::nn_search_factor)
double search_bound(const mvicp_ctx* c, double d2_bound) {
  if (!(c->nn_search_factor > 0.0)) return 1.7976931348623157e308;
  const double f = std::max(1.0, c->nn_search_factor);
  const double b = d2_bound * f * f * (1.0 + 1e-9);
  return std::isfinite(b) ? b : 1.7976931348623157e308;
}

int cached_upload(mvicp_ctx* c, const char* key, const void* src, size_t bytes, void** dptr) {
  mvicp_ctx::CachedTable& t = c->tables[key];
  if (t.d && t.bytes.size() == bytes && std::memcmp(t.bytes.data(), src, bytes) == 0) { *dptr = t.d; return MVICP_OK; }
  MV_HIP(hipStreamSynchronize(c->stream));  // earlier kernels may still read the old copy
  if (bytes > t.cap) {
    if (t.d) MV_HIP(hipFree(t.d));
    t.d = nullptr;
    MV_HIP(hipMalloc(&t.d, std::max<size_t>(bytes, 256)));
    t.cap = std::max<size_t>(bytes, 256);
  }
  MV_HIP(hipMemcpy(t.d, src, bytes, hipMemcpyHostToDevice));
  t.bytes.assign((const char*)src, (const char*)src + bytes);
  *dptr = t.d;
  return MVICP_OK;
}

void scratch_reset(mvicp_ctx* c) { c->scratch_used = 0; }

int scratch_upload(mvicp_ctx* c, const void* src, size_t bytes, void** dptr) {
  const size_t aligned = (bytes + 255) & ~(size_t)255;
  if (c->scratch_used + aligned > c->scratch_bytes) {
    // grow: wait for in-flight users, then reallocate (callers re-upload everything after a reset)
    if (c->scratch_used != 0) { set_error("scratch overflow (%zu + %zu > %zu)", c->scratch_used, aligned, c->scratch_bytes); return MVICP_ERR_STATE; }
    MV_HIP(hipStreamSynchronize(c->stream));
    if (c->d_scratch) MV_HIP(hipFree(c->d_scratch));
    c->scratch_bytes = std::max<size_t>(aligned * 4, 1 << 20);
    MV_HIP(hipMalloc((void**)&c->d_scratch, c->scratch_bytes));
  }
  void* d = c->d_scratch + c->scratch_used;
  // stream-ordered w.r.t. earlier kernels that may still read the scratch, then synchronous for the host source
  MV_HIP(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, c->stream));
  MV_HIP(hipStreamSynchronize(c->stream));
  c->scratch_used += aligned;
  *dptr = d;
  return MVICP_OK;
}

ProfScope::ProfScope(mvicp_ctx* ctx, const char* nm, double bytes) : c(ctx), name(nm), on(ctx->profile) {
  // level 2: only the two roofline scopes (every event pair is two extra queue packets between kernels)
  if (on && ctx->profile_level >= 2 && ((std::strncmp(nm, "nn_", 3) != 0 && std::strcmp(nm, "linearize") != 0 && std::strcmp(nm, "comm") != 0) ||
                                        std::strcmp(nm, "nn_tie") == 0)) on = false;   // (the tie fix-up is a few microseconds per moving round: not worth two packets)
  if (!on) return;
  ProfEntry& pe = c->prof[name];
  auto get = [&]() {
    hipEvent_t ev = nullptr;
    if (!pe.pool.empty()) { ev = pe.pool.back(); pe.pool.pop_back(); }
    else if (hipEventCreate(&ev) != hipSuccess) ev = nullptr;
    return ev;
  };
  a = get(); b = get();
  pe.bytes += bytes;
  if (std::strncmp(nm, "nn_", 3) == 0) { pe.survey_bytes += bytes; pe.queries += bytes / 36.0; }   // NN scopes open with 36 B per query (24 B read + 12 B result)
  pe.launches += 1;
  if (a) (void)hipEventRecord(a, c->stream);
}
ProfScope::~ProfScope() {
  if (!on) return;
  if (b) (void)hipEventRecord(b, c->stream);
  if (a && b) c->prof[name].pending.push_back(std::make_pair(a, b));
}
static double now_ms() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
HostScope::HostScope(mvicp_ctx* ctx, const char* nm) : c(ctx), name(nm), t0(0.0), on(ctx->profile) { if (on) t0 = now_ms(); }
HostScope::~HostScope() {
  if (!on) return;
  ProfEntry& pe = c->prof[name];
  pe.ms += now_ms() - t0;
  pe.launches += 1;
}
// The hot calls (mvicp_correspond / mvicp_optimize / mvicp_linearize) only collect once a scope holds more than 256 unresolved event pairs: resolving
// them costs a hipEventSynchronize + hipEventElapsedTime per pair — 10-30 us of host time per ICP round with four scopes live, INSIDE the timed
// region of a bench run that needs the scopes for its roofline (5 % of a cfg4 fixed-point round, 15 % of a shard8 one).  The readers
// (mvicp_profile_get / _reset) always collect everything.
void prof_collect_lazy(mvicp_ctx* c) {
  for (auto& kv : c->prof) if (kv.second.pending.size() > 256) { prof_collect(c); return; }
}
void prof_collect(mvicp_ctx* c) {
  for (auto& kv : c->prof) {
    ProfEntry& pe = kv.second;
    for (auto& pr : pe.pending) {
      float ms = 0.f;
      if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) pe.ms += ms;
      pe.pool.push_back(pr.first);
      pe.pool.push_back(pr.second);
    }
    pe.pending.clear();
  }
}

namespace {

template <typename T> int dev_alloc(T** p, size_t n) {
  *p = nullptr;
  if (n == 0) n = 1;
  MV_HIP(hipMalloc((void**)p, sizeof(T) * n));
  return MVICP_OK;
}
template <typename T> void dev_free(T*& p) {
  if (p) (void)hipFree(p);
  p = nullptr;
}

void free_graph(mvicp_ctx* c) {
  dev_free(c->d_esrc); dev_free(c->d_edst); dev_free(c->d_cap_off); dev_free(c->d_count);
  dev_free(c->d_ctl); c->d_nsrc = nullptr; c->d_a = nullptr; c->d_xf = nullptr; c->d_rel = nullptr; c->d_dirty = nullptr;
  dev_free(c->d_nn_idx); dev_free(c->d_nn_d2); dev_free(c->d_nn_lb); dev_free(c->d_first); dev_free(c->d_second);
  dev_free(c->d_sblock_off); dev_free(c->d_sel_keys1); dev_free(c->d_sel_keys2); c->n_sblocks = 0;
  dev_free(c->d_cd2); dev_free(c->d_qpos); dev_free(c->d_dirty_slots); dev_free(c->d_dslot_off); dev_free(c->d_stream); dev_free(c->d_cblock_off); dev_free(c->d_cblock_cnt); if (c->d_sel_state) (void)hipFree(c->d_sel_state);
  c->d_sel_state = nullptr; dev_free(c->d_sel_hist); dev_free(c->d_median); dev_free(c->d_chunk_edge); dev_free(c->d_chunk_start);
  dev_free(c->d_chunk_first); dev_free(c->d_partials); dev_free(c->d_out);
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  c->h_pin = nullptr; c->h_pin_doubles = 0; c->d_res_host = nullptr; c->d_blocks_host = nullptr; c->lin_out = nullptr;
  c->census_pending = false; c->spec_ready = false; c->spec2_ready = false; c->spec_arm = false; c->d_spec_host = nullptr; c->d_adev_host = nullptr; c->d_res_target = nullptr; c->d_a_check = nullptr;
  dev_free(c->d_xblock_cnt); c->export_valid = false; c->export_in_flight = false; c->export_chunks = 0;
  for (hipEvent_t ev : c->export_events) (void)hipEventDestroy(ev);
  c->export_events.clear();
  if (c->d_export) (void)hipFree(c->d_export);
  if (c->h_export) (void)hipHostFree(c->h_export);
  c->d_export = nullptr; c->h_export = nullptr; c->export_cap = 0;
  c->E = 0; c->total_cap = 0; c->n_cblocks = 0; c->n_chunks = 0; c->have_corr = false;
}

int bind(mvicp_ctx* c) {
  if (!c) { set_error("null context"); return MVICP_ERR_ARG; }
  MV_HIP(hipSetDevice(c->device));
  return MVICP_OK;
}

// smallest double x such that the correctly rounded sqrt(x) >= t  ==>  (sqrt(d2) < t)  <=>  (d2 < x)
double sqrt_bound(double t) {
  double x = t * t;
  while (std::sqrt(x) >= t && x > 0.0) x = std::nextafter(x, 0.0);
  while (std::sqrt(x) < t) x = std::nextafter(x, INFINITY);
  return x;
}

// Eigen 3x3 inverse (cofactor form), column-major — frame.cpp:118 `dstCloud.pose.linear().inverse()`.
void inverse3(const double* m, double* r) {
#define M(i, j) m[(i) + 3 * (j)]
#define COF(i, j) (M(((i) + 1) % 3, ((j) + 1) % 3) * M(((i) + 2) % 3, ((j) + 2) % 3) - M(((i) + 1) % 3, ((j) + 2) % 3) * M(((i) + 2) % 3, ((j) + 1) % 3))
  const double c00 = COF(0, 0), c10 = COF(1, 0), c20 = COF(2, 0);
  const double det = (c00 * M(0, 0) + c10 * M(1, 0)) + c20 * M(2, 0);
  const double invdet = 1.0 / det;
  r[0] = c00 * invdet; r[3] = c10 * invdet; r[6] = c20 * invdet;
  r[1] = COF(0, 1) * invdet; r[4] = COF(1, 1) * invdet; r[7] = COF(2, 1) * invdet;
  r[2] = COF(0, 2) * invdet; r[5] = COF(1, 2) * invdet; r[8] = COF(2, 2) * invdet;
#undef COF
#undef M
}

int ensure_pin(mvicp_ctx* c, size_t doubles) {
  if (doubles <= c->h_pin_doubles) return MVICP_OK;
  if (c->h_pin) MV_HIP(hipHostFree(c->h_pin));
  c->h_pin = nullptr;
  MV_HIP(hipHostMalloc((void**)&c->h_pin, sizeof(double) * doubles, hipHostMallocMapped));
  std::memset(c->h_pin, 0, sizeof(double) * doubles);
  c->h_pin_doubles = doubles;
  return MVICP_OK;
}

}  // namespace

// Per-edge relative transform for the LM kernels: A = R_d^T R_s, t = R_d^T (t_s - t_d).
static void fill_rel_into(mvicp_ctx* c, const double* poses, double* h);
void fill_rel(mvicp_ctx* c, const double* poses) {
  fill_rel_into(c, poses, c->h_pin + c->ctl_r2_off);  // region 2 of the control block: rel | a (a = SoftLOne scales, set by correspond)
}
static void fill_rel_into(mvicp_ctx* c, const double* poses, double* h) {
  for (int e = 0; e < c->E; ++e) {
    const double* Ps = poses + 16 * (size_t)c->esrc[e];
    const double* Pd = poses + 16 * (size_t)c->edst[e];
    double* r = h + (size_t)e * kEdgeRel;
    for (int j = 0; j < 3; ++j)
      for (int i = 0; i < 3; ++i) r[i + 3 * j] = Pd[0 + 4 * i] * Ps[0 + 4 * j] + Pd[1 + 4 * i] * Ps[1 + 4 * j] + Pd[2 + 4 * i] * Ps[2 + 4 * j];
    const double dt[3] = {Ps[12] - Pd[12], Ps[13] - Pd[13], Ps[14] - Pd[14]};
    for (int i = 0; i < 3; ++i) r[9 + i] = Pd[0 + 4 * i] * dt[0] + Pd[1 + 4 * i] * dt[1] + Pd[2 + 4 * i] * dt[2];
  }
}
int upload_rel(mvicp_ctx* c, const double* poses) {
  fill_rel(c, poses);
  MV_HIP(hipMemcpyAsync(c->d_rel, c->h_pin + c->ctl_r2_off, sizeof(double) * c->ctl_r2, hipMemcpyHostToDevice, c->stream));
  return MVICP_OK;
}

// Wait for the stream.  The hot loop waits ~5 times per ICP round for a few hundred microseconds of GPU work; a blocking
// wait adds a scheduler wake-up to each, so poll first and only block when the work is long.
int stream_wait(mvicp_ctx* c) {
  if (c->spin_wait) {
    const double t0 = now_ms();
    for (;;) {
      const hipError_t q = hipStreamQuery(c->stream);
      if (q == hipSuccess) return MVICP_OK;
      if (q != hipErrorNotReady) { set_error("hipStreamQuery -> %s", hipGetErrorString(q)); return MVICP_ERR_HIP; }
      if (now_ms() - t0 > 2.0) break;
    }
  }
  MV_HIP(hipStreamSynchronize(c->stream));
  return MVICP_OK;
}

// NN census (profiling only): the counters of the last NN launch were copied to pinned memory asynchronously; fold them
// into the "nn" profile entry once the stream has been waited for anyway.
void census_resolve(mvicp_ctx* c) {
  if (!c->census_pending) return;
  c->census_pending = false;
  const unsigned long long* st = c->h_census;
  const double nq = c->census_nq;
  ProfEntry& pe = c->prof[c->census_scope];
  // SURVEY.md §8(d) algorithmic bytes of an NN launch = 36 B per query (already in the scope) + 24 B per candidate point FETCHED FROM
  // MEMORY + 8 B per cell / box looked up; the library's own finer model (record widths, cache state) goes to pe.bytes as before
  if (c->census_kind == 3) {
    // cell-staging kernel: st[4] points staged into LDS (24 B each, once per wave), st[5] distinct cells looked up per wave (16-B brick
    // entry + 8-B cell-table entry), st[0] candidates scanned from LDS (no memory traffic), far part (st[1] boxes) from nn_far_kernel
    const double hits = (double)st[3];
    pe.bytes += 36.0 * hits + 24.0 * (double)st[4] + 24.0 * (double)st[5] + 32.0 * (double)st[1];
    c->nn_candidates += (double)st[0]; c->nn_nodes += (double)st[1] + (double)st[5]; c->nn_far += (double)st[2]; c->nn_queries += nq; c->nn_hits += hits;
    c->nn_fetched += (double)st[4];
    pe.survey_bytes += 24.0 * (double)st[4] + 8.0 * ((double)st[1] + (double)st[5]);
  } else if (c->census_kind == 2) {
    c->nn_fetched += (double)st[0];
    pe.survey_bytes += 24.0 * (double)st[0] + 8.0 * (double)st[1];
    // tile kernel, memory side: every opened tile is loaded ONCE per wave (24 B xyz + 4 B index per point) and every tested
    // box once per wave (24 B); the per-lane distance evaluations (st[2]) are served from LDS.
    pe.bytes += 28.0 * (double)st[0] + 24.0 * (double)st[1];
    c->nn_candidates += (double)st[2]; c->nn_nodes += (double)st[1]; c->nn_queries += nq;
    for (int k = 0; k < 4; ++k) c->nn_dbg[k] += (double)st[4 + k];
    c->nn_hits += (double)st[3]; pe.bytes += 36.0 * (double)st[3];   // cache-aware rounds: lanes answered by the temporal cache (index, bound, neighbour point, bound write)
  } else {
    // cache hit: previous index 4 B + fp32 bound 4 B + one 24-B point + bound write 4 B; searched query: 8 hash slots x 16 B + bound
    // write 4 B; every candidate point examined: one 32-B record; every tree box tested: 32 B
    const double hits = (double)st[3], searched = nq - hits;
    pe.bytes += 36.0 * hits + (c->census_kind == 1 ? 4.0 : 132.0) * searched + 32.0 * (double)st[0] + 32.0 * (double)st[1];
    c->nn_candidates += (double)st[0]; c->nn_nodes += (double)st[1]; c->nn_far += (double)st[2]; c->nn_queries += nq; c->nn_hits += hits;
    c->nn_fetched += (double)st[0];   // per-lane kernel: every candidate examined is a record fetched
    pe.survey_bytes += 24.0 * (double)st[0] + 8.0 * (double)st[1];
  }
}

// ---- background structure builds (mvicp_set_frame) ---------------------------------------------------------------------------------------
namespace {
// at most one build per hardware thread at a time (each build forks a few threads of its own for the k-d order)
struct BuildSlots {
  std::mutex m; std::condition_variable cv; int free_slots;
  BuildSlots() : free_slots((int)std::max(2u, std::min(32u, std::thread::hardware_concurrency()))) {}
  void acquire() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [this]() { return free_slots > 0; }); --free_slots; }
  void release() { { std::lock_guard<std::mutex> lk(m); ++free_slots; } cv.notify_one(); }
};
BuildSlots& build_slots() { static BuildSlots s; return s; }
}  // namespace

// upload of the normals in the cloud's sorted order (needs the order the build produced)
static int upload_sorted_normals(FrameDev& f, const double* nrm) {
  const int n = f.n;
  std::vector<double> sn(3 * (size_t)n);
  for (int i = 0; i < n; ++i) std::memcpy(&sn[3 * (size_t)i], nrm + 3 * (size_t)f.grid.h_order[i], 24);
  MV_HIP(hipMalloc((void**)&f.grid.snor, sizeof(double) * 3 * (size_t)std::max(n, 1)));
  MV_HIP(hipMemcpy(f.grid.snor, sn.data(), sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice));
  return MVICP_OK;
}

static int build_frame_structures(mvicp_ctx* c, FrameDev& f, const double* xyz, const double* nrm) {
  if (c->fault_inject_build.load() > 0 && c->fault_inject_build.fetch_sub(1) == 1) { set_error("injected structure-build failure (option fault_inject_build)"); return MVICP_ERR_INTERNAL; }
  MV_CHECK(build_grid(c, f, xyz));
  if (nrm) MV_CHECK(upload_sorted_normals(f, nrm));
  return MVICP_OK;
}

int finish_builds(mvicp_ctx* c) {
  for (FrameDev& f : c->frames) {
    if (!f.job) continue;
    int s1 = MVICP_ERR_INTERNAL;
    try { s1 = f.job->fut.get(); } catch (...) { f.job->err = "structure build threw"; }
    if (s1 != MVICP_OK) {
      // sticky: the frame stays unusable (and says why) until it is uploaded again; its half-built structures are released now
      f.build_error = f.job->err.empty() ? std::string("structure build failed") : f.job->err;
      (void)hipSetDevice(c->device);
      free_grid(f.grid); f.has_grid = false;
    }
    f.job.reset();
  }
  for (size_t i = 0; i < c->frames.size(); ++i)
    if (!c->frames[i].build_error.empty()) {
      set_error("frame %d: %s (upload the cloud again with mvicp_set_frame)", (int)i, c->frames[i].build_error.c_str());
      return MVICP_ERR_STATE;
    }
  return MVICP_OK;
}

// Back-pressure of the background builds: each pending build holds a private copy of its cloud (48 B per point) and a parked thread; only the
// builds themselves are throttled by the slots.  An upload loop over many large clouds (64 x 1 M points = 3 GB of copies) therefore waits here
// for its oldest pending build once as many are pending as there are slots.  (The job's status is kept for finish_builds.)
static void throttle_builds(mvicp_ctx* c) {
  const int limit = std::max(2, (int)std::min(32u, std::thread::hardware_concurrency()));
  for (;;) {
    int pending = 0;
    BuildJob* oldest = nullptr;
    for (FrameDev& f : c->frames)
      if (f.job && f.job->fut.valid() && f.job->fut.wait_for(std::chrono::seconds(0)) != std::future_status::ready) { if (!oldest) oldest = f.job.get(); ++pending; }
    if (pending < limit || !oldest) return;
    oldest->fut.wait();
  }
}

// One device evaluation of all per-edge blocks at `poses` -> host `out` (E x 91), all-reduced over ranks.
int evaluate_blocks(mvicp_ctx* c, const double* poses, int plane, int robust, double* out) {
  if (!c->have_corr) { set_error("no correspondences: call mvicp_correspond or mvicp_set_correspondences first"); return MVICP_ERR_STATE; }
  if (plane) {
    for (int e = 0; e < c->E; ++e)
      if (c->owned[e] && c->h_count[e] > 0 && c->frames[c->edst[e]].grid.snor == nullptr) {
        set_error("point-to-plane needs normals on frame %d", c->edst[e]);
        return MVICP_ERR_STATE;
      }
  }
  HostScope hs(c, "host.evaluate");
  const size_t n = (size_t)c->E * MVICP_EDGE_BLOCK;
  if (c->spec_ready) {
    // the evaluation mvicp_correspond queued ahead: valid only for exactly these poses and flags
    c->spec_ready = false;
    if (plane == c->spec_q_plane && robust == c->spec_q_robust && c->spec_poses.size() == 16 * (size_t)c->n_frames &&
        std::memcmp(poses, c->spec_poses.data(), sizeof(double) * c->spec_poses.size()) == 0) {
      std::memcpy(out, c->h_pin + c->pin_spec_off, sizeof(double) * n);
      if (c->profile) c->prof["spec.hit"].launches += 1;   // (observable for tests / bench: evaluations served by the queued launch)
      return MVICP_OK;
    }
    c->spec2_ready = false;   // the solve did not start where the queued evaluations assumed: the second one is void as well
  }
  if (c->spec2_ready) {
    // the second queued evaluation (see common.h): valid only for exactly these poses and flags
    c->spec2_ready = false;
    if (plane == c->spec2_plane && robust == c->spec2_robust && c->spec2_poses.size() == 16 * (size_t)c->n_frames &&
        std::memcmp(poses, c->spec2_poses.data(), sizeof(double) * c->spec2_poses.size()) == 0) {
      std::memcpy(out, c->h_pin + c->pin_spec2_off, sizeof(double) * n);
      if (c->profile) c->prof["spec2.hit"].launches += 1;
      c->last_cand_poses.assign(poses, poses + 16 * (size_t)c->n_frames); c->last_cand_plane = plane; c->last_cand_robust = robust;
      return MVICP_OK;
    }
  }
  // (an evaluation beyond the solve's first: what the next solve's candidate evaluation will most likely be asked at, if the poses do not move)
  c->last_cand_poses.assign(poses, poses + 16 * (size_t)c->n_frames); c->last_cand_plane = plane; c->last_cand_robust = robust;
  double* h = c->h_pin + c->pin_blocks_off;
  if (c->comm || c->ar_fn) {
    // One collective per evaluation: [E x 91 blocks | poison slot].  The slot right behind the blocks (the buffer's tail region, rewritten
    // by every search) is zeroed, or — if this rank's upload / launch failed — poisoned with a NaN, so that a local failure reaches every
    // rank through the collective instead of leaving the peers blocked in it (see mvicp_correspond).
    c->lin_out = c->d_out;
    int st_l = upload_rel(c, poses);
    if (st_l == MVICP_OK && c->fault_inject_eval > 0 && --c->fault_inject_eval == 0) { set_error("injected launch failure (option fault_inject_eval)"); st_l = MVICP_ERR_HIP; }
    if (st_l == MVICP_OK) st_l = launch_linearize(c, plane, robust);
    char local_msg[sizeof(g_err)] = "";
    if (st_l != MVICP_OK) std::memcpy(local_msg, g_err, sizeof(local_msg));
    if (hipMemsetAsync(c->d_out + n, st_l == MVICP_OK ? 0 : 0xFF, sizeof(double), c->stream) != hipSuccess) {
      if (st_l != MVICP_OK) return st_l;
      set_error("hipMemsetAsync of the poison slot failed"); return MVICP_ERR_HIP;
    }
    { ProfScope pc(c, "comm", 8.0 * (double)(n + 1)); MV_CHECK(comm_allreduce_sum(c, c->d_out, n + 1)); }
    double* hx = c->h_pin + c->pin_spec_off;   // (the spec region holds E x 94 + 2 doubles: room for the slot; the queued blocks in it were consumed or dropped above)
    MV_HIP(hipMemcpyAsync(hx, c->d_out, sizeof(double) * (n + 1), hipMemcpyDeviceToHost, c->stream));
    MV_CHECK(stream_wait(c));
    if (std::isnan(hx[n])) {
      if (st_l != MVICP_OK) { set_error("%s", local_msg); return st_l; }
      set_error("a peer rank failed before the exchange of this evaluation (poisoned exchange buffer)");
      return MVICP_ERR_COMM;
    }
    std::memcpy(out, hx, sizeof(double) * n);
    return MVICP_OK;
  } else {
    MV_CHECK(upload_rel(c, poses));
    c->lin_out = c->d_blocks_host;   // 8 * 91 * E bytes: the reduce kernel stores them straight into mapped host memory
    MV_CHECK(launch_linearize(c, plane, robust));
  }
  MV_CHECK(stream_wait(c));
  std::memcpy(out, h, sizeof(double) * n);
  return MVICP_OK;
}

}  // namespace mvicp

using namespace mvicp;

extern "C" {

const char* mvicp_last_error(void) { return g_err; }
const char* mvicp_version(void) { return "mvicp_hip 0.5 (gfx950)"; }

int mvicp_create(int device, mvicp_ctx** out) try {
  if (!out) { set_error("out is null"); return MVICP_ERR_ARG; }
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    set_error("no HIP device available (%s): libmvicp_hip has no CPU fallback", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    return MVICP_ERR_HIP;
  }
  if (device < 0 || device >= ndev) { set_error("device %d out of range [0,%d)", device, ndev); return MVICP_ERR_ARG; }
  mvicp_ctx* c = new mvicp_ctx();
  c->device = device;
  MV_HIP(hipSetDevice(device));
  MV_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  *out = c;
  return MVICP_OK;
} MVICP_GUARD_ABI

int mvicp_destroy(mvicp_ctx* c) try {
  if (!c) return MVICP_OK;
  (void)hipSetDevice(c->device);
  (void)finish_builds(c);
  (void)hipStreamSynchronize(c->stream);
  comm_destroy(c);
  free_graph(c);
  for (FrameDev& f : c->frames) { dev_free(f.pts); dev_free(f.nor); free_grid(f.grid); free_tie(f); }
  dev_free(c->d_split_idx); dev_free(c->d_split_d2); dev_free(c->d_scratch);
  for (auto& kv : c->tables) if (kv.second.d) (void)hipFree(kv.second.d);
  if (c->d_census) (void)hipFree(c->d_census);
  if (c->h_census) (void)hipHostFree(c->h_census);
  if (c->d_tie_list) (void)hipFree(c->d_tie_list);
  if (c->d_tie_count) (void)hipFree(c->d_tie_count);
  if (c->h_tie_seen) (void)hipHostFree(c->h_tie_seen);
  if (c->h_far_seen) (void)hipHostFree(c->h_far_seen);
  if (c->d_far_list) (void)hipFree(c->d_far_list);
  if (c->d_far_count) (void)hipFree(c->d_far_count);
  for (auto& kv : c->prof) {
    for (auto& pr : kv.second.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (hipEvent_t ev : kv.second.pool) (void)hipEventDestroy(ev);
  }
  (void)hipStreamDestroy(c->stream);
  delete c;
  return MVICP_OK;
} MVICP_GUARD_ABI

int mvicp_set_num_frames(mvicp_ctx* c, int n_frames) try {
  MV_CHECK(bind(c));
  if (n_frames < 0) { set_error("n_frames < 0"); return MVICP_ERR_ARG; }
  (void)finish_builds(c);   // (builds of clouds that are dropped right here: their outcome no longer matters)
  if (c->E) free_graph(c);
  for (FrameDev& f : c->frames) { dev_free(f.pts); dev_free(f.nor); free_grid(f.grid); free_tie(f); }
  c->frames.assign(n_frames, FrameDev());
  c->n_frames = n_frames;
  return MVICP_OK;
} MVICP_GUARD_ABI

int mvicp_set_frame(mvicp_ctx* c, int frame, const double* xyz, const double* nrm, int n) try {
  MV_CHECK(bind(c));
  if (frame < 0 || frame >= c->n_frames) { set_error("frame %d out of range [0,%d)", frame, c->n_frames); return MVICP_ERR_ARG; }
  if (n < 0 || (n > 0 && !xyz)) { set_error("bad cloud (n=%d)", n); return MVICP_ERR_ARG; }
  if (c->E) { set_error("set frames before mvicp_set_graph"); return MVICP_ERR_STATE; }
  FrameDev& f = c->frames[frame];
  if (f.job) { try { (void)f.job->fut.get(); } catch (...) {} f.job.reset(); }   // a build of the cloud this call replaces
  dev_free(f.pts); dev_free(f.nor); free_grid(f.grid); free_tie(f);
  f.has_grid = false; f.build_error.clear();
  f.n = n;
  MV_CHECK(dev_alloc(&f.pts, 3 * (size_t)n));
  if (n) MV_HIP(hipMemcpy(f.pts, xyz, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice));
  if (nrm) {
    MV_CHECK(dev_alloc(&f.nor, 3 * (size_t)n));
    if (n) MV_HIP(hipMemcpy(f.nor, nrm, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice));
  }
  f.max_norm = 0.0;
  for (int i = 0; i < n; ++i) {
    const double* p = xyz + 3 * (size_t)i;
    if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) { set_error("non-finite coordinate in cloud"); return MVICP_ERR_ARG; }   // (reported by this call, not by a later one)
    f.max_norm = std::max(f.max_norm, std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]));
  }
  if (n == 0) return MVICP_OK;
  if (!c->async_build) {
    const int st = build_frame_structures(c, f, xyz, nrm);
    if (st != MVICP_OK) { f.build_error = g_err; free_grid(f.grid); f.has_grid = false; }
    return st;
  }
  // The per-cloud structures (k-d order, box hierarchy, matrix-pipe operands, hash: ~0.1 s of host work per 200 k points) are built on a
  // thread of their own from a private copy of the cloud, so that a driver's upload loop builds its clouds side by side (cfg4: 32 clouds,
  // 2.9 s one after the other); whoever needs a structure first waits for the pending builds (finish_builds).
  throttle_builds(c);
  std::shared_ptr<BuildJob> job = std::make_shared<BuildJob>();
  job->xyz.assign(xyz, xyz + 3 * (size_t)n);
  if (nrm) job->nrm.assign(nrm, nrm + 3 * (size_t)n);
  FrameDev* fp = &f;   // (stable: mvicp_set_num_frames, the only call that moves the frames, waits for the builds first)
  BuildJob* jp = job.get();
  job->fut = std::async(std::launch::async, [c, fp, jp]() -> int {
    build_slots().acquire();
    int st = MVICP_ERR_INTERNAL;
    try {
      st = hipSetDevice(c->device) == hipSuccess ? build_frame_structures(c, *fp, jp->xyz.data(), jp->nrm.empty() ? nullptr : jp->nrm.data()) : MVICP_ERR_HIP;
      if (st != MVICP_OK) jp->err = g_err;   // (this worker's thread-local message)
    } catch (const std::exception& e) { jp->err = std::string("structure build: ") + e.what(); }
    catch (...) { jp->err = "structure build threw"; }
    std::vector<double>().swap(jp->xyz); std::vector<double>().swap(jp->nrm);
    build_slots().release();
    return st;
  });
  f.job = job;
  return MVICP_OK;
} MVICP_GUARD_ABI

int mvicp_recompute_normals(mvicp_ctx* c, int frame, int k, double* nrm_out, int* knn_out) try {
  MV_CHECK(bind(c));
  if (frame < 0 || frame >= c->n_frames) { set_error("frame %d out of range", frame); return MVICP_ERR_ARG; }
  MV_CHECK(finish_builds(c));
  FrameDev& f = c->frames[frame];
  if (f.n < k) { set_error("frame %d has %d points < k = %d (common.h:333 asserts >= 3)", frame, f.n, k); return MVICP_ERR_STATE; }
  if (!f.nor) MV_CHECK(dev_alloc(&f.nor, 3 * (size_t)f.n));
  if (!f.grid.snor) MV_CHECK(dev_alloc(&f.grid.snor, 3 * (size_t)f.n));
  int* d_knn = nullptr;
  if (knn_out) MV_CHECK(dev_alloc(&d_knn, (size_t)f.n * k));
  int st = launch_normals(c, f, k, d_knn);
  if (st == MVICP_OK) {
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && nrm_out) e = hipMemcpy(nrm_out, f.nor, sizeof(double) * 3 * (size_t)f.n, hipMemcpyDeviceToHost);
    if (e == hipSuccess && knn_out) e = hipMemcpy(knn_out, d_knn, sizeof(int) * (size_t)f.n * k, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error("recompute_normals: %s", hipGetErrorString(e)); st = MVICP_ERR_HIP; }
  }
  dev_free(d_knn);
  c->spec_ready = false; c->spec2_ready = false;
  if (st == MVICP_OK && c->E > 0) {
    // The packed operand stream bakes the dst normals in (n and c = n . q, gathered at correspond time) while the reference
    // reads dstCloud.nor when it builds the problem (icp-ceres.cpp:270-292): every list that points INTO this frame is stale.
    // Re-gather those edges now and never reuse their lists unchecked.
    bool any = false;
    std::vector<int> dirty(c->E, 0);
    for (int e = 0; e < c->E; ++e)
      if (c->owned[e] && c->edst[e] == frame) { c->list_valid[e] = 0; if (c->have_corr && c->h_count[e] > 0) { dirty[e] = 1; any = true; } }
    if (any) {
      MV_HIP(hipMemcpy(c->d_dirty, dirty.data(), sizeof(int) * c->E, hipMemcpyHostToDevice));
      MV_CHECK(launch_gather_stream(c));
      MV_HIP(hipStreamSynchronize(c->stream));
    }
  }
  if (c->profile) prof_collect_lazy(c);
  return st;
} MVICP_GUARD_ABI

int mvicp_set_shard(mvicp_ctx* c, int rank, int world) try {
  MV_CHECK(bind(c));
  if (world < 1 || rank < 0 || rank >= world) { set_error("bad shard %d/%d", rank, world); return MVICP_ERR_ARG; }
  if (c->E) { set_error("call mvicp_set_shard before mvicp_set_graph"); return MVICP_ERR_STATE; }
  c->rank = rank; c->world = world;
  return MVICP_OK;
} MVICP_GUARD_ABI

int mvicp_edge_owner(int n_edges, const int* n_src, int world, int* owner) try {
  if (n_edges < 0 || world < 1 || (n_edges > 0 && (!n_src || !owner))) { set_error("bad arguments"); return MVICP_ERR_ARG; }
  double total = 0, cum = 0;
  for (int e = 0; e < n_edges; ++e) total += n_src[e];
  for (int e = 0; e < n_edges; ++e) {
    const double n = n_src[e];
    int o = total > 0 ? (int)std::floor((cum + 0.5 * n) * world / total) : 0;
    owner[e] = std::min(std::max(o, 0), world - 1);
    cum += n;
  }
  return MVICP_OK;
} MVICP_GUARD_ABI

int mvicp_set_graph(mvicp_ctx* c, int n_edges, const int* src, const int* dst) try {
  MV_CHECK(bind(c));
  if (n_edges < 0 || (n_edges > 0 && (!src || !dst))) { set_error("bad edge list"); return MVICP_ERR_ARG; }
  for (int e = 0; e < n_edges; ++e)
    if (src[e] < 0 || src[e] >= c->n_frames || dst[e] < 0 || dst[e] >= c->n_frames || src[e] == dst[e]) {
      set_error("edge %d (%d->%d) invalid", e, src[e], dst[e]);
      return MVICP_ERR_ARG;
    }
  MV_CHECK(finish_builds(c));   // every cloud's structures are needed from here on
  free_graph(c);
  const int E = n_edges;
  c->E = E;
  c->esrc.assign(src, src + E);
  c->edst.assign(dst, dst + E);
  c->owned.assign(E, 0);
  c->active.assign(E, 0);
  c->h_count.assign(E, 0);
  c->h_weight.assign(E, 0.f);
  // contiguous edge chunks balanced by N_src
  {
    std::vector<int> ns(E), owner(E);
    for (int e = 0; e < E; ++e) ns[e] = c->frames[src[e]].n;
    MV_CHECK(mvicp_edge_owner(E, ns.data(), c->world, owner.data()));
    for (int e = 0; e < E; ++e) c->owned[e] = owner[e] == c->rank;
  }
  // linearize workgroup chunk: from the GLOBAL source-point total (identical on every rank, so per-edge sums
  // have the same association order for any GPU count); small problems get small chunks to fill 256 CUs.
  {
    double total = 0;
    for (int e = 0; e < E; ++e) total += c->frames[src[e]].n;
    c->lin_chunk = total >= 6e6 ? 8192 : total >= 2e6 ? 4096 : total >= 5e5 ? 1024 : 512;
    if (c->lin_chunk_override > 0) c->lin_chunk = c->lin_chunk_override;
  }
  const int kLinChunk = c->lin_chunk;
  c->cap_off.assign(E + 1, 0);
  c->cblock_off.assign(E + 1, 0);
  c->chunk_first.assign(E + 1, 0);
  std::vector<int> chunk_edge, chunk_start;
  for (int e = 0; e < E; ++e) {
    const long long n = c->owned[e] ? c->frames[src[e]].n : 0;
    c->cap_off[e + 1] = c->cap_off[e] + ((n + 63) / 64) * 64;
    c->cblock_off[e + 1] = c->cblock_off[e] + (int)((n + kCompactBlock - 1) / kCompactBlock);
    const int nch = (int)((n + kLinChunk - 1) / kLinChunk);
    c->chunk_first[e + 1] = c->chunk_first[e] + nch;
    for (int k = 0; k < nch; ++k) { chunk_edge.push_back(e); chunk_start.push_back(k * kLinChunk); }
  }
  c->total_cap = c->cap_off[E];
  c->n_cblocks = c->cblock_off[E];
  c->n_chunks = c->chunk_first[E];
  if (c->lin_interleave) {
    // Launch order of the linearize workgroups (option "lin_interleave"): the edges of one source cloud read the same p (identity lists, lin_share_p).  Their
    // chunks are interleaved in groups of 8 — A0..A7, B0..B7, A8..A15, ... — so that chunk k of the second edge runs on the XCD (workgroup b runs on XCD b % 8)
    // that chunk k of the first edge ran on a moment ago: its piece of p is still in that XCD's L2.  Slots of the partials are per edge (linearize.hip).
    std::vector<int> ce, cs;
    for (int e0 = 0; e0 < E;) {
      int e1 = e0 + 1;
      while (e1 < E && src[e1] == src[e0]) ++e1;
      int maxc = 0;
      for (int e = e0; e < e1; ++e) maxc = std::max(maxc, c->chunk_first[e + 1] - c->chunk_first[e]);
      for (int g = 0; g < maxc; g += 8)
        for (int e = e0; e < e1; ++e) {
          const int nch = c->chunk_first[e + 1] - c->chunk_first[e];
          for (int k = g; k < std::min(g + 8, nch); ++k) { ce.push_back(e); cs.push_back(k * kLinChunk); }
        }
      e0 = e1;
    }
    chunk_edge.swap(ce); chunk_start.swap(cs);
  }
  const size_t cap = (size_t)c->total_cap;
  MV_CHECK(dev_alloc(&c->d_esrc, E)); MV_CHECK(dev_alloc(&c->d_edst, E)); MV_CHECK(dev_alloc(&c->d_cap_off, E + 1));
  MV_CHECK(dev_alloc(&c->d_count, E));
  // control block (see common.h): region 1 = xf | nsrc | dirty, region 2 = rel | a
  c->ctl_r1 = (size_t)E * kEdgeXf + (size_t)E + 2 * (size_t)E;  // xf | 2E ints = E doubles | select brackets (lo, hi) per edge
  c->ctl_r2_off = (c->ctl_r1 + 1) & ~(size_t)1;                // 16-B aligned
  c->ctl_r2 = (size_t)E * (kEdgeRel + 1);
  MV_CHECK(dev_alloc(&c->d_ctl, c->ctl_r2_off + c->ctl_r2));
  MV_HIP(hipMemset(c->d_ctl, 0, sizeof(double) * std::max<size_t>(c->ctl_r2_off + c->ctl_r2, 1)));
  c->d_xf = c->d_ctl; c->d_nsrc = reinterpret_cast<int*>(c->d_ctl + (size_t)E * kEdgeXf); c->d_dirty = c->d_nsrc + E;
  c->d_sel_lohi = c->d_ctl + (size_t)E * (kEdgeXf + 1);
  c->sel_med1.assign(E, -1.0); c->sel_med2.assign(E, -1.0);
  c->d_rel = c->d_ctl + c->ctl_r2_off; c->d_a = c->d_rel + (size_t)E * kEdgeRel;
  MV_CHECK(dev_alloc(&c->d_nn_idx, cap)); MV_CHECK(dev_alloc(&c->d_nn_d2, cap)); MV_CHECK(dev_alloc(&c->d_nn_lb, cap));
  c->nn_cache_valid = false; c->prev_q.assign((size_t)E * 12, 0.0); c->nn_cache_edge.assign(E, 0);
  c->auto_prev_dist = 0.0; c->auto_last_method = -1; c->corr_tie_seen = c->corr_far_seen = 1u;
  MV_CHECK(dev_alloc(&c->d_first, cap)); MV_CHECK(dev_alloc(&c->d_second, cap)); MV_CHECK(dev_alloc(&c->d_cd2, cap));
  MV_CHECK(dev_alloc(&c->d_stream, 10 * cap));
  MV_CHECK(dev_alloc(&c->d_qpos, cap));
  MV_HIP(hipMemset(c->d_qpos, 0xff, sizeof(int) * std::max<size_t>(cap, 1)));
  c->list_valid.assign(E, 0); c->explicit_list.assign(E, 0); c->export_valid = false; c->export_off.assign((size_t)E + 1, 0);
  c->qpos_valid.assign(E, 0); c->corr_epoch.assign(E, 0);
  for (int e = 0; e < E; ++e) c->corr_epoch[e] = ++c->epoch_counter;
  c->dslot_off.assign(E + 1, 0);
  for (int e = 0; e < E; ++e) c->dslot_off[e + 1] = c->dslot_off[e] + (int)(((c->owned[e] ? c->frames[src[e]].n : 0) + 255) / 256);
  c->n_dslots = c->dslot_off[E];
  MV_CHECK(dev_alloc(&c->d_dirty_slots, (size_t)c->n_dslots)); MV_CHECK(dev_alloc(&c->d_dslot_off, (size_t)E + 1));
  MV_HIP(hipMemset(c->d_dirty_slots, 0, sizeof(int) * std::max<size_t>((size_t)c->n_dslots, 1)));   // dirty_reduce_kernel re-zeroes what it reads
  MV_HIP(hipMemcpy(c->d_dslot_off, c->dslot_off.data(), sizeof(int) * (E + 1), hipMemcpyHostToDevice));
  MV_CHECK(dev_alloc(&c->d_cblock_off, E + 1)); MV_CHECK(dev_alloc(&c->d_cblock_cnt, (size_t)c->n_cblocks));
  MV_HIP(hipMalloc(&c->d_sel_state, 16 * 3 * (size_t)std::max(E, 1))); MV_CHECK(dev_alloc(&c->d_sel_hist, (size_t)E * (3 * 2048 + 5)));
  c->sblock_off.assign(E + 1, 0);
  for (int e = 0; e < E; ++e) c->sblock_off[e + 1] = c->sblock_off[e] + (int)(((c->owned[e] ? c->frames[src[e]].n : 0) + kSelBlock - 1) / kSelBlock);
  c->n_sblocks = c->sblock_off[E];
  MV_CHECK(dev_alloc(&c->d_sblock_off, (size_t)E + 1));
  MV_HIP(hipMemcpy(c->d_sblock_off, c->sblock_off.data(), sizeof(int) * (E + 1), hipMemcpyHostToDevice));
  MV_CHECK(dev_alloc(&c->d_sel_keys1, cap)); MV_CHECK(dev_alloc(&c->d_sel_keys2, cap));
  // the grid kernel's far-query list (worst case every query): allocated here, not lazily inside the first grid launch — a
  // 100 MB hipMalloc in the middle of an ICP loop costs milliseconds
  if (cap > c->far_cap) {
    if (c->d_far_list) MV_HIP(hipFree(c->d_far_list));
    c->d_far_list = nullptr; c->far_cap = 0;
    MV_HIP(hipMalloc(&c->d_far_list, sizeof(int) * 2 * cap));
    c->far_cap = cap;
  }
  MV_CHECK(warm_nn_grid(c)); MV_CHECK(warm_nn_tile(c)); MV_CHECK(warm_nn_mfma(c));
  if (c->tie_rule && !(c->tie_lazy && c->world == 1)) {   // the reference's own trees over the targets of the edges THIS RANK owns: they decide exact distance ties (the others are never searched here)
    std::vector<int> need;
    for (int e = 0; e < E; ++e) if (c->owned[e]) need.push_back(dst[e]);
    MV_CHECK(ensure_tie_trees(c, need));
  }
  MV_CHECK(dev_alloc(&c->d_median, E));
  MV_CHECK(dev_alloc(&c->d_chunk_edge, (size_t)c->n_chunks)); MV_CHECK(dev_alloc(&c->d_chunk_start, (size_t)c->n_chunks));
  MV_CHECK(dev_alloc(&c->d_chunk_first, E + 1));
  MV_CHECK(dev_alloc(&c->d_partials, (size_t)c->n_chunks * kLinPartial));
  MV_CHECK(dev_alloc(&c->d_out, (size_t)E * (MVICP_EDGE_BLOCK + 3) + 2));   // blocks | (count, median d2) x E | armed | a x E  (see common.h)
  MV_HIP(hipMemset(c->d_out, 0, sizeof(double) * ((size_t)E * (MVICP_EDGE_BLOCK + 3) + 2)));
  if (E) {
    MV_HIP(hipMemcpy(c->d_esrc, src, sizeof(int) * E, hipMemcpyHostToDevice));
    MV_HIP(hipMemcpy(c->d_edst, dst, sizeof(int) * E, hipMemcpyHostToDevice));
    MV_HIP(hipMemset(c->d_count, 0, sizeof(int) * E));
  }
  MV_HIP(hipMemcpy(c->d_cap_off, c->cap_off.data(), sizeof(long long) * (E + 1), hipMemcpyHostToDevice));
  MV_HIP(hipMemcpy(c->d_cblock_off, c->cblock_off.data(), sizeof(int) * (E + 1), hipMemcpyHostToDevice));
  MV_HIP(hipMemcpy(c->d_chunk_first, c->chunk_first.data(), sizeof(int) * (E + 1), hipMemcpyHostToDevice));
  if (c->n_chunks) {
    MV_HIP(hipMemcpy(c->d_chunk_edge, chunk_edge.data(), sizeof(int) * c->n_chunks, hipMemcpyHostToDevice));
    MV_HIP(hipMemcpy(c->d_chunk_start, chunk_start.data(), sizeof(int) * c->n_chunks, hipMemcpyHostToDevice));
  }
  // pinned, device-mapped staging: control-block mirror | blocks | results | misc
  c->pin_blocks_off = c->ctl_r2_off + c->ctl_r2;
  c->pin_res_off = c->pin_blocks_off + (size_t)E * MVICP_EDGE_BLOCK;
  c->pin_misc_off = c->pin_res_off + 2 * (size_t)E + 2;           // (+ the "armed" slot the select kernels write behind the E pairs)
  c->pin_spec_off = c->pin_misc_off + 4 * (size_t)E + 64;          // blocks of the speculative first evaluation (+ the exchanged tail with N > 1 ranks)
  c->pin_adev_off = c->pin_spec_off + (size_t)E * (MVICP_EDGE_BLOCK + 3) + 2; // SoftLOne scales as the device computed them (single rank)
  if (c->h_pin) { MV_HIP(hipHostFree(c->h_pin)); c->h_pin = nullptr; c->h_pin_doubles = 0; }
  c->pin_spec2_off = (c->pin_adev_off + (size_t)E + 8 + 1) & ~(size_t)1;                 // blocks of the second queued evaluation (16-B aligned)
  c->pin_rel2_off = c->pin_spec2_off + (size_t)E * MVICP_EDGE_BLOCK + ((size_t)E * MVICP_EDGE_BLOCK & 1);   // its relative transforms (E x 12), copied into d_rel in stream order
  MV_CHECK(ensure_pin(c, c->pin_rel2_off + (size_t)E * kEdgeRel + 8));
  {
    void* dp = nullptr;
    MV_HIP(hipHostGetDevicePointer(&dp, c->h_pin, 0));
    c->d_blocks_host = (double*)dp + c->pin_blocks_off;
    c->d_res_host = (double*)dp + c->pin_res_off;
    c->d_spec_host = (double*)dp + c->pin_spec_off;
    c->d_adev_host = (double*)dp + c->pin_adev_off;
    c->d_spec2_host = (double*)dp + c->pin_spec2_off;
  }
  c->spec_ready = false; c->spec_arm = false; c->bracket_counters_clean = false;
  c->spec2_ready = false; c->spec2_armed = false; c->last_cand_poses.clear();
  c->prev_xf.assign((size_t)E * 24, 0.0);
  if (!c->h_census) MV_HIP(hipHostMalloc((void**)&c->h_census, 8 * sizeof(unsigned long long), hipHostMallocDefault));
  return MVICP_OK;
} MVICP_GUARD_ABI

// everything earlier searches left behind (host-side bookkeeping only)
static void forget_history(mvicp_ctx* c) {
  const int E = c->E;
  c->nn_cache_valid = false; c->nn_cache_thresh = -1.f;
  c->prev_q.assign((size_t)E * 12, 0.0); c->prev_xf.assign((size_t)E * 24, 0.0);
  c->nn_cache_edge.assign(E, 0);                      // no seeds, no temporal cache: d_nn_idx / d_nn_lb are dead until the next search rewrites them
  c->auto_prev_dist = 0.0; c->auto_last_method = -1; c->last_rms = -1.0; c->prev_grid_kernel = false;
  c->corr_tie_seen = c->corr_far_seen = 1u;
  c->list_valid.assign(E, 0);                          // every list is re-compacted and re-gathered
  c->export_valid = false;
  c->qpos_valid.assign(E, 0);
  for (int e = 0; e < E; ++e) c->corr_epoch[e] = ++c->epoch_counter;
  c->sel_med1.assign(E, -1.0); c->sel_med2.assign(E, -1.0);
  c->spec_ready = false; c->spec_arm = false; c->spec_flags_valid = false;
  c->spec2_ready = false; c->spec2_armed = false; c->last_cand_poses.clear();
  c->have_corr = false;
  std::fill(c->h_count.begin(), c->h_count.end(), 0); std::fill(c->h_weight.begin(), c->h_weight.end(), 0.f);
}

int mvicp_reset_history(mvicp_ctx* c) try {
  MV_CHECK(bind(c));
  MV_HIP(hipStreamSynchronize(c->stream));
  forget_history(c);
  return MVICP_OK;
} MVICP_GUARD_ABI

// mvicp_correspond — cross-round state at a glance.  A search is a PURE FUNCTION of (clouds, graph, poses, fixed mask, cutoff): every field below
// only decides HOW FAST the same answer is found, and tests/test_gpu_parity.py::test_correspond_regime_transitions_match_a_fresh_context checks
// after every round of randomly perturbed registrations that the answer equals a fresh context's.  mvicp_reset_history / a poisoned exchange
// (forget_history) drop all of it.
//   field (common.h)              written by                               read by / meaning                                    invalidated by
//   nn_cache_valid, _thresh       this call's end (grid / BND tile round)  temporal cache may be consulted next search          cutoff change, set_correspondences, options, reset
//   nn_cache_edge[e]              this call's end (= active mask)          edge e was searched last time: seeds + bounds usable   fixed-mask change (per edge), reset
//   prev_q / prev_xf[e]           the per-edge loop below                  dM, dv of the temporal cache; bit-identical transform  every search rewrites them
//   list_valid[e], explicit_list  end of the NN stage / set_correspondences  the edge's compacted list may be maintained in place  set_correspondences, recompute_normals (dst), reset
//   sel_med1/2[e]                 after the wait below                     one-pass bracket select once the median has settled   inactive edge, set_correspondences, reset
//   auto_prev_dist, auto_last_method   the AUTO policy block               hand-over tile -> grid, "already handed over"          set_graph, reset
//   corr_tie_seen, corr_far_seen  after the wait below (own launches only) tie fix-up / far launch may be skipped at a fixed point  any search that is not bit-identical; reset
//   prev_grid_kernel              end of the NN stage                      corr_far_seen describes the last search                every search rewrites it
//   spec_flags_valid, spec_param/plane/robust   mvicp_optimize             arm the queued first evaluation of the NEXT solve      failed solve / search, spec_eval option, reset
//   spec_ready, spec_poses        end of this call                         evaluate_blocks may serve the first evaluation from it  consumed by the next evaluation, any list change
//   last_rms                      mvicp_optimize                           nn_cell policy only                                    consumed here
//   export_valid                  ensure_export                            h_export holds the lists as they are on the device     every search that can change a list, set_correspondences, reset
//   qpos_valid[e]                 end of the NN stage                      d_qpos / second / cd2 = the last search's result (export)  set_correspondences, failed search, reset
//   corr_epoch[e]                 end of this call                         callers skip copying a list whose epoch they hold      bumped unless the edge's inputs are bit-identical to last search's
static int correspond_once(mvicp_ctx* c, const double* poses, const unsigned char* fixed, float thresh, int nn_method, int* counts, float* weights, bool* tie_unresolved);

int mvicp_correspond(mvicp_ctx* c, const double* poses, const unsigned char* fixed, float thresh, int nn_method, int* counts, float* weights) try {
  MV_CHECK(bind(c));
  bool unresolved = false;
  MV_CHECK(correspond_once(c, poses, fixed, thresh, nn_method, counts, weights, &unresolved));
  if (!unresolved) return MVICP_OK;
  // Lazy tie trees (single rank): the search reported an exact distance tie on a target whose reference-equivalent tree does not exist yet, so that
  // query still carries the kernels' own rule (lowest index).  Build the trees of every searched target now — once per cloud, like the reference's
  // lazily built index — forget what this search left behind and search again: this time the fix-up decides the ties the way nanoflann does.
  std::vector<int> need;
  for (int e = 0; e < c->E; ++e) if (c->active[e] && !c->frames[c->edst[e]].has_tie) need.push_back(c->edst[e]);
  MV_HIP(hipStreamSynchronize(c->stream));
  MV_CHECK(ensure_tie_trees(c, need));
  forget_history(c);
  unresolved = false;
  MV_CHECK(correspond_once(c, poses, fixed, thresh, nn_method, counts, weights, &unresolved));
  if (unresolved) { set_error("tie fix-up still without a tree after building them"); return MVICP_ERR_INTERNAL; }
  return MVICP_OK;
} MVICP_GUARD_ABI

static int correspond_once(mvicp_ctx* c, const double* poses, const unsigned char* fixed, float thresh, int nn_method, int* counts, float* weights, bool* tie_unresolved) {
  if (!poses) { set_error("poses is null"); return MVICP_ERR_ARG; }
  if (c->E == 0) { set_error("no graph: call mvicp_set_graph first"); return MVICP_ERR_STATE; }
  const int E = c->E;
  HostScope hs_all(c, "host.correspond");
  // per-edge query transforms (frame.cpp:117-118,131,136) + active mask (frame.cpp:93)
  std::vector<int> nsrc(E, 0);
  std::vector<char> same_edge(E, 0);   // the edge's query transform is bit-identical to last search's (and the temporal cache is on for it)
  // the edge's list after this search IS last search's, bit for bit: a search is a pure function of (clouds, transform, cutoff), and all three are
  // last search's — whichever kernel runs.  Such an edge keeps its epoch (mvicp_correspondence_epochs), and if every edge does, the export too.
  std::vector<char> unchanged(E, 0);
  const bool hist_ok = c->have_corr && (int)c->nn_cache_edge.size() == E && c->nn_cache_thresh == thresh && (int)c->qpos_valid.size() == E;
  std::vector<double> eps_ratio;       // per active edge: displacement bound of its queries since the last search / guard band mu
  bool same_active_set = (int)c->nn_cache_edge.size() == E;   // the set of searched edges is last search's (a changed `fixed` mask changes it)
  double* hx = c->h_pin;
  for (int e = 0; e < E; ++e) {
    c->active[e] = c->owned[e] && !(fixed && fixed[c->esrc[e]]);
    if (same_active_set && (c->active[e] != 0) != (c->nn_cache_edge[e] != 0)) same_active_set = false;
    nsrc[e] = c->active[e] ? c->frames[c->esrc[e]].n : 0;
    const double* Ps = poses + 16 * (size_t)c->esrc[e];
    const double* Pd = poses + 16 * (size_t)c->edst[e];
    double* x = hx + (size_t)e * kEdgeXf;
    double Rd[9];
    for (int j = 0; j < 3; ++j)
      for (int i = 0; i < 3; ++i) { x[i + 3 * j] = Ps[i + 4 * j]; Rd[i + 3 * j] = Pd[i + 4 * j]; }
    for (int i = 0; i < 3; ++i) { x[9 + i] = Ps[12 + i]; x[21 + i] = Pd[12 + i]; }
    inverse3(Rd, x + 12);
    // temporal cache: q = M p + v with M = Rd^-1 Rs, v = Rd^-1 (ts - td).  Between two searches every query of the edge
    // moves by at most ||dM||_F max|p| + |dv|  (+ a rounding allowance far above the 1e-16-relative error of the fp64 map).
    double Mq[12];
    for (int j = 0; j < 3; ++j)
      for (int i = 0; i < 3; ++i) Mq[i + 3 * j] = x[12 + i] * x[0 + 3 * j] + x[12 + i + 3] * x[1 + 3 * j] + x[12 + i + 6] * x[2 + 3 * j];
    const double dt[3] = {x[9] - x[21], x[10] - x[22], x[11] - x[23]};
    for (int i = 0; i < 3; ++i) Mq[9 + i] = x[12 + i] * dt[0] + x[12 + i + 3] * dt[1] + x[12 + i + 6] * dt[2];
    double* pq = &c->prev_q[(size_t)e * 12];
    double scale = 0.0;
    for (int k = 0; k < 12; ++k) { x[25 + k] = Mq[k] - pq[k]; scale = std::max(scale, std::fabs(Mq[k])); }
    const double rmax = c->frames[c->esrc[e]].max_norm;
    if (c->active[e]) {
      // how far this edge's queries can have moved since the last search, in units of the bounds-leaving builds' guard band: what decides the
      // temporal-cache hit rate of a cache-aware round before it runs (a hit needs d_new < d_old + mu - eps)
      double fro = 0.0;
      for (int k = 0; k < 9; ++k) fro += x[25 + k] * x[25 + k];
      const double eps_e = std::sqrt(fro) * rmax + std::sqrt(x[34] * x[34] + x[35] * x[35] + x[36] * x[36]);
      const double mu_e = c->tile_mu * c->frames[c->edst[e]].grid.cell;
      eps_ratio.push_back(mu_e > 0.0 ? eps_e / mu_e : 1e300);
    }
    const bool cache_on = c->nn_cache_valid && c->nn_cache_enable && c->active[e] && (int)c->nn_cache_edge.size() == E && c->nn_cache_edge[e] &&
                          c->nn_cache_thresh == thresh;
    // allowance for the rounding of the fp64 query map itself (both evaluations): ~1e-16 (|M||p| + |v|), taken 1e4 times larger.
    // A transform that is BIT-IDENTICAL to last search's (a converged registration: the LM ends without stepping) reproduces every
    // query bit for bit: no allowance, and dM = dv = 0 below, so the kernel sees eps == 0 and re-verifies without rewriting anything.
    double* pxf = &c->prev_xf[(size_t)e * 24];
    if (!c->owned[e]) unchanged[e] = 1;   // (another rank's edge: nothing of it lives here)
    else if (c->active[e]) unchanged[e] = hist_ok && c->nn_cache_edge[e] && c->qpos_valid[e] && !c->explicit_list[e] && std::memcmp(pxf, x, sizeof(double) * 24) == 0;
    else unchanged[e] = hist_ok && !c->nn_cache_edge[e] && !c->explicit_list[e] && c->h_count[e] == 0;   // not searched now, not searched then: stays empty
    const bool same_xf = cache_on && std::memcmp(pxf, x, sizeof(double) * 24) == 0;
    x[24] = cache_on ? (same_xf ? 0.0 : 1e-12 * (scale * (rmax + 1.0) + 1.0)) : -1.0;
    same_edge[e] = same_xf;
    std::memcpy(pxf, x, sizeof(double) * 24);
    for (int k = 37; k < kEdgeXf; ++k) x[k] = 0.0;
    std::memcpy(pq, Mq, sizeof(Mq));
  }
  // region 1 of the control block (xf | nsrc | dirty) goes up in ONE copy once the dirty flags are known (below)
  int* hn = reinterpret_cast<int*>(hx + (size_t)E * kEdgeXf);
  int* hd = hn + E;
  std::memcpy(hn, nsrc.data(), sizeof(int) * E);

  // Failure semantics with N > 1 ranks.  A search has ONE collective; a rank that failed locally before it (an allocation,
  // an upload, a launch) used to return at once — and its peers blocked in the collective forever.  Now every local step up to the exchange
  // only records its status (st_local); a rank whose share cannot be delivered still ENTERS the collective, with the "armed" slot of the
  // exchanged buffer poisoned (NaN): the sum is NaN on every rank, every rank sees it after its own wait, drops its cross-round state and
  // returns an error from this same call — the failing rank its own status, the others MVICP_ERR_COMM.  (A device that no longer executes
  // anything cannot poison or exchange: that stays fatal for the job, as any collective library has it.)
  int st_local = MVICP_OK;
  if (c->tie_rule && !(c->tie_lazy && c->world == 1)) {   // a cloud uploaded after mvicp_set_graph has no tree yet (the reference builds its index lazily as well, frame.cpp:188-193)
    std::vector<int> need;
    for (int e = 0; e < E; ++e) if (c->active[e] && !c->frames[c->edst[e]].has_tie) need.push_back(c->edst[e]);
    if (!need.empty()) st_local = ensure_tie_trees(c, need);
  }
  const double bound = sqrt_bound((double)thresh);
  double t_mark = now_ms();
  auto mark = [&](const char* nm) { if (c->profile) { const double t = now_ms(); ProfEntry& pe = c->prof[nm]; pe.ms += t - t_mark; pe.launches += 1; t_mark = t; } };
  mark("host.corr.setup");
  int method = nn_method;
  if (method == MVICP_NN_AUTO) {
    // Two exact kernels, two regimes.  While the poses still move by more than the spacing of the points, nearly every
    // query needs a fresh search and the wave-cooperative tile kernel (seeded with last round's neighbours) is fastest.
    // Once the registration settles, the hash-grid kernel wins: its temporal cache answers a query whose neighbour provably
    // did not change with one distance evaluation.  The switch costs one uncached grid round, so it is made when the
    // median correspondence distance has stopped contracting (last > half of the one before) — or, with a single round
    // of history, when it is already well inside a hash cell.
    method = MVICP_NN_TILE;
    if (c->have_corr) {
      double dist = 0.0, cell = 0.0;
      int m = 0;
      for (int e = 0; e < E; ++e)
        if (c->h_count[e] > 0) { dist += c->h_weight[e] / 1.5; cell += c->frames[c->edst[e]].grid.cell; ++m; }
      if (m > 0) {
        dist /= m; cell /= m;
        const bool settled = c->auto_prev_dist > 0.0 ? dist > c->auto_settle * c->auto_prev_dist : dist < 0.5 * cell;
        if (c->nn_cell) {
          // cell-staging grid kernel: it resolves a query whose neighbour is within about a cell of it (home block + ball) at a cost
          // that follows that distance, not the cache hit rate — so it takes over as soon as the queries are EXPECTED to land that
          // close: the smaller of last round's median distance and the residual the LM solve has just left (RMS over all
          // correspondences, with a floor of a fraction of a cell for the sampling of the surface)
          double pred = dist;
          if (c->last_rms >= 0.0) pred = std::min(pred, std::sqrt(c->last_rms * c->last_rms + 0.09 * cell * cell));
          if (pred < c->auto_switch * cell || c->auto_last_method == MVICP_NN_GRID) method = MVICP_NN_GRID;
        } else if (dist < 1.5 * cell && (settled || c->auto_last_method == MVICP_NN_GRID)) method = MVICP_NN_GRID;
        c->auto_prev_dist = dist;
      }
    }
  }
  if (method == MVICP_NN_GRID || method == MVICP_NN_TILE) {
    for (int e = 0; e < E; ++e)
      if (c->active[e] && (!c->frames[c->edst[e]].has_grid || !c->frames[c->esrc[e]].has_grid)) method = MVICP_NN_BRUTE;
  }
  // The round in which AUTO hands over to the grid kernel would be an UNCACHED grid round (every query searched through the hash:
  // 2.4-2.6 ms on cfg4, the slowest round after the first).  Instead the tile kernel runs once more in its BND build (~1.2 ms), which
  // also leaves the per-query lower bounds the temporal cache needs; the grid kernel takes over one round later, with cache hits.
  bool tile_lb = false, handed_over = false;
  if (nn_method == MVICP_NN_AUTO && method == MVICP_NN_GRID && c->auto_last_method == MVICP_NN_TILE && c->tile_bounds >= 1 &&
      c->nn_cache_enable && !c->nn_cell) { method = MVICP_NN_TILE; tile_lb = true; handed_over = true; }
  if (method == MVICP_NN_TILE && c->tile_bounds >= 2 && c->nn_cache_enable) tile_lb = true;
  // Cache-aware tile rounds (round 3).  After the hand-over the grid kernel answers cache hits at streaming speed but pays 1.5-2 ns for
  // every isolated miss (per-lane hash probing), which dominates the rounds in which the poses still move a little (hit rates 74-99 %).
  // Those rounds run the tile kernel's bounds-leaving build with the cache check as its prologue instead: a missed lane is searched by
  // its wave, cooperatively.  Once every transform is bit-identical to last search's (the fixed point) the grid kernel's verify pass is
  // the cheaper one and takes over.
  bool tile_cached = false;
  bool all_same = true;   // every active edge's query transform is bit-identical to last search's
  {
    for (int e = 0; e < E; ++e) if (c->active[e] && !same_edge[e]) all_same = false;
    if (nn_method == MVICP_NN_AUTO && method == MVICP_NN_GRID && !handed_over && c->tile_cache && c->tile_bounds >= 1 && c->nn_cache_valid &&
        c->nn_cache_enable && !c->nn_cell && !all_same) { method = MVICP_NN_TILE; tile_lb = true; tile_cached = true; handed_over = true; }
    // experiment (round 6, "tile_cache" = 2): the cache prologue in EVERY tile round that follows a bounds-leaving one (with "tile_bounds" = 2: from
    // round 3 on), not only after the hand-over — measures how early the temporal cache starts to hit (profiles/r06_tile_ab.txt)
    if (method == MVICP_NN_TILE && !tile_cached && c->tile_cache >= 2 && tile_lb && c->nn_cache_valid && c->nn_cache_enable && !all_same) tile_cached = true;
  }
  {
    // an edge may keep last round's compacted list only if a kernel that checks every query's acceptance and patches changed
    // neighbours in place runs (the grid kernel, the tile kernel: nn_list.h) and the list on the device really is last round's result
    // for this edge
    std::vector<int> dirty(E, 1);
    if ((method == MVICP_NN_TILE || (method == MVICP_NN_GRID && !c->nn_tree_only && !c->nn_skip_far)) && c->list_reuse)
      for (int e = 0; e < E; ++e) if (c->active[e] && c->list_valid[e]) dirty[e] = 0;
    std::memcpy(hd, dirty.data(), sizeof(int) * E);
  }
  // One-pass bracket select (corr.hip) instead of the 3-pass radix select: only when EVERY active edge of this rank has a
  // median that has settled (last two rounds within 0.1 %); the bracket is +-0.6 % in d2 around the last one.
  bool use_bracket = c->sel_bracket && c->have_corr;
  {
    double* lohi = hx + (size_t)E * (kEdgeXf + 1);
    for (int e = 0; e < E; ++e) {
      lohi[2 * e] = lohi[2 * e + 1] = 0.0;
      if (!c->active[e]) continue;
      const double m1 = c->sel_med1[e], m2 = c->sel_med2[e];
      if (!(m1 > 0.0 && m2 > 0.0 && std::fabs(m1 - m2) <= 0.001 * m1)) { use_bracket = false; continue; }
      lohi[2 * e] = m1 * 0.994; lohi[2 * e + 1] = m1 * 1.006;
    }
  }
  // A search whose every edge has last round's exact transform and a valid list reproduces every query bit for bit: no list can
  // change, so the (data-dependent, early-exiting) list-maintenance kernels — dirty-flag reduction, compaction, gather — are not
  // even launched.
  // (not with the cell-staging variant: nn_cell_kernel has no bit-identical-query shortcut, so a fixed-point round is not a pure no-op there)
  bool nothing_can_change = method == MVICP_NN_GRID && c->list_reuse && !c->nn_tree_only && !c->nn_skip_far && same_active_set && !c->nn_cell;
  for (int e = 0; e < E && nothing_can_change; ++e)
    if (c->active[e] && !(same_edge[e] && hd[e] == 0)) nothing_can_change = false;
  c->skip_dirty_reduce = nothing_can_change;
  // ... and the tie fix-up (nn_tie.hip) neither, if last round's search reported no tie: the same queries meet the same targets
  c->tie_skip = nothing_can_change && c->corr_tie_seen == 0u;
  c->far_skip = nothing_can_change && c->corr_far_seen == 0u && c->prev_grid_kernel && method == MVICP_NN_GRID;
  const bool tie_launched = c->tie_rule && !c->tie_skip, far_launched = method == MVICP_NN_GRID && !c->far_skip;
  // Speculative first evaluation of the solve that follows (see common.h).  Every rank decides for itself (its own last solve set
  // the flags); with N > 1 ranks the decisions are SUMMED in the "armed" slot of the one exchanged buffer and the queued blocks are
  // used only if every rank armed — a rank that did not arm still takes part in the same collective with the same size.  (A rank that fails
  // LOCALLY before the exchange — a launch error — returns without it and its peers block in the collective: a failed launch is fatal for the job.)
  const bool exchange = c->comm != nullptr || c->ar_fn != nullptr;
  c->spec_ready = false; c->spec2_ready = false;
  c->spec_arm = c->spec_enable && c->spec_flags_valid;
  if (c->spec_arm && c->spec_plane)
    for (int e = 0; e < E; ++e) if (!(fixed && fixed[c->esrc[e]]) && c->frames[c->edst[e]].n > 0 && c->frames[c->edst[e]].grid.snor == nullptr) c->spec_arm = false;
  const size_t nb = (size_t)E * MVICP_EDGE_BLOCK, ntail = 3 * (size_t)E + 1;   // tail = (count, median d2) x E | armed | a x E
  c->d_res_target = exchange ? c->d_out + nb : nullptr;
  c->d_a_check = exchange ? c->d_out + nb + 2 * (size_t)E + 1 : c->d_adev_host;   // where the select kernels copy the scales they derive
  size_t upload_doubles = c->ctl_r1;
  if (c->spec_arm) {
    // the solve evaluates at x_to_pose(pose_to_x(P)) (host/lm.cpp): the same round trip here, so the poses match bit for bit.  The
    // relative transforms of that evaluation ride on the control-block upload (region 2 follows region 1; the SoftLOne scales
    // behind them are written on the device by the select kernels later in the stream)
    c->spec_poses.resize(16 * (size_t)c->n_frames);
    double xp[7];
    for (int k = 0; k < c->n_frames; ++k) { se3::pose_to_x(c->spec_param, poses + 16 * (size_t)k, xp); se3::x_to_pose(c->spec_param, xp, &c->spec_poses[16 * (size_t)k]); }
    fill_rel(c, c->spec_poses.data());
    c->spec_q_plane = c->spec_plane; c->spec_q_robust = c->spec_robust;
    upload_doubles = c->ctl_r2_off + (size_t)E * kEdgeRel;
  }
  if (method != MVICP_NN_BRUTE && method != MVICP_NN_GRID && method != MVICP_NN_TILE) { set_error("unknown nn_method %d", nn_method); return MVICP_ERR_ARG; }   // (an argument error: the same on every rank)
  bool all_unchanged = true;
  for (int e = 0; e < E; ++e) { if (!unchanged[e]) all_unchanged = false; if (c->active[e]) c->qpos_valid[e] = 0; }   // (set again once the NN stage is queued)
  auto local_search = [&]() -> int {
  MV_HIP(hipMemcpyAsync(c->d_ctl, hx, sizeof(double) * upload_doubles, hipMemcpyHostToDevice, c->stream));
  c->far_narrow = nn_method == MVICP_NN_AUTO && method == MVICP_NN_GRID && c->nn_cache_valid && c->nn_cache_enable && all_same;   // the fixed point: (almost) every query is a cache hit
  if (c->fault_inject > 0 && --c->fault_inject == 0) { set_error("injected launch failure (option fault_inject)"); return MVICP_ERR_HIP; }   // tests: a local failure before the exchange
  if (method == MVICP_NN_BRUTE) MV_CHECK(launch_nn_brute_edges(c));
  else if (method == MVICP_NN_GRID) MV_CHECK(launch_nn_grid_edges(c, bound));
  else {
    // Cache-aware rounds: which build?  With most lanes finished by the cache, nn_tile_kernel (per-lane box tests, miss_block) wins; with hit rates
    // below ~80 % the matrix-pipe build does (cfg4_partial rounds 5-13: -4 ... -10 %, profiles/r06_tile_ab.txt).  The hit rate is not known before
    // the launch, but what decides it is: the median displacement bound of the queries in units of the guard band (eps / mu).
    c->cached_on_mfma = false;
    if (tile_cached && c->tile_mfma == 1 && c->cache_mfma_ratio > 0.0 && !eps_ratio.empty()) {
      std::nth_element(eps_ratio.begin(), eps_ratio.begin() + eps_ratio.size() / 2, eps_ratio.end());
      const double med = eps_ratio[eps_ratio.size() / 2];
      if (c->profile) { ProfEntry& pe = c->prof["auto.eps_over_mu"]; pe.ms = med; pe.launches += 1; }   // (observable: the last cache-aware round's ratio)
      c->cached_on_mfma = med > c->cache_mfma_ratio;
    }
    MV_CHECK(launch_nn_tile_edges(c, bound, tile_lb, tile_cached, c->list_reuse));
  }
  c->tie_skip = false; c->far_skip = false;
  c->prev_grid_kernel = method == MVICP_NN_GRID;
  // only the grid kernel and the tile kernel's BND build leave the per-query lower bounds the temporal cache needs; the cutoff must
  // not change either
  c->last_rms = -1.0;   // consumed: only a solve that follows THIS search may predict the next one
  c->nn_cache_valid = ((method == MVICP_NN_GRID) && !c->nn_tree_only && !c->nn_skip_far) || tile_lb;
  c->nn_cache_thresh = thresh;
  c->auto_last_method = handed_over ? MVICP_NN_GRID : method;   // (the policy's "already handed over" state)
  c->nn_cache_edge.assign(c->active.begin(), c->active.end());
  for (int e = 0; e < E; ++e) { c->list_valid[e] = c->active[e]; if (c->owned[e]) c->qpos_valid[e] = c->active[e]; if (c->active[e]) c->explicit_list[e] = 0; }
  if (!all_unchanged) c->export_valid = false;   // (a search that reproduces every list leaves the exported copy what it is)

  mark("host.corr.nn_launch");
  if (!nothing_can_change) {
    MV_CHECK(launch_compact(c, bound));
    MV_CHECK(launch_gather_stream(c));
  }
  return use_bracket ? launch_select_bracket(c) : launch_select_median(c);
  };   // local_search
  if (st_local == MVICP_OK) st_local = local_search();
  if (st_local != MVICP_OK && !exchange) { c->spec_flags_valid = false; return st_local; }
  // the queued first evaluation (its relative transforms went up with the control block).  With N > 1 ranks: ONE collective per
  // search — [E x 91 blocks | (count, median d2) x E | armed | a x E], always the full buffer — and ONE wait.
  int st_q = MVICP_OK;
  char local_msg[sizeof(g_err)] = "";
  if (exchange) {
    c->lin_out = c->d_out;
    if (st_local == MVICP_OK) {
      if (c->spec_arm) st_local = launch_linearize(c, c->spec_q_plane, c->spec_q_robust);
      else {
        // this rank queued no evaluation: its share of the block region and of the scale slots is ZERO, not whatever the last exchange left
        // there (already-summed values would be summed again round after round).  Every rank sends the same buffer size whatever it decided.
        if (hipMemsetAsync(c->d_out, 0, sizeof(double) * nb, c->stream) != hipSuccess ||
            hipMemsetAsync(c->d_out + nb + 2 * (size_t)E + 1, 0, sizeof(double) * (size_t)E, c->stream) != hipSuccess) { set_error("hipMemsetAsync of the exchange buffer failed"); st_local = MVICP_ERR_HIP; }
      }
    }
    if (st_local != MVICP_OK) {
      // this rank's share cannot be delivered: enter the collective all the same, with the armed slot poisoned (eight 0xFF bytes = a NaN)
      std::memcpy(local_msg, g_err, sizeof(local_msg));
      if (hipMemsetAsync(c->d_out + nb + 2 * (size_t)E, 0xFF, sizeof(double), c->stream) != hipSuccess) {
        set_error("%s — and the exchange buffer could not be poisoned: the peers of this rank will block in the collective", local_msg);
        c->spec_flags_valid = false; c->spec_arm = false;
        return st_local;
      }
    }
    { ProfScope pc(c, "comm", 8.0 * (double)(nb + ntail)); st_q = comm_allreduce_sum(c, c->d_out, nb + ntail); }
    if (st_q == MVICP_OK && hipMemcpyAsync(c->h_pin + c->pin_spec_off, c->d_out, sizeof(double) * (nb + ntail), hipMemcpyDeviceToHost, c->stream) != hipSuccess) {
      set_error("hipMemcpyAsync of the exchanged buffer failed"); st_q = MVICP_ERR_HIP;
    }
  } else if (c->spec_arm) {
    c->lin_out = c->d_spec_host;
    st_q = launch_linearize(c, c->spec_q_plane, c->spec_q_robust);
    // second queued evaluation: this search's poses are last search's bit for bit, so the solve that follows will ask for last solve's candidate again
    c->spec2_armed = false;
    if (st_q == MVICP_OK && c->spec2_enable && all_same && same_active_set && c->have_corr && c->last_cand_poses.size() == 16 * (size_t)c->n_frames &&
        c->last_cand_plane == c->spec_q_plane && c->last_cand_robust == c->spec_q_robust) {
      fill_rel_into(c, c->last_cand_poses.data(), c->h_pin + c->pin_rel2_off);
      if (hipMemcpyAsync(c->d_rel, c->h_pin + c->pin_rel2_off, sizeof(double) * (size_t)E * kEdgeRel, hipMemcpyHostToDevice, c->stream) == hipSuccess) {
        c->lin_out = c->d_spec2_host;
        st_q = launch_linearize(c, c->spec_q_plane, c->spec_q_robust);
        if (st_q == MVICP_OK) { c->spec2_armed = true; c->spec2_poses = c->last_cand_poses; c->spec2_plane = c->spec_q_plane; c->spec2_robust = c->spec_q_robust; }
      }
    }
  }
  mark("host.corr.post_launch");
  // (count, median d2) per edge arrive in mapped host memory, written by select_final_kernel (single rank), or summed over ranks
  // behind the blocks; weight = (float)(1.5 * sqrt(median d2))  (frame.cpp:168-176)
  if (st_q == MVICP_OK) st_q = stream_wait(c);
  if (st_q != MVICP_OK) { c->spec_flags_valid = false; c->spec_arm = false; return st_q; }
  if (exchange && std::isnan(c->h_pin[c->pin_spec_off + nb + 2 * (size_t)E])) {
    // some rank (this one if st_local says so) entered the collective without its share: nothing of this search is usable anywhere.  Every
    // rank forgets what earlier searches left behind — the next search is a first search on all of them, so their states agree again.
    forget_history(c);
    c->census_pending = false;
    if (st_local != MVICP_OK) { set_error("%s", local_msg); return st_local; }
    set_error("a peer rank failed before the exchange of this search (poisoned exchange buffer); nothing was updated, cross-round state dropped on every rank");
    return MVICP_ERR_COMM;
  }
  census_resolve(c);
  // what THIS search's tie fix-up / far launch reported (a skipped launch keeps last search's zero): the next search's skip decisions
  if (tie_launched) c->corr_tie_seen = c->h_tie_seen ? *c->h_tie_seen : 1u;
  if (tie_launched && c->h_tie_seen && c->h_tie_seen[1] != 0u) *tie_unresolved = true;
  if (far_launched) c->corr_far_seen = c->h_far_seen ? *c->h_far_seen : 1u;
  else if (method != MVICP_NN_GRID) c->corr_far_seen = 1u;
  mark("host.corr.wait");
  const double* hr = exchange ? c->h_pin + c->pin_spec_off + nb : c->h_pin + c->pin_res_off;
  bool spec_bad = false;
  {
    // a median that left its bracket is flagged -1 in its slot; with N > 1 ranks every rank sees the summed slots, so every rank takes
    // the same decision: full select for everything (rare once the registration has settled) and a second, tail-only exchange
    bool redo = false;
    for (int e = 0; e < E; ++e) if (hr[2 * e + 1] < 0.0) redo = true;
    if (redo) {
      spec_bad = true;   // the queued evaluation used the scales of the failed select
      int st_r = launch_select_median(c);
      if (exchange) {
        int st_x = MVICP_OK;
        if (st_r != MVICP_OK) {   // same protocol as above: the second (tail-only) collective is entered with a poisoned armed slot
          std::memcpy(local_msg, g_err, sizeof(local_msg));
          if (hipMemsetAsync(c->d_out + nb + 2 * (size_t)E, 0xFF, sizeof(double), c->stream) != hipSuccess) { c->spec_flags_valid = false; c->spec_arm = false; return st_r; }
        }
        { ProfScope pc(c, "comm", 8.0 * (double)ntail); st_x = comm_allreduce_sum(c, c->d_out + nb, ntail); }
        if (st_x == MVICP_OK && hipMemcpyAsync(c->h_pin + c->pin_spec_off + nb, c->d_out + nb, sizeof(double) * ntail, hipMemcpyDeviceToHost, c->stream) != hipSuccess) {
          set_error("hipMemcpyAsync of the exchanged tail failed"); st_x = MVICP_ERR_HIP;
        }
        if (st_x == MVICP_OK) st_x = stream_wait(c);
        if (st_x != MVICP_OK) { c->spec_flags_valid = false; c->spec_arm = false; return st_x; }
        if (std::isnan(hr[2 * (size_t)E])) {
          forget_history(c);
          if (st_r != MVICP_OK) { set_error("%s", local_msg); return st_r; }
          set_error("a peer rank failed before the second exchange of this search (poisoned exchange buffer); cross-round state dropped on every rank");
          return MVICP_ERR_COMM;
        }
      } else {
        if (st_r == MVICP_OK) st_r = stream_wait(c);
        if (st_r != MVICP_OK) { c->spec_flags_valid = false; c->spec_arm = false; return st_r; }
      }
    }
  }
  for (int e = 0; e < E; ++e) {
    if (!(c->owned[e] && c->active[e])) { c->sel_med1[e] = c->sel_med2[e] = -1.0; continue; }
    c->sel_med2[e] = c->sel_med1[e];
    c->sel_med1[e] = hr[2 * e] > 0 ? hr[2 * e + 1] : -1.0;
  }
  std::vector<double> pack(2 * (size_t)E, 0.0);
  for (int e = 0; e < E; ++e)
    if (exchange || (c->owned[e] && c->active[e])) { pack[2 * e] = hr[2 * e]; pack[2 * e + 1] = hr[2 * e] > 0 ? hr[2 * e + 1] : 0.0; }
  if (c->spec_arm) {
    // trust the queued evaluation only if (i) every rank queued one, and (ii) the scales the device derived from the medians are the
    // host's (IEEE sqrt) bit for bit — checked on ALL edges from exchanged data, so every rank reaches the same verdict
    const double* ad = exchange ? hr + 2 * (size_t)E + 1 : c->h_pin + c->pin_adev_off;
    if (hr[2 * (size_t)E] != (double)(exchange ? c->world : 1)) spec_bad = true;
    for (int e = 0; e < E && !spec_bad; ++e)
      if (exchange || (c->owned[e] && c->active[e])) {
        const double a_host = pack[2 * e] > 0 ? (double)(float)(std::sqrt(pack[2 * e + 1]) * 1.5) : 0.0;
        if (ad[e] != a_host) spec_bad = true;
      }
  }
  c->spec_ready = c->spec_arm && !spec_bad;
  c->spec2_ready = c->spec_ready && c->spec2_armed;   // (same scales, same lists: valid whenever the first one is)
  c->spec2_armed = false;
  c->spec_arm = false;
  double* ha = c->h_pin + c->ctl_r2_off + (size_t)E * kEdgeRel;   // `a` slice of region 2: uploaded with rel by the next evaluation
  for (int e = 0; e < E; ++e) {
    c->h_count[e] = (int)pack[2 * e];
    const double nth = std::sqrt(pack[2 * e + 1]);
    c->h_weight[e] = c->h_count[e] > 0 ? (float)(nth * 1.5) : 0.f;
    ha[e] = (double)c->h_weight[e];
    if (counts) counts[e] = c->h_count[e];
    if (weights) weights[e] = c->h_weight[e];
  }
  for (int e = 0; e < E; ++e) if (!unchanged[e]) c->corr_epoch[e] = ++c->epoch_counter;
  c->have_corr = true;
  mark("host.corr.finish");
  if (c->profile) prof_collect_lazy(c);
  return MVICP_OK;
}

// every exportable edge's list of the last search, un-sorted on the device into the reference's layout and copied once (export.hip)
static int ensure_export(mvicp_ctx* c, bool wait = true) {
  if (!c->export_valid) {
    HostScope hs(c, "host.export");
    MV_CHECK(launch_export(c));
    c->export_valid = true;
  }
  if (wait && c->export_in_flight) {
    MV_CHECK(stream_wait(c));
    c->export_in_flight = false;
    if (c->profile) prof_collect_lazy(c);
  }
  return MVICP_OK;
}

int mvicp_map_correspondences(mvicp_ctx* c, const mvicp_corr** triples, const long long** offsets) try {
  MV_CHECK(bind(c));
  if (!triples || !offsets) { set_error("null output"); return MVICP_ERR_ARG; }
  if (!c->have_corr) { set_error("no correspondences yet"); return MVICP_ERR_STATE; }
  MV_CHECK(ensure_export(c));
  *triples = (const mvicp_corr*)c->h_export;
  *offsets = c->export_off.data();
  return MVICP_OK;
} MVICP_GUARD_ABI

int mvicp_map_correspondences_async(mvicp_ctx* c, const mvicp_corr** triples, const long long** offsets) try {
  MV_CHECK(bind(c));
  if (!triples || !offsets) { set_error("null output"); return MVICP_ERR_ARG; }
  if (!c->have_corr) { set_error("no correspondences yet"); return MVICP_ERR_STATE; }
  MV_CHECK(ensure_export(c, false));
  *triples = (const mvicp_corr*)c->h_export;
  *offsets = c->export_off.data();
  return MVICP_OK;
} MVICP_GUARD_ABI

int mvicp_wait_correspondences(mvicp_ctx* c, int edge) try {
  MV_CHECK(bind(c));
  if (edge < 0 || edge >= c->E) { set_error("edge %d out of range", edge); return MVICP_ERR_ARG; }
  if (!c->export_valid) { set_error("no export in flight: call mvicp_map_correspondences_async after the search"); return MVICP_ERR_STATE; }
  if (!c->export_in_flight || c->export_chunks == 0) return MVICP_OK;
  const int k = c->export_edge_chunk[edge];
  MV_HIP(hipEventSynchronize(c->export_events[k]));
  if (k == c->export_chunks - 1) c->export_in_flight = false;   // (the last chunk's event is behind every other one on the stream)
  return MVICP_OK;
} MVICP_GUARD_ABI

int mvicp_correspondence_epochs(mvicp_ctx* c, const unsigned long long** epochs) try {
  MV_CHECK(bind(c));
  if (!epochs) { set_error("null output"); return MVICP_ERR_ARG; }
  if (c->E == 0) { set_error("no graph"); return MVICP_ERR_STATE; }
  *epochs = c->corr_epoch.data();
  return MVICP_OK;
} MVICP_GUARD_ABI

int mvicp_get_correspondences(mvicp_ctx* c, int edge, int cap, int* first, int* second, double* dist) try {
  MV_CHECK(bind(c));
  if (edge < 0 || edge >= c->E) { set_error("edge %d out of range", edge); return MVICP_ERR_ARG; }
  if (!c->have_corr) { set_error("no correspondences yet"); return MVICP_ERR_STATE; }
  if (!c->owned[edge]) { set_error("edge %d is owned by another rank", edge); return MVICP_ERR_STATE; }
  const int n = c->h_count[edge];
  if (cap < n) { set_error("capacity %d < count %d", cap, n); return MVICP_ERR_ARG; }
  if (!c->explicit_list[edge]) {
    // a list built by mvicp_correspond: slice of the one export of this search (first call after a search runs it for ALL edges)
    if (n == 0) return 0;
    MV_CHECK(ensure_export(c));
    const mvicp_corr* t = (const mvicp_corr*)c->h_export + c->export_off[edge];
    if (c->export_off[edge + 1] - c->export_off[edge] != n) { set_error("export of edge %d holds %lld triples, expected %d", edge, c->export_off[edge + 1] - c->export_off[edge], n); return MVICP_ERR_STATE; }
    for (int i = 0; i < n; ++i) {
      if (first) first[i] = t[i].first;
      if (second) second[i] = t[i].second;
      if (dist) dist[i] = t[i].dist;
    }
    return n;
  }
  // an explicit list (mvicp_set_correspondences: any order, repeats allowed) has no per-query positions: host-side un-sort
  const size_t off = (size_t)c->cap_off[edge];
  MV_HIP(hipStreamSynchronize(c->stream));
  std::vector<int> a(n), b(n);
  std::vector<double> d(n);
  if (n) {
    MV_HIP(hipMemcpy(a.data(), c->d_first + off, sizeof(int) * n, hipMemcpyDeviceToHost));
    MV_HIP(hipMemcpy(b.data(), c->d_second + off, sizeof(int) * n, hipMemcpyDeviceToHost));
    MV_HIP(hipMemcpy(d.data(), c->d_cd2 + off, sizeof(double) * n, hipMemcpyDeviceToHost));
  }
  const std::vector<int>& so = c->frames[c->esrc[edge]].grid.h_order;
  const std::vector<int>& dorder = c->frames[c->edst[edge]].grid.h_order;
  std::vector<int> perm(n);
  for (int i = 0; i < n; ++i) { a[i] = so[a[i]]; b[i] = dorder[b[i]]; perm[i] = i; }
  std::stable_sort(perm.begin(), perm.end(), [&](int x, int y) { return a[x] < a[y]; });
  for (int i = 0; i < n; ++i) {
    const int k = perm[i];
    if (first) first[i] = a[k];
    if (second) second[i] = b[k];
    if (dist) dist[i] = std::sqrt(d[k]);  // frame.cpp:139 pointDist = sqrt(pointDistSquared), IEEE on the host
  }
  return n;
} MVICP_GUARD_ABI

int mvicp_set_correspondences(mvicp_ctx* c, int edge, int n, const int* first, const int* second, float weight) try {
  MV_CHECK(bind(c));
  if (edge < 0 || edge >= c->E) { set_error("edge %d out of range", edge); return MVICP_ERR_ARG; }
  if (!c->owned[edge]) { set_error("edge %d is owned by another rank", edge); return MVICP_ERR_STATE; }
  const int ns = c->frames[c->esrc[edge]].n, nd = c->frames[c->edst[edge]].n;
  if (n < 0 || n > ns) { set_error("n=%d exceeds the edge capacity N_src=%d", n, ns); return MVICP_ERR_ARG; }
  for (int i = 0; i < n; ++i)
    if (first[i] < 0 || first[i] >= ns || second[i] < 0 || second[i] >= nd) { set_error("correspondence %d out of range", i); return MVICP_ERR_ARG; }
  const size_t off = (size_t)c->cap_off[edge];
  MV_HIP(hipStreamSynchronize(c->stream));
  if (n) {
    // device lists are kept in sorted positions
    const std::vector<int>& si = c->frames[c->esrc[edge]].grid.h_inv;
    const std::vector<int>& di = c->frames[c->edst[edge]].grid.h_inv;
    std::vector<int> a(n), b(n);
    for (int i = 0; i < n; ++i) { a[i] = si[first[i]]; b[i] = di[second[i]]; }
    MV_HIP(hipMemcpy(c->d_first + off, a.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    MV_HIP(hipMemcpy(c->d_second + off, b.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    MV_HIP(hipMemset(c->d_cd2 + off, 0, sizeof(double) * n));
  }
  c->nn_cache_valid = false;
  c->spec_ready = false; c->spec2_ready = false;
  c->list_valid[edge] = 0; c->explicit_list[edge] = 1; c->export_valid = false;
  c->qpos_valid[edge] = 0; c->corr_epoch[edge] = ++c->epoch_counter;
  c->sel_med1[edge] = c->sel_med2[edge] = -1.0;
  const double a = (double)weight;
  MV_HIP(hipMemcpy(c->d_count + edge, &n, sizeof(int), hipMemcpyHostToDevice));
  {   // an explicit list is arbitrary (repeats, any order): never the identity list the linearize kernel may shortcut
    const int not_identity = -1;
    MV_HIP(hipMemcpy(c->d_nsrc + edge, &not_identity, sizeof(int), hipMemcpyHostToDevice));
  }
  MV_HIP(hipMemcpy(c->d_a + edge, &a, sizeof(double), hipMemcpyHostToDevice));
  c->h_pin[c->ctl_r2_off + (size_t)c->E * kEdgeRel + edge] = a;   // host mirror: region 2 is re-uploaded by every evaluation
  c->h_count[edge] = n;
  c->h_weight[edge] = weight;
  c->have_corr = true;
  const int one = 1;
  MV_HIP(hipMemcpy(c->d_dirty + edge, &one, sizeof(int), hipMemcpyHostToDevice));  // this edge's operands must be (re)gathered
  MV_CHECK(launch_gather_stream(c));
  MV_HIP(hipStreamSynchronize(c->stream));
  return MVICP_OK;
} MVICP_GUARD_ABI

int mvicp_nn_query(mvicp_ctx* c, int frame, const double* queries, int n, int nn_method, int* idx, double* d2) try {
  MV_CHECK(bind(c));
  if (frame < 0 || frame >= c->n_frames) { set_error("frame %d out of range", frame); return MVICP_ERR_ARG; }
  if (n < 0 || (n && (!queries || !idx || !d2))) { set_error("bad query buffers"); return MVICP_ERR_ARG; }
  MV_CHECK(finish_builds(c));
  const FrameDev& f = c->frames[frame];
  if (f.n == 0) { set_error("frame %d is empty (nanoflann throws here: nanoflann.hpp:904)", frame); return MVICP_ERR_STATE; }
  if (n == 0) return MVICP_OK;
  double* dq = nullptr; int* di = nullptr; double* dd = nullptr;
  MV_CHECK(dev_alloc(&dq, 3 * (size_t)n)); MV_CHECK(dev_alloc(&di, (size_t)n)); MV_CHECK(dev_alloc(&dd, (size_t)n));
  MV_HIP(hipMemcpy(dq, queries, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice));
  int method = (nn_method == MVICP_NN_AUTO || nn_method == MVICP_NN_TILE) ? MVICP_NN_GRID : nn_method;  // raw queries are not patch-ordered
  if (c->tie_rule) MV_CHECK(ensure_tie_trees(c, std::vector<int>(1, frame)));
  c->tie_skip = false;
  if (method == MVICP_NN_GRID && !f.has_grid) method = MVICP_NN_BRUTE;
  int st;
  if (method == MVICP_NN_BRUTE) st = launch_nn_brute_queries(c, f, dq, n, di, dd);
  else if (method == MVICP_NN_GRID) st = launch_nn_grid_queries(c, f, dq, n, di, dd);
  else { set_error("unknown nn_method %d", nn_method); st = MVICP_ERR_ARG; }
  if (st == MVICP_OK) {
    hipError_t e1 = hipStreamSynchronize(c->stream);
    if (e1 == hipSuccess) census_resolve(c);
    hipError_t e2 = hipMemcpy(idx, di, sizeof(int) * n, hipMemcpyDeviceToHost);
    hipError_t e3 = hipMemcpy(d2, dd, sizeof(double) * n, hipMemcpyDeviceToHost);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) { set_error("nn_query copy-back failed"); st = MVICP_ERR_HIP; }
  }
  dev_free(dq); dev_free(di); dev_free(dd);
  if (c->profile) prof_collect_lazy(c);
  return st;
} MVICP_GUARD_ABI

int mvicp_linearize(mvicp_ctx* c, const double* poses, int point_to_plane, int robust, double* out) try {
  MV_CHECK(bind(c));
  if (!poses || !out) { set_error("null argument"); return MVICP_ERR_ARG; }
  if (c->E == 0) { set_error("no graph"); return MVICP_ERR_STATE; }
  MV_CHECK(evaluate_blocks(c, poses, point_to_plane, robust, out));
  if (c->profile) prof_collect_lazy(c);
  return MVICP_OK;
} MVICP_GUARD_ABI

int mvicp_set_option(mvicp_ctx* c, const char* name, double value) try {
  MV_CHECK(bind(c));
  if (!name) { set_error("null option name"); return MVICP_ERR_ARG; }
  if (std::strcmp(name, "async_build") == 0) { c->async_build = value != 0.0; return MVICP_OK; }
  if (std::strcmp(name, "grid_target") == 0 || std::strcmp(name, "grid_curve") == 0) MV_CHECK(finish_builds(c));   // (pending builds read them)
  if (std::strcmp(name, "nn_tree_only") == 0) { c->nn_tree_only = value != 0.0; return MVICP_OK; }
  if (std::strcmp(name, "nn_cache") == 0) { c->nn_cache_enable = value != 0.0; c->nn_cache_valid = false; return MVICP_OK; }
  if (std::strcmp(name, "lin_chunk") == 0) { c->lin_chunk_override = (int)value; return MVICP_OK; }  // takes effect at the next mvicp_set_graph
  if (std::strcmp(name, "list_reuse") == 0) { c->list_reuse = value != 0.0; return MVICP_OK; }
  if (std::strcmp(name, "lin_interleave") == 0) { c->lin_interleave = value != 0.0; return MVICP_OK; }   // takes effect at the next mvicp_set_graph
  if (std::strcmp(name, "lin_share_p") == 0) { c->lin_share_p = value != 0.0; return MVICP_OK; }
  if (std::strcmp(name, "nn_census") == 0) { c->nn_census = value != 0.0; return MVICP_OK; }
  if (std::strcmp(name, "nn_skip_far") == 0) { c->nn_skip_far = value != 0.0; return MVICP_OK; }
  if (std::strcmp(name, "sel_bracket") == 0) { c->sel_bracket = value != 0.0; return MVICP_OK; }
  if (std::strcmp(name, "prune_rho") == 0) { c->prune_rho = value; return MVICP_OK; }
  if (std::strcmp(name, "grid_curve") == 0) { c->grid_curve = (int)value; return MVICP_OK; }   // takes effect at the next mvicp_set_frame
  if (std::strcmp(name, "auto_settle") == 0) { c->auto_settle = value; return MVICP_OK; }
  if (std::strcmp(name, "auto_switch") == 0) { c->auto_switch = value; return MVICP_OK; }
  if (std::strcmp(name, "nn_cell") == 0) { c->nn_cell = value != 0.0; return MVICP_OK; }
  if (std::strcmp(name, "nn_search_factor") == 0) { if (!(value >= 0.0)) { set_error("nn_search_factor < 0"); return MVICP_ERR_ARG; } c->nn_search_factor = value; c->nn_cache_valid = false; return MVICP_OK; }
  if (std::strcmp(name, "tie_rule") == 0) { c->tie_rule = value != 0.0; return MVICP_OK; }
  if (std::strcmp(name, "tie_lazy") == 0) { c->tie_lazy = value != 0.0; return MVICP_OK; }   // takes effect at the next mvicp_set_graph
  if (std::strcmp(name, "tile_seed") == 0) { c->tile_seed = value != 0.0; return MVICP_OK; }
  if (std::strcmp(name, "tile_bounds") == 0) { c->tile_bounds = (int)value; return MVICP_OK; }
  if (std::strcmp(name, "tile_mu") == 0) { if (!(value > 0.0 && value <= 1.0)) { set_error("tile_mu outside (0, 1]"); return MVICP_ERR_ARG; } c->tile_mu = value; return MVICP_OK; }
  if (std::strcmp(name, "mfma_kacc") == 0) { if (!(value >= 1.0 && value <= 1024.0)) { set_error("mfma_kacc outside [1, 1024]"); return MVICP_ERR_ARG; } c->mfma_kacc = value; return MVICP_OK; }
  if (std::strcmp(name, "mfma_lbt") == 0) { c->mfma_lbt = (int)value; return MVICP_OK; }
  if (std::strcmp(name, "mfma_entry") == 0) { c->mfma_entry = (int)value; return MVICP_OK; }
  if (std::strcmp(name, "mfma_trig") == 0) { c->mfma_trig = (int)value; return MVICP_OK; }
  if (std::strcmp(name, "tile_mfma") == 0) { c->tile_mfma = (int)value; return MVICP_OK; }
  if (std::strcmp(name, "tile_waves") == 0) { c->tile_waves = (int)value; return MVICP_OK; }
  if (std::strcmp(name, "cache_mfma_ratio") == 0) { c->cache_mfma_ratio = value; return MVICP_OK; }
  if (std::strcmp(name, "reject_cache") == 0) { c->reject_cache = value != 0.0; return MVICP_OK; }
  if (std::strcmp(name, "tile_miss") == 0) { if (!(value >= 0.0 && value <= 64.0)) { set_error("tile_miss outside [0, 64]"); return MVICP_ERR_ARG; } c->tile_miss = (int)value; return MVICP_OK; }
  if (std::strcmp(name, "tile_cache") == 0) { c->tile_cache = (int)value; return MVICP_OK; }
  if (std::strcmp(name, "spin_wait") == 0) { c->spin_wait = value != 0.0; return MVICP_OK; }
  if (std::strcmp(name, "fault_inject") == 0) { c->fault_inject = (int)value; return MVICP_OK; }            // tests: the value-th mvicp_correspond from now fails locally before its exchange
  if (std::strcmp(name, "fault_inject_build") == 0) { c->fault_inject_build.store((int)value); return MVICP_OK; }      // tests: the value-th structure build from now fails
  if (std::strcmp(name, "fault_inject_eval") == 0) { c->fault_inject_eval = (int)value; return MVICP_OK; }  // tests: the value-th exchanged LM evaluation from now fails locally
  if (std::strcmp(name, "spec_eval") == 0) { c->spec_enable = value != 0.0; c->spec_ready = false; c->spec2_ready = false; return MVICP_OK; }
  if (std::strcmp(name, "spec2_eval") == 0) { c->spec2_enable = value != 0.0; c->spec2_ready = false; return MVICP_OK; }
  if (std::strcmp(name, "grid_target") == 0) {
    if (!(value >= 0.5 && value <= 64.0)) { set_error("grid_target out of range"); return MVICP_ERR_ARG; }
    c->grid_target = value;
    return MVICP_OK;
  }
  set_error("unknown option '%s'", name);
  return MVICP_ERR_ARG;
} MVICP_GUARD_ABI
int mvicp_nn_census_ex(mvicp_ctx* c, double* out, int cap) try {
  MV_CHECK(bind(c));
  if (!out || cap < 0) { set_error("bad output buffer"); return MVICP_ERR_ARG; }
  const double v[10] = {c->nn_queries, c->nn_candidates, c->nn_nodes, c->nn_far, c->nn_hits, c->nn_fetched, c->nn_dbg[0], c->nn_dbg[1], c->nn_dbg[2], c->nn_dbg[3]};
  const int n = std::min(cap, 10);
  for (int i = 0; i < n; ++i) out[i] = v[i];
  return n;
} MVICP_GUARD_ABI
int mvicp_nn_census(mvicp_ctx* c, double* out5) try {   // the original 5-counter contract (a caller's 5-element buffer is never overrun)
  const int n = mvicp_nn_census_ex(c, out5, 5);
  return n < 0 ? n : MVICP_OK;
} MVICP_GUARD_ABI

int mvicp_profile_enable(mvicp_ctx* c, int on) try {
  MV_CHECK(bind(c));
  c->profile = on != 0;
  c->profile_level = on;
  return MVICP_OK;
} MVICP_GUARD_ABI
int mvicp_profile_reset(mvicp_ctx* c) try {
  MV_CHECK(bind(c));
  MV_HIP(hipStreamSynchronize(c->stream));
  prof_collect(c);
  for (auto& kv : c->prof) { kv.second.ms = 0; kv.second.launches = 0; kv.second.bytes = 0; kv.second.survey_bytes = 0; kv.second.queries = 0; }
  c->nn_candidates = c->nn_nodes = c->nn_far = c->nn_queries = c->nn_hits = c->nn_fetched = 0;
  for (int k = 0; k < 4; ++k) c->nn_dbg[k] = 0;
  return MVICP_OK;
} MVICP_GUARD_ABI
// "nn" = every NN kernel scope together (nn_brute, nn_grid, nn_tile, nn_mfma)
static void prof_sum(mvicp_ctx* c, const char* kernel, double out[5]) {
  for (int k = 0; k < 5; ++k) out[k] = 0.0;
  const std::string key = kernel ? kernel : "";
  for (const auto& kv : c->prof) {
    if (!(kv.first == key || (key == "nn" && kv.first.compare(0, 3, "nn_") == 0))) continue;
    out[0] += kv.second.ms;
    // "nn": the auxiliary scopes (nn_far: phase 2 + list flags, nn_tie: the tie fix-up) add their time, but a search is ONE launch of a primary kernel
    if (key == "nn" && (kv.first == "nn_far" || kv.first == "nn_tie")) continue;
    out[1] += (double)kv.second.launches; out[2] += kv.second.bytes; out[3] += kv.second.survey_bytes; out[4] += kv.second.queries;
  }
}
int mvicp_profile_get(mvicp_ctx* c, const char* kernel, double* total_ms, long long* launches, double* alg_bytes) try {
  MV_CHECK(bind(c));
  MV_HIP(hipStreamSynchronize(c->stream));
  prof_collect(c);
  double v[5];
  prof_sum(c, kernel, v);
  if (total_ms) *total_ms = v[0];
  if (launches) *launches = (long long)v[1];
  if (alg_bytes) *alg_bytes = v[2];
  return MVICP_OK;
} MVICP_GUARD_ABI
int mvicp_profile_get_ex(mvicp_ctx* c, const char* kernel, double* out, int cap) try {
  MV_CHECK(bind(c));
  if (!out || cap < 0) { set_error("bad output buffer"); return MVICP_ERR_ARG; }
  MV_HIP(hipStreamSynchronize(c->stream));
  prof_collect(c);
  double v[5];
  prof_sum(c, kernel, v);
  const int n = std::min(cap, 5);
  for (int k = 0; k < n; ++k) out[k] = v[k];
  return n;
} MVICP_GUARD_ABI
void* mvicp_stream(mvicp_ctx* c) { return c ? (void*)c->stream : nullptr; }
int mvicp_sync(mvicp_ctx* c) try {
  MV_CHECK(bind(c));
  MV_HIP(hipStreamSynchronize(c->stream));
  return MVICP_OK;
} MVICP_GUARD_ABI

}  // extern "C"
