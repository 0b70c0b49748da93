// RCCL over xGMI: the only collective on the path is an in-place fp64 sum all-reduce of the per-edge
// normal-equation blocks (E x 91 doubles, <= 92 KB at 126 edges) once per LM evaluation; the per-round counts / median d2 /
// "use the queued evaluation" decision ride in the tail of the same buffer (api.cpp).  Every edge slot is written by exactly one
// rank (the others contribute +0.0), so the sum is exact and the result is bit-identical for any
// number of GPUs.  Latency-bound, not link-bound: one collective per evaluation, nothing to bucket.
#include "comm.h"

#include <dlfcn.h>

#include <cstring>

namespace mvicp {

struct Id128 { char b[128]; };  // ncclUniqueId, passed by value

struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Id128, int) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*CommCount)(void*, int*) = nullptr;   // optional
};

namespace {
RcclApi* g_api = nullptr;

RcclApi* load(const char* path) {
  if (g_api) return g_api;
  const char* p = (path && *path) ? path : "librccl.so.1";
  void* h = dlopen(p, RTLD_NOW | RTLD_GLOBAL);
  if (!h && !(path && *path)) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) { set_error("dlopen(%s) failed: %s", p, dlerror()); return nullptr; }
  RcclApi* a = new RcclApi();
  a->handle = h;
  a->GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
  a->CommInitRank = (int (*)(void**, int, Id128, int))dlsym(h, "ncclCommInitRank");
  a->AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclAllReduce");
  a->CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
  a->GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
  a->CommCount = (int (*)(void*, int*))dlsym(h, "ncclCommCount");
  if (!a->GetUniqueId || !a->CommInitRank || !a->AllReduce || !a->CommDestroy) {
    set_error("%s lacks the nccl* entry points", p);
    delete a;
    return nullptr;
  }
  g_api = a;
  return a;
}
const char* errstr(RcclApi* a, int r) { return a->GetErrorString ? a->GetErrorString(r) : "rccl error"; }
}  // namespace

int comm_unique_id(const char* path, void* id128) {
  RcclApi* a = load(path);
  if (!a) return MVICP_ERR_COMM;
  const int r = a->GetUniqueId(id128);
  if (r != 0) { set_error("ncclGetUniqueId: %s", errstr(a, r)); return MVICP_ERR_COMM; }
  return MVICP_OK;
}

int comm_init(mvicp_ctx* c, const char* path, const void* id128, int rank, int world) {
  RcclApi* a = load(path);
  if (!a) return MVICP_ERR_COMM;
  if (rank != c->rank || world != c->world) { set_error("comm rank/world %d/%d differs from the shard %d/%d", rank, world, c->rank, c->world); return MVICP_ERR_ARG; }
  Id128 id;
  std::memcpy(id.b, id128, 128);
  void* comm = nullptr;
  const int r = a->CommInitRank(&comm, world, id, rank);
  if (r != 0) { set_error("ncclCommInitRank: %s", errstr(a, r)); return MVICP_ERR_COMM; }
  c->rccl = a;
  c->comm = comm;
  return MVICP_OK;
}

void comm_destroy(mvicp_ctx* c) {
  if (c->comm && c->rccl) c->rccl->CommDestroy(c->comm);
  c->comm = nullptr;
}

int comm_allreduce_sum(mvicp_ctx* c, double* d_buf, size_t n) {
  if (c->ar_fn) {   // an explicit callback overrides the RCCL communicator
    // host-staged exchange through the launcher's callback (e.g. gloo): device -> host -> all-reduce -> device
    c->ar_host.resize(n);
    MV_HIP(hipMemcpyAsync(c->ar_host.data(), d_buf, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
    MV_HIP(hipStreamSynchronize(c->stream));
    if (c->ar_fn(c->ar_user, c->ar_host.data(), n) != 0) { set_error("all-reduce callback failed"); return MVICP_ERR_COMM; }
    MV_HIP(hipMemcpyAsync(d_buf, c->ar_host.data(), sizeof(double) * n, hipMemcpyHostToDevice, c->stream));
    MV_HIP(hipStreamSynchronize(c->stream));
    return MVICP_OK;
  }
  if (!c->comm) return MVICP_OK;
  const int r = c->rccl->AllReduce(d_buf, d_buf, n, /*ncclFloat64*/ 8, /*ncclSum*/ 0, c->comm, c->stream);
  if (r != 0) { set_error("ncclAllReduce: %s", errstr(c->rccl, r)); return MVICP_ERR_COMM; }
  return MVICP_OK;
}

// ranks of the communicator as RCCL itself reports them (ncclCommCount); 0 without a communicator
int comm_nranks(mvicp_ctx* c) {
  if (!c->comm || !c->rccl) return 0;
  int n = 0;
  if (!c->rccl->CommCount || c->rccl->CommCount(c->comm, &n) != 0) return -1;
  return n;
}

}  // namespace mvicp

using namespace mvicp;
extern "C" {
int mvicp_comm_unique_id(const char* librccl_path, void* unique_id_128) try {
  if (!unique_id_128) { set_error("null id"); return MVICP_ERR_ARG; }
  return comm_unique_id(librccl_path, unique_id_128);
} MVICP_GUARD_ABI
int mvicp_comm_nranks(mvicp_ctx* c) try {
  if (!c) { set_error("null context"); return MVICP_ERR_ARG; }
  return comm_nranks(c);
} MVICP_GUARD_ABI
int mvicp_comm_set_callback(mvicp_ctx* c, mvicp_allreduce_fn fn, void* user) try {
  if (!c) { set_error("null context"); return MVICP_ERR_ARG; }
  c->ar_fn = fn; c->ar_user = user;
  return MVICP_OK;
} MVICP_GUARD_ABI
int mvicp_comm_init(mvicp_ctx* c, const char* librccl_path, const void* unique_id_128, int rank, int world) try {
  if (!c || !unique_id_128) { set_error("null argument"); return MVICP_ERR_ARG; }
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) { set_error("hipSetDevice failed"); return MVICP_ERR_HIP; }
  return comm_init(c, librccl_path, unique_id_128, rank, world);
} MVICP_GUARD_ABI
}
