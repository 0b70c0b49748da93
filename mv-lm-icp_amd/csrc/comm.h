// RCCL glue (dlopen'ed so the library has no link-time dependency on a particular librccl copy: inside
// a torch process the copy torch already loaded is reused by soname).
#pragma once
#include "common.h"

namespace mvicp {
int comm_unique_id(const char* path, void* id128);
int comm_init(mvicp_ctx* c, const char* path, const void* id128, int rank, int world);
void comm_destroy(mvicp_ctx* c);
// in-place sum over ranks of a device fp64 buffer, on the context's stream
int comm_allreduce_sum(mvicp_ctx* c, double* d_buf, size_t n);
int comm_nranks(mvicp_ctx* c);   // ncclCommCount of the communicator (0: none, -1: not reported)
}  // namespace mvicp
