// K2 — spatial-hash (uniform grid) exact NN.  (placeholder TU: structure build + kernels land next)
#include "common.h"

namespace mvicp {

int build_grid(mvicp_ctx* c, FrameDev& f, const double* h_xyz) {
  (void)c; (void)h_xyz;
  f.has_grid = false;
  return MVICP_OK;
}
void free_grid(GridDev& g) { (void)g; }
int launch_nn_grid_edges(mvicp_ctx* c, double d2_bound) {
  (void)c; (void)d2_bound;
  set_error("grid NN not built");
  return MVICP_ERR_STATE;
}
int launch_nn_grid_queries(mvicp_ctx* c, const FrameDev& f, const double* d_q, int n, int* d_idx, double* d_d2) {
  (void)c; (void)f; (void)d_q; (void)n; (void)d_idx; (void)d_d2;
  set_error("grid NN not built");
  return MVICP_ERR_STATE;
}

}  // namespace mvicp
