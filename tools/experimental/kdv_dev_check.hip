// Debug aid: evaluates mvicp::visited_before on the DEVICE for (query, a, b) triples and compares with the host.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -o kdv_dev_check kdv_dev_check.hip ; run: ./kdv_dev_check cloud.bin triples.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../mv-lm-icp_amd/csrc/kdvisit.h"
using namespace mvicp;
__global__ void k(VisitTree T, const double* xyz, const int* tri, int m, int* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int q = tri[3 * i], a = tri[3 * i + 1], b = tri[3 * i + 2];
  out[i] = visited_before(T, xyz[3 * q], xyz[3 * q + 1], xyz[3 * q + 2], (long long)a, (long long)b) ? 1 : 0;
}
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb"); int n; fread(&n, 4, 1, f); std::vector<double> xyz(3 * (size_t)n); fread(xyz.data(), 8, xyz.size(), f); fclose(f);
  f = fopen(argv[2], "rb"); int m; fread(&m, 4, 1, f); std::vector<int> tri(3 * (size_t)m); fread(tri.data(), 4, tri.size(), f); fclose(f);
  std::vector<VisitNode> nodes; std::vector<int> slot;
  build_visit_tree(xyz.data(), n, nodes, slot);
  VisitTree H{nodes.data(), slot.data()};
  VisitNode* dn; int* ds; double* dx; int* dt; int* dout;
  hipMalloc(&dn, sizeof(VisitNode) * nodes.size()); hipMalloc(&ds, 4 * n); hipMalloc(&dx, 8 * xyz.size()); hipMalloc(&dt, 4 * tri.size()); hipMalloc(&dout, 4 * m);
  hipMemcpy(dn, nodes.data(), sizeof(VisitNode) * nodes.size(), hipMemcpyHostToDevice); hipMemcpy(ds, slot.data(), 4 * n, hipMemcpyHostToDevice);
  hipMemcpy(dx, xyz.data(), 8 * xyz.size(), hipMemcpyHostToDevice); hipMemcpy(dt, tri.data(), 4 * tri.size(), hipMemcpyHostToDevice);
  VisitTree D{dn, ds};
  hipLaunchKernelGGL(k, dim3((m + 255) / 256), dim3(256), 0, 0, D, dx, dt, m, dout);
  std::vector<int> out(m); hipMemcpy(out.data(), dout, 4 * m, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < m; ++i) {
    const int q = tri[3 * i], a = tri[3 * i + 1], b = tri[3 * i + 2];
    const int h = visited_before(H, xyz[3 * q], xyz[3 * q + 1], xyz[3 * q + 2], a, b) ? 1 : 0;
    if (h != out[i]) { if (bad < 5) printf("MISMATCH q=%d a=%d b=%d host %d dev %d\n", q, a, b, h, out[i]); ++bad; }
  }
  printf("sizeof(VisitNode)=%zu triples %d mismatches %d err %s\n", sizeof(VisitNode), m, bad, hipGetErrorString(hipGetLastError()));
  return 0;
}
