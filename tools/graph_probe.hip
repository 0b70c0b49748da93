// tools/graph_probe.hip — does a hipGraph lower the fixed cost of a FIXED-POINT ICP round?  (measured on the GPU box, not costed)
//
// A fixed-point round of mvicp_correspond is: one 4-KB H2D copy of the control block from pinned memory, then five dependent kernels on one stream
// (nn_grid verify, bracket pass, bracket final, linearize, reduce), then ONE host wait.  This probe replays exactly that shape with spin kernels of
// the durations the real kernels have on (a) the per-rank share of cfg4 on 8 GPUs (`shard8`: 15 / 5 / 3 / 14 / 3 us) and (b) cfg4 on one GPU
// (120 / 25 / 10 / 107 / 5 us), eagerly (six API calls) and as one hipGraphLaunch of the captured sequence, and reports medians of
//   enqueue  = host time until the last API call returned,
//   total    = host time until hipStreamSynchronize returned (what a round waits for),
//   overhead = total - sum of the kernel durations.
//     hipcc --offload-arch=gfx950 -O2 -o tools/_build/graph_probe tools/graph_probe.hip && tools/_build/graph_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin_kernel(const double* __restrict__ in, double* __restrict__ out, long long ticks) {   // wall_clock64: 100 MHz, constant
  const long long t0 = wall_clock64();
  double v = in[threadIdx.x & 63];
  while (wall_clock64() - t0 < ticks) v = v * 1.0000001 + 1e-9;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = v;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  double *h = nullptr, *d = nullptr, *o = nullptr;
  CK(hipHostMalloc((void**)&h, 4096, hipHostMallocDefault));
  CK(hipMalloc((void**)&d, 4096)); CK(hipMalloc((void**)&o, 4096));
  for (int i = 0; i < 512; ++i) h[i] = 1.0;
  const int sets[2][5] = {{15, 5, 3, 14, 3}, {120, 25, 10, 107, 5}};
  const char* names[2] = {"shard8-like", "cfg4-like"};
  for (int s = 0; s < 2; ++s) {
    double sum = 0; for (int k = 0; k < 5; ++k) sum += sets[s][k];
    auto enqueue = [&]() -> int {
      CK(hipMemcpyAsync(d, h, 4096, hipMemcpyHostToDevice, st));
      for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(64), 0, st, (const double*)d, o, (long long)sets[s][k] * 100);
      return 0;
    };
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    if (enqueue()) return 1;
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int mode = 0; mode < 2; ++mode) {
      std::vector<double> tq, tt;
      for (int it = 0; it < 300; ++it) {
        CK(hipStreamSynchronize(st));
        const double t0 = now_us();
        if (mode == 0) { if (enqueue()) return 1; } else CK(hipGraphLaunch(ge, st));
        const double t1 = now_us();
        CK(hipStreamSynchronize(st));
        const double t2 = now_us();
        if (it >= 50) { tq.push_back(t1 - t0); tt.push_back(t2 - t0); }
      }
      std::printf("%-12s %-6s enqueue %6.1f us   total %7.1f us   overhead over the %3.0f us of kernels %6.1f us\n", names[s], mode == 0 ? "eager" : "graph", median(tq), median(tt), sum,
                  median(tt) - sum);
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
