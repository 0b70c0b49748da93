// Known-byte kernels in the ACCESS PATTERNS of this library's kernels, to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// (MI355X_MICROARCH.md, "HBM": FETCH_SIZE reports half the bytes of a 16-B/lane coalesced stream; "other access widths and WRITE_SIZE
// are uncalibrated: calibrate on a known byte count in your own access pattern").  VERDICT r4 item 6c.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/pmc_calib tools/pmc_calib.hip
//   bash tools/pmc_kernel.sh calib_fetch FETCH_SIZE tools/_build/pmc_calib ; bash tools/pmc_kernel.sh calib_write WRITE_SIZE tools/_build/pmc_calib
//   python tools/pmc_calib_summary.py        -> profiles/r05_pmc_calibration.txt
//
// Every kernel touches each byte of a 1-GiB region exactly once (4x the 256-MiB Infinity Cache), so bytes moved = bytes counted below:
//   read_stream4 / 8 / 16   coalesced streams, 4 / 8 / 16 B per lane  (box SoA arrays; the SoA operand stream of linearize_kernel and d2 keys;
//                            nn_mfma's 16-B A fragments within a tile and the query / record streams)
//   read_tile1k             a wave reads 64 x 16 B = one contiguous 1-KiB tile fragment, tiles in pseudo-random order  (nn_mfma_kernel: A fragments)
//   read_rec32              every lane reads one 32-B record (two 16-B loads) at a pseudo-random position  (fp64 confirmations, seeds, gather_kernel)
//   write_stream4 / 8 / 16  coalesced stores  (nn_idx 4 B, nn_d2 8 B per query; the operand stream; export triples 16 B)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr size_t kBytes = 1ull << 30;

template <typename T> __global__ __launch_bounds__(256) void read_stream(const T* __restrict__ p, size_t n, unsigned long long* sink) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const T v = p[i];
    const unsigned int* w = reinterpret_cast<const unsigned int*>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc += w[k];
  }
  if (acc == 0x123456789abcull) *sink = acc;
}
__global__ __launch_bounds__(128) void read_tile1k(const uint4* __restrict__ p, size_t tiles, unsigned long long* sink) {
  unsigned long long acc = 0;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63, waves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t t = wave; t < tiles; t += waves) {
    const size_t tile = (t * 2654435761ull) % tiles;   // (tiles is a power of two times an odd factor-free count: a permutation, see main)
    const uint4 v = p[tile * 64 + lane];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 0x123456789abcull) *sink = acc;
}
__global__ __launch_bounds__(256) void read_rec32(const uint4* __restrict__ p, size_t recs, unsigned long long* sink) {
  unsigned long long acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < recs; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = (i * 2654435761ull) % recs;
    const uint4 a = p[2 * r], b = p[2 * r + 1];
    acc += a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
  }
  if (acc == 0x123456789abcull) *sink = acc;
}
template <typename T> __global__ __launch_bounds__(256) void write_stream(T* __restrict__ p, size_t n, T v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

int main() {
  void* buf = nullptr; unsigned long long* sink = nullptr;
  CK(hipMalloc(&buf, kBytes)); CK(hipMalloc((void**)&sink, 8));
  CK(hipMemset(buf, 1, kBytes));
  const dim3 grid(256 * 16), blk(256);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(read_stream<float>, grid, blk, 0, 0, (const float*)buf, kBytes / 4, sink);
    hipLaunchKernelGGL(read_stream<double>, grid, blk, 0, 0, (const double*)buf, kBytes / 8, sink);
    hipLaunchKernelGGL(read_stream<uint4>, grid, blk, 0, 0, (const uint4*)buf, kBytes / 16, sink);
    hipLaunchKernelGGL(read_tile1k, dim3(256 * 32), dim3(128), 0, 0, (const uint4*)buf, kBytes / 1024, sink);   // 2^20 tiles: odd multiplier -> a permutation
    hipLaunchKernelGGL(read_rec32, grid, blk, 0, 0, (const uint4*)buf, kBytes / 32, sink);                      // 2^25 records: the same
    hipLaunchKernelGGL(write_stream<float>, grid, blk, 0, 0, (float*)buf, kBytes / 4, 1.0f);
    hipLaunchKernelGGL(write_stream<double>, grid, blk, 0, 0, (double*)buf, kBytes / 8, 1.0);
    hipLaunchKernelGGL(write_stream<uint4>, grid, blk, 0, 0, (uint4*)buf, kBytes / 16, make_uint4(1, 2, 3, 4));
    CK(hipDeviceSynchronize());
  }
  std::printf("every kernel moves %zu bytes per launch (3 launches each)\n", kBytes);
  CK(hipFree(buf)); CK(hipFree(sink));
  return 0;
}
