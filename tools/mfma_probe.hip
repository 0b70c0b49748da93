// Probe for the matrix-pipe screen of nn_mfma.hip (GPU box): checks, on the device,
//   1. the operand / result lane maps of v_mfma_f32_32x32x16_f16 the kernel relies on
//      (A: lane l = row l & 31, k = 8 (l >> 5) + e; B: lane l = column l & 31, same k; D: column l & 31, row (r & 3) + 8 (r >> 2) + 4 (l >> 5)),
//   2. the half exchange of v_permlane32_swap_b32,
//   3. the accumulation error of one instruction against the exact sum of its (exact) products, as a multiple of
//      2^-24 * sum |term|: the guard band of the screen is derived from that ratio.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/_build/mfma_probe tools/mfma_probe.hip     Run: tools/_build/mfma_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void probe_kernel(const _Float16* __restrict__ A, const _Float16* __restrict__ B, float* __restrict__ D, unsigned int* __restrict__ swp) {
  const int l = threadIdx.x;
  h8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = A[(l & 31) * 16 + 8 * (l >> 5) + e];   // A[m][k], row-major 32 x 16
    b[e] = B[(8 * (l >> 5) + e) * 32 + (l & 31)]; // B[k][n], row-major 16 x 32
  }
  f16v c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    D[row * 32 + (l & 31)] = c[r];
  }
  unsigned int v0 = 1000u + l, v1 = 2000u + l;
  auto rr = __builtin_amdgcn_permlane32_swap(v0, v1, false, false);
  swp[l] = rr[0]; swp[64 + l] = rr[1];
}

int main() {
  std::mt19937_64 rng(7);
  std::vector<_Float16> A(32 * 16), B(16 * 32);
  std::vector<float> D(32 * 32);
  _Float16 *dA, *dB; float* dD; unsigned int* dS;
  hipMalloc(&dA, sizeof(_Float16) * A.size()); hipMalloc(&dB, sizeof(_Float16) * B.size()); hipMalloc(&dD, sizeof(float) * D.size()); hipMalloc(&dS, 128 * 4);
  double worst_ratio = 0.0, worst_abs = 0.0;
  int layout_bad = 0;
  for (int trial = 0; trial < 400; ++trial) {
    // screen-like operands: coordinates up to 2^7..2^9 with hi / lo parts, norms up to 2^15, mixed signs
    const double mag = std::ldexp(1.0, 5 + (int)(rng() % 6));
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    for (auto& x : A) x = (_Float16)(float)(U(rng) * mag * ((rng() & 3) == 0 ? 1.0 / 1024 : 1.0));
    for (auto& x : B) x = (_Float16)(float)(U(rng) * mag * ((rng() & 3) == 0 ? 1.0 / 1024 : 1.0));
    if (trial == 0) {   // asymmetric integers: any row / column / k permutation shows
      for (int m = 0; m < 32; ++m) for (int k = 0; k < 16; ++k) A[m * 16 + k] = (_Float16)(float)((m * 3 + k * 5) % 17 - 8);
      for (int k = 0; k < 16; ++k) for (int n = 0; n < 32; ++n) B[k * 32 + n] = (_Float16)(float)((k * 7 + n * 11) % 13 - 6);
    }
    hipMemcpy(dA, A.data(), sizeof(_Float16) * A.size(), hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), sizeof(_Float16) * B.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD, dS);
    hipMemcpy(D.data(), dD, sizeof(float) * D.size(), hipMemcpyDeviceToHost);
    for (int m = 0; m < 32; ++m)
      for (int n = 0; n < 32; ++n) {
        double s = 0.0, sa = 0.0;
        for (int k = 0; k < 16; ++k) { const double t = (double)(float)A[m * 16 + k] * (double)(float)B[k * 32 + n]; s += t; sa += std::fabs(t); }
        const double err = std::fabs((double)D[m * 32 + n] - s);
        if (trial == 0 && err != 0.0) ++layout_bad;
        if (sa > 0) { worst_ratio = std::max(worst_ratio, err / (sa * std::ldexp(1.0, -24))); worst_abs = std::max(worst_abs, err); }
      }
  }
  std::vector<unsigned int> S(128);
  hipMemcpy(S.data(), dS, 128 * 4, hipMemcpyDeviceToHost);
  int swap_bad = 0;
  for (int l = 0; l < 64; ++l) {
    const unsigned int e0 = l < 32 ? 1000u + l : 2000u + (l - 32);   // vdst: upper half <- src's lower half
    const unsigned int e1 = l < 32 ? 1000u + (l + 32) : 2000u + l;   // src: lower half <- vdst's upper half
    if (S[l] != e0 || S[64 + l] != e1) ++swap_bad;
  }
  printf("mfma_f32_32x32x16_f16 layout mismatches (integer case): %d\n", layout_bad);
  printf("permlane32_swap mismatches: %d\n", swap_bad);
  printf("accumulation error: worst |err| / (2^-24 * sum|term|) = %.3f   (worst abs %.3g)\n", worst_ratio, worst_abs);
  return (layout_bad || swap_bad) ? 1 : 0;
}
