// Store-path microbenchmark: n particles x 12 f32 columns (48 B per particle) written as the importance
// kernels write them.  A: one particle per lane, 12 dword stores.  B: two adjacent particles per lane, 12
// dwordx2 stores.  C: four adjacent particles per lane, 12 dwordx4 stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
struct Cols { float* c[12]; };
__global__ __launch_bounds__(256) void kA(Cols cols, uint64_t n) {
  for (uint64_t row = blockIdx.x; row * 256 < n; row += gridDim.x) {
    const uint64_t i = row * 256 + threadIdx.x;
    if (i < n) {
      float v = (float)i;
#pragma unroll
      for (int k = 0; k < 12; ++k) { v = v * 1.0001f + 0.5f; cols.c[k][i] = v; }
    }
  }
}
__global__ __launch_bounds__(128) void kB(Cols cols, uint64_t n) {
  for (uint64_t row = blockIdx.x; row * 256 < n; row += gridDim.x) {
    const uint64_t i = row * 256 + 2 * (uint64_t)threadIdx.x;
    if (i + 1 < n) {
      float v = (float)i;
#pragma unroll
      for (int k = 0; k < 12; ++k) { v = v * 1.0001f + 0.5f; reinterpret_cast<float2*>(cols.c[k] + i)[0] = make_float2(v, v + 1.0f); }
    }
  }
}
__global__ __launch_bounds__(128) void kB1(Cols cols, uint64_t n) {  // two scalar stores per column (as emitted today)
  for (uint64_t row = blockIdx.x; row * 256 < n; row += gridDim.x) {
    const uint64_t i = row * 256 + 2 * (uint64_t)threadIdx.x;
    if (i + 1 < n) {
      float v = (float)i;
#pragma unroll
      for (int k = 0; k < 12; ++k) { v = v * 1.0001f + 0.5f; cols.c[k][i] = v; cols.c[k][i + 1] = v + 1.0f; }
    }
  }
}
__global__ __launch_bounds__(64) void kC(Cols cols, uint64_t n) {
  for (uint64_t row = blockIdx.x; row * 256 < n; row += gridDim.x) {
    const uint64_t i = row * 256 + 4 * (uint64_t)threadIdx.x;
    if (i + 3 < n) {
      float v = (float)i;
#pragma unroll
      for (int k = 0; k < 12; ++k) { v = v * 1.0001f + 0.5f; reinterpret_cast<float4*>(cols.c[k] + i)[0] = make_float4(v, v + 1.0f, v + 2.0f, v + 3.0f); }
    }
  }
}
template <class F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  std::vector<float> ts;
  for (int rep = 0; rep < 20; ++rep) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); ts.push_back(ms * 1e3f); }
  std::sort(ts.begin(), ts.end()); return ts[ts.size() / 2];
}
int main() {
  const uint64_t n = 1000000; Cols cols;
  for (int k = 0; k < 12; ++k) (void)hipMalloc(&cols.c[k], (n + 1024) * 4);
  const unsigned rows = (unsigned)((n + 255) / 256);
  printf("A  1/lane dword    : %.1f us\n", timeit([&] { hipLaunchKernelGGL(kA, dim3(rows), dim3(256), 0, 0, cols, n); }));
  printf("B1 2/lane 2xdword  : %.1f us\n", timeit([&] { hipLaunchKernelGGL(kB1, dim3(rows), dim3(128), 0, 0, cols, n); }));
  printf("B  2/lane dwordx2  : %.1f us\n", timeit([&] { hipLaunchKernelGGL(kB, dim3(rows), dim3(128), 0, 0, cols, n); }));
  printf("C  4/lane dwordx4  : %.1f us\n", timeit([&] { hipLaunchKernelGGL(kC, dim3(rows), dim3(64), 0, 0, cols, n); }));
  for (unsigned g : {512u, 1024u, 2048u}) {
    printf("A grid %u: %.1f us;  C grid %u: %.1f us\n", g, timeit([&] { hipLaunchKernelGGL(kA, dim3(g), dim3(256), 0, 0, cols, n); }), g,
           timeit([&] { hipLaunchKernelGGL(kC, dim3(g), dim3(64), 0, 0, cols, n); }));
  }
  printf("(event floor ~6 us)\n");
  return 0;
}
