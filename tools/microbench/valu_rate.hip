// VALU issue rate on MI355X: wave-instructions per second for a few instruction kinds, at full occupancy with
// 8 independent chains per lane (no dependency stalls), plus the shader clock seen by clock64() vs wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, long long* clk) {
  float v[8];
  uint32_t w[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { v[k] = threadIdx.x * 0.001f + k; w[k] = threadIdx.x * 2654435761u + k; }
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (KIND == 0) v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);                       // v_fma_f32
      if (KIND == 1) w[k] = w[k] ^ (w[k] >> 3);                                         // shift + xor (2 ops)
      if (KIND == 2) { const uint64_t p = (uint64_t)w[k] * 0xD2511F53u; w[k] = (uint32_t)(p >> 32) ^ (uint32_t)p; }  // v_mad_u64_u32 + xor
      if (KIND == 3) v[k] = __builtin_amdgcn_sqrtf(v[k]) + 1.5f;                        // v_sqrt_f32 + add
      if (KIND == 4) v[k] = v[k] > 2.0f ? v[k] - 1.0f : v[k] + 0.75f;                   // cmp + sub + add + cndmask
      if (KIND == 7) { uint64_t t = ((uint64_t)w[k] << 32) | w[(k + 1) & 7]; t += 0x123456789ull * (k + 1); w[k] = (uint32_t)(t >> 32) ^ (uint32_t)t; }  // 64-bit add + xor
    }
  }
  const long long t1 = clock64();
  float s = 0; uint32_t x = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) { s += v[k]; x ^= w[k]; }
  if (s == -1.0f || x == 0x12345u) out[blockIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}
template <int KIND>
void run(const char* name, int ops_per_iter, float* out, long long* clk) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 4000, grid = 256 * 8 * 4;  // 8 workgroups of 4 waves per CU, 4 rounds
  hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(256), 0, 0, out, iters, clk);
  hipEventRecord(a);
  hipLaunchKernelGGL(k_rate<KIND>, dim3(grid), dim3(256), 0, 0, out, iters, clk);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
  const double waves = (double)grid * 4, winstr = waves * iters * 8.0 * ops_per_iter;
  printf("%-28s %.3f ms  %.3e wave-instr/s  (= %.2f cycles per wave-instr per SIMD at 2.4 GHz)  clock64 span of one wave %lld\n", name, ms,
         winstr / (ms * 1e-3), 1024.0 * 2.4e9 / (winstr / (ms * 1e-3)), c);
}
typedef float float2v __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_pk(float* out, int iters) {
  float2v v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = float2v{threadIdx.x * 0.001f + k, threadIdx.x * 0.002f + k};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = v[k] * float2v{1.0001f, 0.9999f};  // v_pk_mul_f32
  }
  float s = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += v[k].x + v[k].y;
  if (s == -1.0f) out[blockIdx.x] = s;
}
void run_pk(float* out, long long*) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 4000, grid = 256 * 8 * 4;
  hipLaunchKernelGGL(k_pk, dim3(grid), dim3(256), 0, 0, out, iters);
  hipEventRecord(a);
  hipLaunchKernelGGL(k_pk, dim3(grid), dim3(256), 0, 0, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double winstr = (double)grid * 4 * iters * 8.0;
  printf("%-28s %.3f ms  %.3e wave-instr/s  (= %.2f cycles per wave-instr per SIMD at 2.4 GHz)\n", "pk_mul_f32", ms, winstr / (ms * 1e-3),
         1024.0 * 2.4e9 / (winstr / (ms * 1e-3)));
}
int main() {
  float* out; long long* clk; hipMalloc(&out, 1 << 20); hipMalloc(&clk, 8);
  run<0>("fma_f32", 1, out, clk);
  run<1>("lshr+xor", 2, out, clk);
  run<2>("mad_u64_u32+xor", 2, out, clk);
  run<3>("sqrt_f32+add", 2, out, clk);
  run<4>("cmp+sub+add+cndmask", 4, out, clk);
  run<7>("add_u64+xor", 2, out, clk);
  run_pk(out, clk);
  return 0;
}
