// How long does a wave wait for its kernel arguments?  Three kernels at the SMC step's launch shape (977 x 256), each launched
// 200 times: (a) never touches its arguments, (b) branches on one of them, (c) = (b) in a translation unit compiled with
// -mllvm -amdgpu-kernarg-preload-count=8 (kernarg_lat_preload.hip includes this file with KPRE defined).  Wave lifetime =
// SQ_WAVE_CYCLES / SQ_WAVES under rocprofv3 --pmc; wall time per launch by HIP events.
//   hipcc --offload-arch=gfx950 -O3 -o kernarg_lat kernarg_lat.hip kernarg_lat_preload.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#ifndef KPRE
__global__ __launch_bounds__(256) void k_noarg(uint32_t* out, int x) {}
__global__ __launch_bounds__(256) void k_arg(uint32_t* out, int x) {
  if (x == 12345) out[threadIdx.x] = 1;
}
void launch_pre(uint32_t* d, int x, hipStream_t s);
template <class F>
static float per_launch_us(F f) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 200; ++i) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms * 1000.0f / 200.0f;
}
int main() {
  uint32_t* d;
  hipMalloc(&d, 4096);
  printf("no argument read : %.2f us per launch\n", per_launch_us([&] { k_noarg<<<977, 256>>>(d, 1); }));
  printf("argument read    : %.2f us per launch\n", per_launch_us([&] { k_arg<<<977, 256>>>(d, 1); }));
  printf("argument preload : %.2f us per launch\n", per_launch_us([&] { launch_pre(d, 1, 0); }));
  return 0;
}
#else
__global__ __launch_bounds__(256) void k_arg_pre(uint32_t* out, int x) {
  if (x == 12345) out[threadIdx.x] = 1;
}
void launch_pre(uint32_t* d, int x, hipStream_t s) { k_arg_pre<<<977, 256, 0, s>>>(d, x); }
#endif
