// Service time of agent-scope ticket atomics on gfx950: G workgroups, one fetch_add each on counter (blockIdx % S).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_ticket(unsigned* t, int S, int stride, int spin, float* sink) {
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  if (threadIdx.x == 0) {
    unsigned v = __hip_atomic_fetch_add(t + (blockIdx.x % S) * stride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v == 0xffffffffu) sink[0] = a;
  }
  if (a == 12345.0f) sink[1] = a;
}
__global__ void k_ticket_noret(unsigned* t, int S, int stride, int spin, float* sink) {
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  if (threadIdx.x == 0) __hip_atomic_fetch_add(t + (blockIdx.x % S) * stride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (a == 12345.0f) sink[1] = a;
}
int main() {
  unsigned* t; float* sink;
  hipMalloc(&t, 4096 * 64 * 4); hipMalloc(&sink, 64);
  hipMemset(t, 0, 4096 * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int ret = 0; ret < 2; ++ret)
  for (int spin : {0, 20000})
  for (int G : {977, 3907})
    for (int S : {1, 16, 64, 256, 1024}) {
      for (int w = 0; w < 3; ++w) (ret ? k_ticket : k_ticket_noret)<<<G, 256>>>(t, S, 64, spin, sink);
      hipEventRecord(e0);
      for (int r = 0; r < 20; ++r) (ret ? k_ticket : k_ticket_noret)<<<G, 256>>>(t, S, 64, spin, sink);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("ret %d spin %5d G %4d S %4d: %.2f us per launch\n", ret, spin, G, S, ms * 1e3 / 20);
    }
  return 0;
}
