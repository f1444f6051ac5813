// Cost of agent-scope (sc1, write-through) stores from G one-wave workgroups, by the distance between the words written.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k_store(unsigned* t, unsigned long long* t2, int stride, float* sink) {
  if (threadIdx.x == 0) {
    if (MODE == 0) {
      __hip_atomic_store(t + (size_t)blockIdx.x * stride, blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(t2 + (size_t)blockIdx.x * stride, (unsigned long long)blockIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (MODE == 1) {
      t[(size_t)blockIdx.x * stride] = blockIdx.x;
      t2[(size_t)blockIdx.x * stride] = blockIdx.x;
    } else {
      __builtin_nontemporal_store(blockIdx.x, t + (size_t)blockIdx.x * stride);
      __builtin_nontemporal_store((unsigned long long)blockIdx.x, t2 + (size_t)blockIdx.x * stride);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}
int main() {
  unsigned* t; unsigned long long* t2; float* sink;
  hipMalloc(&t, 16384 * 64 * 4); hipMalloc(&t2, 16384 * 64 * 8); hipMalloc(&sink, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode)
    for (int G : {977, 3907, 15625})
      for (int stride : {1, 4, 16, 32, 64}) {
        auto run = [&] {
          if (mode == 0) k_store<0><<<G, 64>>>(t, t2, stride, sink);
          else if (mode == 1) k_store<1><<<G, 64>>>(t, t2, stride, sink);
          else k_store<2><<<G, 64>>>(t, t2, stride, sink);
        };
        for (int w = 0; w < 3; ++w) run();
        hipEventRecord(e0);
        for (int r = 0; r < 20; ++r) run();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d (0 sc1 atomic store, 1 plain, 2 nontemporal) G %5d stride %2d words: %.2f us per launch\n", mode, G, stride, ms * 1e3 / 20);
      }
  return 0;
}
