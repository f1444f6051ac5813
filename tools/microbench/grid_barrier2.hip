// Variant of grid_barrier.hip WITHOUT fences: the exchanged data moves with agent-scope relaxed atomic stores / loads (they
// bypass the non-coherent part of the caches by themselves), the barrier counters are relaxed atomics, and ordering is by
// waiting for the outstanding stores (s_waitcnt) before arriving.  Does the hand-over still check out, and what does a round cost?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int kGroups = 32;
struct Bar { unsigned int cnt[kGroups * 64]; unsigned int root[64]; unsigned int gen[64]; };

__device__ __forceinline__ bool grid_barrier(Bar* b, unsigned int round, unsigned int G, unsigned int* err) {
  __builtin_amdgcn_s_waitcnt(0);  // this wave's stores have been issued and acknowledged
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    const unsigned int g = blockIdx.x % kGroups;
    const unsigned int members = (G - g + kGroups - 1) / kGroups;
    if (__hip_atomic_fetch_add(&b->cnt[g * 64], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == (round + 1) * members) {
      const unsigned int ng = G < (unsigned)kGroups ? G : (unsigned)kGroups;
      if (__hip_atomic_fetch_add(&b->root[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == (round + 1) * ng)
        __hip_atomic_store(&b->gen[0], round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    unsigned int spins = 0;
    while (__hip_atomic_load(&b->gen[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round + 1) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) { ok = false; atomicExch(err, 1u); break; }
    }
  }
  __syncthreads();
  return ok;
}

__global__ __launch_bounds__(256) void k_rounds(Bar* b, float* buf0, float* buf1, int rounds, unsigned int* err, int work) {
  const unsigned int G = gridDim.x;
  float acc = 0.0f;
  for (int r = 0; r < rounds; ++r) {
    float* wr = (r & 1) ? buf1 : buf0;
    float* rd = (r & 1) ? buf0 : buf1;
    if (r > 0) {
      const unsigned int src = (blockIdx.x + 427u) % G;
      float v[4];
      for (int k = 0; k < 4; ++k) v[k] = __hip_atomic_load(rd + (size_t)src * 1024 + 4 * threadIdx.x + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v[0] != (float)(r - 1) + (float)src) atomicExch(err, 2u);
      acc += v[1];
    }
    float x = acc;
    for (int i = 0; i < work; ++i) x = x * 1.0001f + 0.5f;
    const float out[4] = {(float)r + (float)blockIdx.x, x, 0.f, 0.f};
    for (int k = 0; k < 4; ++k) __hip_atomic_store(wr + (size_t)blockIdx.x * 1024 + 4 * threadIdx.x + k, out[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!grid_barrier(b, (unsigned)r, G, err)) return;
  }
}

int main() {
  Bar* bar; float *b0, *b1; unsigned int* err;
  (void)hipMalloc(&bar, sizeof(Bar)); (void)hipMalloc(&b0, 4096 * 1024 * 4); (void)hipMalloc(&b1, 4096 * 1024 * 4); (void)hipMalloc(&err, 4);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int G : {256, 977}) for (int work : {0}) for (int rounds : {1, 101, 201}) {
    std::vector<float> ts; unsigned int herr = 0;
    for (int rep = 0; rep < 7; ++rep) {
      (void)hipMemset(bar, 0, sizeof(Bar)); (void)hipMemset(err, 0, 4); (void)hipDeviceSynchronize();
      (void)hipEventRecord(a);
      void* args[] = {&bar, &b0, &b1, &rounds, &err, &work};
      hipError_t e = hipLaunchCooperativeKernel((void*)k_rounds, dim3(G), dim3(256), args, 0, 0);
      if (e != hipSuccess) { printf("cooperative launch failed G=%d: %s\n", G, hipGetErrorString(e)); break; }
      (void)hipEventRecord(b); (void)hipEventSynchronize(b);
      float ms; (void)hipEventElapsedTime(&ms, a, b); ts.push_back(ms * 1e3f);
      unsigned int h; (void)hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost); herr |= h;
    }
    if (ts.empty()) continue;
    std::sort(ts.begin(), ts.end());
    printf("G %4d rounds %3d : %8.1f us total  err %u\n", G, rounds, ts[ts.size() / 2], herr);
  }
  return 0;
}
