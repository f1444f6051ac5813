// Cost of a hierarchical grid barrier on MI355X (8 XCDs): a persistent kernel of G workgroups runs R barrier rounds;
// between rounds every workgroup writes 4 KB and reads 4 KB written by ANOTHER workgroup in the previous round (checked),
// so the barrier includes the release / acquire fences that make the data visible across XCDs.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int kGroups = 32;  // first-level counters, each on its own 256-byte line
struct Bar {
  unsigned int cnt[kGroups * 64];
  unsigned int root[64];
  unsigned int gen[64];
};

__device__ __forceinline__ bool grid_barrier(Bar* b, unsigned int round, unsigned int G, unsigned int* err) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();  // release: this workgroup's writes
    const unsigned int g = blockIdx.x % kGroups;
    const unsigned int members = (G - g + kGroups - 1) / kGroups;
    const unsigned int target = (round + 1) * members;
    if (atomicAdd(&b->cnt[g * 64], 1u) + 1 == target) {
      const unsigned int ng = G < (unsigned)kGroups ? G : (unsigned)kGroups;
      if (atomicAdd(&b->root[0], 1u) + 1 == (round + 1) * ng) {
        __threadfence();
        __hip_atomic_store(&b->gen[0], round + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    unsigned int spins = 0;
    while (__hip_atomic_load(&b->gen[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < round + 1) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 22)) { ok = false; atomicExch(err, 1u); break; }
    }
    __threadfence();  // acquire
  }
  __syncthreads();
  return ok;
}

__global__ __launch_bounds__(256) void k_rounds(Bar* b, float* buf0, float* buf1, int rounds, unsigned int* err, int work) {
  const unsigned int G = gridDim.x;
  float acc = 0.0f;
  for (int r = 0; r < rounds; ++r) {
    float* wr = (r & 1) ? buf1 : buf0;
    const float* rd = (r & 1) ? buf0 : buf1;
    if (r > 0) {  // read what workgroup (b + 7 * 61) % G wrote in the previous round
      const unsigned int src = (blockIdx.x + 427u) % G;
      const float4 v = reinterpret_cast<const float4*>(rd + (size_t)src * 1024)[threadIdx.x];
      if (v.x != (float)(r - 1) + (float)src) atomicExch(err, 2u);
      acc += v.y;
    }
    float x = acc;
    for (int i = 0; i < work; ++i) x = x * 1.0001f + 0.5f;
    reinterpret_cast<float4*>(wr + (size_t)blockIdx.x * 1024)[threadIdx.x] = make_float4((float)r + (float)blockIdx.x, x, 0.f, 0.f);
    if (!grid_barrier(b, (unsigned)r, G, err)) return;
  }
}

int main() {
  Bar* bar; float *b0, *b1; unsigned int* err;
  hipMalloc(&bar, sizeof(Bar)); hipMalloc(&b0, 4096 * 1024 * 4); hipMalloc(&b1, 4096 * 1024 * 4); hipMalloc(&err, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int G : {256, 977, 1536}) for (int work : {0, 400}) for (int rounds : {1, 101, 201}) {
    std::vector<float> ts;
    unsigned int herr = 0;
    for (int rep = 0; rep < 7; ++rep) {
      hipMemset(bar, 0, sizeof(Bar)); hipMemset(err, 0, 4);
      hipDeviceSynchronize();
      hipEventRecord(a);
      void* args[] = {&bar, &b0, &b1, &rounds, &err, &work};
      hipError_t e = hipLaunchCooperativeKernel((void*)k_rounds, dim3(G), dim3(256), args, 0, 0);
      if (e != hipSuccess) { printf("cooperative launch failed G=%d: %s\n", G, hipGetErrorString(e)); break; }
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); ts.push_back(ms * 1e3f);
      unsigned int h; hipMemcpy(&h, err, 4, hipMemcpyDeviceToHost); herr |= h;
    }
    if (ts.empty()) continue;
    std::sort(ts.begin(), ts.end());
    printf("G %4d work %3d rounds %3d : %8.1f us total  err %u\n", G, work, rounds, ts[ts.size() / 2], herr);
  }
  return 0;
}
