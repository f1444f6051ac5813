// store16_out (gjx_device.hpp) against plain stores: every 16-byte slot of a buffer written through the write-through
// builtin by divergent lanes, at several buffer offsets; run on the GPU box: hipcc --offload-arch=gfx950 -O3 -I../../genjax-chi_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "gjx_device.hpp"
__global__ void k(uint32_t* out, uint64_t n4, int mode) {
  const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (i >= n4) return;
  if (mode == 1 && (i % 3) == 0) return;           // divergent: some lanes skip
  if (mode == 2 && (threadIdx.x & 63) < 5) return;  // the first lanes of every wave skip
  gjx::store16_out(out + 4 * i, make_uint4((uint32_t)i, 1u, 2u, (uint32_t)(i >> 3)), true);
}
int main() {
  const uint64_t n4 = (1ull << 22) + 77;
  for (uint64_t shift : {0ull, 4ull, 1ull << 20, (1ull << 28) + 16}) {
    uint32_t* d = nullptr;
    if (hipMalloc(&d, (n4 * 4 + shift) * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (int mode = 0; mode < 3; ++mode) {
      hipMemset(d, 0xff, (n4 * 4 + shift) * 4);
      k<<<(unsigned)((n4 + 255) / 256), 256>>>(d + shift, n4, mode);
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed shift %llu mode %d\n", (unsigned long long)shift, mode); return 1; }
      std::vector<uint32_t> h(n4 * 4);
      hipMemcpy(h.data(), d + shift, n4 * 16, hipMemcpyDeviceToHost);
      uint64_t bad = 0;
      for (uint64_t i = 0; i < n4; ++i) {
        const bool skip = (mode == 1 && (i % 3) == 0) || (mode == 2 && (i & 63) < 5);
        const uint32_t e0 = skip ? 0xffffffffu : (uint32_t)i, e3 = skip ? 0xffffffffu : (uint32_t)(i >> 3);
        if (h[4 * i] != e0 || h[4 * i + 3] != e3 || h[4 * i + 1] != (skip ? 0xffffffffu : 1u)) ++bad;
      }
      printf("ptr %p shift %llu mode %d bad %llu\n", (void*)d, (unsigned long long)shift, mode, (unsigned long long)bad);
    }
    hipFree(d);
  }
  return 0;
}
