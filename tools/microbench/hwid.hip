// Where do the waves of a workgroup run?  For the launch shape of the one-filter SMC step (977 workgroups of 256 threads,
// ~16 KB of LDS each: four workgroups per CU) every wave records HW_REG_HW_ID and the XCC id; the host prints, per wave
// index, the histogram of SIMD ids, and for a few CUs the (workgroup, TG_ID, wave -> SIMD) tuples resident on it.
//   hipcc --offload-arch=gfx950 -O3 -o hwid hwid.hip && ./hwid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k(uint32_t* out, int spin) {
  __shared__ uint32_t pad[4096];
  pad[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
  const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
  uint32_t x = pad[(threadIdx.x * 7) & 4095];
  for (int i = 0; i < spin; ++i) x = x * 1664525u + 1013904223u;  // (keep the workgroups resident together)
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = (xcc & 0xf) | (x == 12345u ? 16u : 0u);
  }
}
int main() {
  const int nb = 977;
  uint32_t* d;
  hipMalloc(&d, nb * 4 * 2 * 4);
  k<<<nb, 256>>>(d, 20000);
  hipDeviceSynchronize();
  std::vector<uint32_t> h(nb * 8);
  hipMemcpy(h.data(), d, nb * 32, hipMemcpyDeviceToHost);
  int hist[4][4] = {};
  int same_simd_pairs = 0;
  std::map<uint32_t, std::vector<int>> by_cu;
  for (int b = 0; b < nb; ++b) {
    int simds[4];
    for (int w = 0; w < 4; ++w) {
      const uint32_t hw = h[(b * 4 + w) * 2];
      simds[w] = (hw >> 4) & 3;
      hist[w][simds[w]]++;
    }
    for (int a = 0; a < 4; ++a) for (int c = a + 1; c < 4; ++c) same_simd_pairs += simds[a] == simds[c];
    const uint32_t hw0 = h[b * 8], xcc = h[b * 8 + 1] & 0xf;
    const uint32_t cu = (hw0 >> 8) & 0xf, sh = (hw0 >> 12) & 1, se = (hw0 >> 13) & 7;
    by_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu].push_back(b);
  }
  printf("wave index -> SIMD id histogram (rows: wave 0..3)\n");
  for (int w = 0; w < 4; ++w) printf("  wave %d: %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  printf("pairs of waves of one workgroup on the same SIMD: %d\n", same_simd_pairs);
  printf("distinct CUs: %zu\n", by_cu.size());
  int shown = 0;
  for (auto& kv : by_cu) {
    if (shown++ >= 6) break;
    printf("CU %06x:", kv.first);
    for (int b : kv.second) {
      const uint32_t hw0 = h[b * 8];
      printf("  wg %d tg %u simd[", b, (hw0 >> 16) & 0xf);
      for (int w = 0; w < 4; ++w) printf("%u", (h[(b * 4 + w) * 2] >> 4) & 3);
      printf("] slot[");
      for (int w = 0; w < 4; ++w) printf("%u", h[(b * 4 + w) * 2] & 0xf);
      printf("]");
    }
    printf("\n");
  }
  return 0;
}
