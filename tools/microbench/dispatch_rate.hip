// Workgroup dispatch rate on MI355X: time of a near-empty kernel vs grid size and workgroup size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k_empty(float* out, int spin) {
  float v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  if (v == -1.0f) out[blockIdx.x] = v;
}
int main() {
  float* out; hipMalloc(&out, 1 << 20);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int spin : {0, 2000}) for (int block : {64, 128, 256, 512, 1024}) for (int grid : {512, 1024, 2048, 4096, 8192, 16384}) {
    std::vector<float> ts;
    for (int rep = 0; rep < 12; ++rep) {
      hipEventRecord(a);
      hipLaunchKernelGGL(k_empty, dim3(grid), dim3(block), 0, 0, out, spin);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    printf("spin %d block %4d grid %5d : %.1f us  (%.1f WG/us)\n", spin, block, grid, ts[ts.size()/2], grid / ts[ts.size()/2]);
  }
  return 0;
}
