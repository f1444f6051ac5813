// What does a heavy wave cost at launch?  977 x 256 threads, every kernel reads one argument and exits; variants differ in the
// resources the dispatcher must reserve: none / 20 KB of LDS / ~120 VGPRs / both.  Wave lifetime = SQ_WAVE_CYCLES / SQ_WAVES
// (rocprofv3 --pmc), launch-to-launch time by HIP events.
//   hipcc --offload-arch=gfx950 -O3 -o wave_start wave_start.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_plain(uint32_t* out, int x) {
  if (x == 12345) out[threadIdx.x] = 1;
}
__global__ __launch_bounds__(256) void k_lds(uint32_t* out, int x) {
  __shared__ uint32_t pad[5120];  // 20 KB
  if (x == 12345) { pad[threadIdx.x] = x; __syncthreads(); out[threadIdx.x] = pad[(threadIdx.x * 7) & 5119]; }
}
template <bool LDS>
__global__ __launch_bounds__(256) void k_vgpr(uint32_t* out, const uint32_t* in, int x) {
  __shared__ uint32_t pad[LDS ? 5120 : 1];
  if (x == 12345) {  // (never taken: the registers are reserved all the same)
    const int t = threadIdx.x;
    uint32_t va0, va1, va2, va3, va4, va5, va6, va7, va8, va9, vaa, vab, vac, vad, vae, vaf, vag, vah, vai, vaj, vak, val, vam, van, vao, vap, vaq, var, vas, vau;
    uint32_t vb0, vb1, vb2, vb3, vb4, vb5, vb6, vb7, vb8, vb9, vba, vbb, vbc, vbd, vbe, vbf, vbg, vbh, vbi, vbj, vbk, vbl, vbm, vbn, vbo, vbp, vbq, vbr, vbs, vbu;
    uint32_t vc0, vc1, vc2, vc3, vc4, vc5, vc6, vc7, vc8, vc9, vca, vcb, vcc, vcd, vce, vcf, vcg, vch, vci, vcj, vck, vcl, vcm, vcn, vco, vcp, vcq, vcr, vcs, vcu;
    uint32_t vd0, vd1, vd2, vd3, vd4, vd5, vd6, vd7, vd8, vd9, vda, vdb, vdc, vdd, vde, vdf, vdg, vdh, vdi, vdj, vdk, vdl, vdm, vdn, vdo, vdp, vdq, vdr, vds, vdu;
    asm volatile("" : "=v"(va0), "=v"(va1), "=v"(va2), "=v"(va3), "=v"(va4), "=v"(va5), "=v"(va6), "=v"(va7), "=v"(va8), "=v"(va9), "=v"(vaa), "=v"(vab), "=v"(vac), "=v"(vad), "=v"(vae), "=v"(vaf), "=v"(vag), "=v"(vah), "=v"(vai), "=v"(vaj), "=v"(vak), "=v"(val), "=v"(vam), "=v"(van), "=v"(vao), "=v"(vap), "=v"(vaq), "=v"(var), "=v"(vas), "=v"(vau));
    asm volatile("" : "=v"(vb0), "=v"(vb1), "=v"(vb2), "=v"(vb3), "=v"(vb4), "=v"(vb5), "=v"(vb6), "=v"(vb7), "=v"(vb8), "=v"(vb9), "=v"(vba), "=v"(vbb), "=v"(vbc), "=v"(vbd), "=v"(vbe), "=v"(vbf), "=v"(vbg), "=v"(vbh), "=v"(vbi), "=v"(vbj), "=v"(vbk), "=v"(vbl), "=v"(vbm), "=v"(vbn), "=v"(vbo), "=v"(vbp), "=v"(vbq), "=v"(vbr), "=v"(vbs), "=v"(vbu));
    asm volatile("" : "=v"(vc0), "=v"(vc1), "=v"(vc2), "=v"(vc3), "=v"(vc4), "=v"(vc5), "=v"(vc6), "=v"(vc7), "=v"(vc8), "=v"(vc9), "=v"(vca), "=v"(vcb), "=v"(vcc), "=v"(vcd), "=v"(vce), "=v"(vcf), "=v"(vcg), "=v"(vch), "=v"(vci), "=v"(vcj), "=v"(vck), "=v"(vcl), "=v"(vcm), "=v"(vcn), "=v"(vco), "=v"(vcp), "=v"(vcq), "=v"(vcr), "=v"(vcs), "=v"(vcu));
    asm volatile("" : "=v"(vd0), "=v"(vd1), "=v"(vd2), "=v"(vd3), "=v"(vd4), "=v"(vd5), "=v"(vd6), "=v"(vd7), "=v"(vd8), "=v"(vd9), "=v"(vda), "=v"(vdb), "=v"(vdc), "=v"(vdd), "=v"(vde), "=v"(vdf), "=v"(vdg), "=v"(vdh), "=v"(vdi), "=v"(vdj), "=v"(vdk), "=v"(vdl), "=v"(vdm), "=v"(vdn), "=v"(vdo), "=v"(vdp), "=v"(vdq), "=v"(vdr), "=v"(vds), "=v"(vdu));
    if (LDS) { pad[t] = x; __syncthreads(); out[t + 300] = pad[(t * 7) & 5119]; }
    asm volatile("" :: "v"(va0), "v"(va1), "v"(va2), "v"(va3), "v"(va4), "v"(va5), "v"(va6), "v"(va7), "v"(va8), "v"(va9), "v"(vaa), "v"(vab), "v"(vac), "v"(vad), "v"(vae), "v"(vaf), "v"(vag), "v"(vah), "v"(vai), "v"(vaj), "v"(vak), "v"(val), "v"(vam), "v"(van), "v"(vao), "v"(vap), "v"(vaq), "v"(var), "v"(vas), "v"(vau));
    asm volatile("" :: "v"(vb0), "v"(vb1), "v"(vb2), "v"(vb3), "v"(vb4), "v"(vb5), "v"(vb6), "v"(vb7), "v"(vb8), "v"(vb9), "v"(vba), "v"(vbb), "v"(vbc), "v"(vbd), "v"(vbe), "v"(vbf), "v"(vbg), "v"(vbh), "v"(vbi), "v"(vbj), "v"(vbk), "v"(vbl), "v"(vbm), "v"(vbn), "v"(vbo), "v"(vbp), "v"(vbq), "v"(vbr), "v"(vbs), "v"(vbu));
    asm volatile("" :: "v"(vc0), "v"(vc1), "v"(vc2), "v"(vc3), "v"(vc4), "v"(vc5), "v"(vc6), "v"(vc7), "v"(vc8), "v"(vc9), "v"(vca), "v"(vcb), "v"(vcc), "v"(vcd), "v"(vce), "v"(vcf), "v"(vcg), "v"(vch), "v"(vci), "v"(vcj), "v"(vck), "v"(vcl), "v"(vcm), "v"(vcn), "v"(vco), "v"(vcp), "v"(vcq), "v"(vcr), "v"(vcs), "v"(vcu));
    asm volatile("" :: "v"(vd0), "v"(vd1), "v"(vd2), "v"(vd3), "v"(vd4), "v"(vd5), "v"(vd6), "v"(vd7), "v"(vd8), "v"(vd9), "v"(vda), "v"(vdb), "v"(vdc), "v"(vdd), "v"(vde), "v"(vdf), "v"(vdg), "v"(vdh), "v"(vdi), "v"(vdj), "v"(vdk), "v"(vdl), "v"(vdm), "v"(vdn), "v"(vdo), "v"(vdp), "v"(vdq), "v"(vdr), "v"(vds), "v"(vdu));
    out[t] = va0 + vb1 + vc2 + vd3;
  }
}
// ... and the argument block: 848 bytes by value, 17 scattered fields read at entry (the SMC step's shape)
struct Big { uint64_t w[106]; };
__global__ __launch_bounds__(256) void k_bigargs(uint32_t* out, Big b) {
  asm volatile("" ::"s"(b.w[0]), "s"(b.w[2]), "s"(b.w[5]), "s"(b.w[9]), "s"(b.w[10]), "s"(b.w[14]), "s"(b.w[16]), "s"(b.w[67]));
  asm volatile("" ::"s"(b.w[68]), "s"(b.w[73]), "s"(b.w[74]), "s"(b.w[12]), "s"(b.w[13]), "s"(b.w[95]), "s"(b.w[100]), "s"(b.w[102]), "s"(b.w[104]));
  if (b.w[0] + b.w[67] + b.w[104] == 12345) out[threadIdx.x] = 1;
}
__global__ __launch_bounds__(256) void k_bigargs_one(uint32_t* out, Big b) {
  if (b.w[0] == 12345) out[threadIdx.x] = 1;
}
template <class F>
static float per_launch_us(F f) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) f();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int i = 0; i < 200; ++i) f();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1000.0f / 200.0f;
}
int main() {
  uint32_t* d;
  (void)hipMalloc(&d, 1 << 20);
  printf("plain      : %.2f us per launch\n", per_launch_us([&] { k_plain<<<977, 256>>>(d, 1); }));
  printf("20 KB LDS  : %.2f us per launch\n", per_launch_us([&] { k_lds<<<977, 256>>>(d, 1); }));
  printf("120 VGPRs  : %.2f us per launch\n", per_launch_us([&] { k_vgpr<false><<<977, 256>>>(d, d, 1); }));
  printf("both       : %.2f us per launch\n", per_launch_us([&] { k_vgpr<true><<<977, 256>>>(d, d, 1); }));
  Big bg{};
  printf("848-B args, 17 fields read : %.2f us per launch\n", per_launch_us([&] { k_bigargs<<<977, 256>>>(d, bg); }));
  printf("848-B args, 1 field read   : %.2f us per launch\n", per_launch_us([&] { k_bigargs_one<<<977, 256>>>(d, bg); }));
  return 0;
}
