// Issue rate of the vector instructions a bf16-row accumulation can be built from (v_add_f32, v_pk_add_f32, v_dot2c_f32_bf16,
// v_perm_b32, v_lshlrev_b32): 16 independent chains per wave, 1 / 2 / 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ void k(int n, float* sink, const unsigned* src) {
  float acc[16];
  f32x2 pacc[16];
  unsigned u[16];
  for (int j = 0; j < 16; ++j) { acc[j] = threadIdx.x * 1e-3f + j; pacc[j] = f32x2{acc[j], acc[j] + 1.f}; u[j] = src[(threadIdx.x + j) & 63]; }
  const unsigned ones = 0x3f803f80u;
  const float one = src[1] ? 1.0f : 2.0f;
  const f32x2 pone = {one, one};
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[j]) : "v"(one));
      if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pacc[j]) : "v"(pone));
      if (OP == 2) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc[j]) : "v"(u[j]), "v"(ones));
      if (OP == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[j]) : "v"(ones), "v"(0x05040100u));
      if (OP == 4) asm volatile("v_lshlrev_b32 %0, 16, %0" : "+v"(u[j]));
    }
  }
  float s = 0.f;
  for (int j = 0; j < 16; ++j) s += acc[j] + pacc[j][0] + pacc[j][1] + (float)u[j];
  if (s == 12345.f) sink[0] = s;
}
template <int OP>
void run(const char* name, float* sink, unsigned* src) {
  for (int threads : {256, 512, 1024}) {
    const int n = 4096;
    double best = 1e30;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 5; ++r) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, n, sink, src);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const int wps = threads / 256;
    const double cyc = best * 1e-3 * 2.4e9 / ((double)n * 16 * wps);  // cycles of one SIMD per instruction (2.4 GHz nominal)
    printf("%-18s waves/SIMD=%d: %.2f cycles per instruction per SIMD\n", name, wps, cyc);
  }
}
int main() {
  float* sink; unsigned* src;
  hipMalloc(&sink, 64); hipMalloc(&src, 256); hipMemset(src, 0x3f, 256);
  run<0>("v_add_f32", sink, src);
  run<1>("v_pk_add_f32", sink, src);
  run<2>("v_dot2c_f32_bf16", sink, src);
  run<3>("v_perm_b32", sink, src);
  run<4>("v_lshlrev_b32", sink, src);
  return 0;
}
