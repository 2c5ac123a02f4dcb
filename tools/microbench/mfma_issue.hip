// Issue rate of v_mfma_f32_32x32x2_f32 from ONE wave with NACC independent accumulators, and with several waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_issue mfma_issue.hip && ./mfma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void k(long long* out, int n, float* sink) {
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j)
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  const float a = threadIdx.x * 1e-3f, b = 1.0f;
  long long w0 = wall_clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + j, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j)
    for (int i = 0; i < 16; ++i) s += acc[j][i];
  long long w1 = wall_clock64();
  if (s == 12345.f) sink[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = w1 - w0;
}
template <int NACC>
void run(long long* out, float* sink) {
  for (int threads : {64, 256, 512, 1024}) {
    const int n = 8192;
    double best = 1e30;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 5; ++r) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, n, sink);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double ns_per = best * 1e6 / ((double)n * NACC);  // whole-kernel time (every wave done) per MFMA of one wave
    const int waves_per_simd = threads >= 256 ? threads / 256 : 1;
    printf("accumulators/wave=%d waves/SIMD=%d: %.1f ns per MFMA per wave = %.1f cycles; SIMD rate one MFMA per %.1f cycles -> %.0f TFLOP/s chip-wide\n", NACC,
           waves_per_simd, ns_per, ns_per * 2.4, ns_per * 2.4 / waves_per_simd, 4096.0 / (ns_per / waves_per_simd) * 4 * 256 / 1e3);
  }
}
int main() {
  long long* out; float* sink;
  hipMalloc(&out, 64); hipMalloc(&sink, 64);
  run<1>(out, sink);
  run<2>(out, sink);
  run<4>(out, sink);
  return 0;
}
