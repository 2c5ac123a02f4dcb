// Shader clock during short kernels: clock64() (s_memtime) ticks per wall_clock64() tick (100 MHz), and the time of a chain of
// dependent fp32 MFMAs (32x32x2: 16 passes = 64 cycles each) -- both in a 1-block launch and in a 256-block launch.
//   hipcc --offload-arch=gfx950 -O3 -o shader_clock shader_clock.hip && ./shader_clock
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(long long* out, int n_mfma, float* sink) {
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const float a = threadIdx.x * 1e-3f, b = 1.0f;
  long long w0 = wall_clock64(), c0 = clock64();
  for (int i = 0; i < n_mfma; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i];
  long long c1 = clock64(), w1 = wall_clock64();
  if (s == 12345.f) sink[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = w1 - w0; out[1] = c1 - c0; }
}
int main() {
  long long* out; float* sink;
  hipMalloc(&out, 64); hipMalloc(&sink, 64);
  for (int blocks : {1, 256})
    for (int threads : {64, 256, 1024})
      for (int n : {256, 4096}) {
        long long best[2] = {1LL << 60, 0};
        for (int r = 0; r < 10; ++r) {
          hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, n, sink);
          hipDeviceSynchronize();
          long long o[2]; hipMemcpy(o, out, 16, hipMemcpyDeviceToHost);
          if (o[0] < best[0]) { best[0] = o[0]; best[1] = o[1]; }
        }
        printf("blocks=%3d threads=%4d mfma=%4d: %.2f us, clock64 ticks %lld (%.1f MHz), %.1f ns per MFMA => %.2f GHz if 64 cycles\n", blocks, threads, n,
               best[0] / 100.0, best[1], best[1] / (best[0] / 100.0), best[0] * 10.0 / n, 64.0 / (best[0] * 10.0 / n));
      }
  return 0;
}
