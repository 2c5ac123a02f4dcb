// Cost of cold instruction fetch: a loop whose body is N_BODY x 8 independent v_fma (straight-line, no memory access) runs three
// times; trip 1 fetches the code, trips 2 and 3 find it in the instruction cache (if it fits: 64 KB per pair of CUs).
//   hipcc --offload-arch=gfx950 -O3 -o icache_cold icache_cold.hip && ./icache_cold
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N_BODY>
__global__ void k(long long* out, float* sink, float y, int trips) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
  long long t[4];
  t[0] = wall_clock64();
  for (int trip = 0; trip < trips; ++trip) {
#pragma unroll
    for (int j = 0; j < N_BODY; ++j) {
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = __builtin_fmaf(x[i], y, 1.0f + j);  // a different literal per step: no rolling back
    }
    asm volatile("" ::: "memory");
    if (trip < 3) t[trip + 1] = wall_clock64();
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += x[i];
  if (s == 12345.f) sink[0] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int i = 0; i < 3; ++i) out[i] = t[i + 1] - t[i];
}
template <int N_BODY>
void run(long long* out, float* sink, const char* name) {
  for (int blocks : {1, 256}) {
    long long best[3] = {1LL << 60, 1LL << 60, 1LL << 60};
    for (int r = 0; r < 10; ++r) {
      hipLaunchKernelGGL(k<N_BODY>, dim3(blocks), dim3(256), 0, 0, out, sink, 1.0001f, 3);
      hipDeviceSynchronize();
      long long o[3];
      hipMemcpy(o, out, 24, hipMemcpyDeviceToHost);
      for (int i = 0; i < 3; ++i) best[i] = o[i] < best[i] ? o[i] : best[i];
    }
    printf("%s (%d fma instructions) blocks=%3d: trip 1 %.2f us, trip 2 %.2f us, trip 3 %.2f us\n", name, N_BODY * 8, blocks, best[0] / 100.0, best[1] / 100.0,
           best[2] / 100.0);
  }
}
int main() {
  long long* out; float* sink;
  hipMalloc(&out, 64); hipMalloc(&sink, 64);
  run<128>(out, sink, "8 KB body ");
  run<512>(out, sink, "32 KB body");
  run<1024>(out, sink, "64 KB body");
  return 0;
}
