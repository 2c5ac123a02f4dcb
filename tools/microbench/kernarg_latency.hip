// Latency of dependent scalar reads of kernel arguments: by-value kernarg segment vs a device-memory copy of the same table.
//   hipcc --offload-arch=gfx950 -O3 -o kernarg_latency kernarg_latency.hip && ./kernarg_latency
// Each hop lands on a new 64-byte line (idx[i] = i + 16 ...); block 0 / thread 0 stamps wall_clock64 (100 MHz) around the chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct Big { int idx[960]; };  // 3840 B (< 4 KB kernarg limit)
__global__ void k_kernarg(Big b, long long* out, int hops, int slot) {
  long long t0 = wall_clock64();
  int i = 0;
  for (int h = 0; h < hops; ++h) i = b.idx[i];
  long long t1 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[2 * slot] = t1 - t0; out[2 * slot + 1] = i; }
}
__global__ void k_devmem(const Big* __restrict__ b, long long* out, int hops, int slot) {
  long long t0 = wall_clock64();
  int i = 0;
  for (int h = 0; h < hops; ++h) i = b->idx[i];
  long long t1 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[2 * slot] = t1 - t0; out[2 * slot + 1] = i; }
}
__global__ void k_devmem_vec(const Big* __restrict__ b, long long* out, int hops, int slot) {
  long long t0 = wall_clock64();
  int i = threadIdx.x & 0;  // not provably uniform: vector loads
  i += (int)(threadIdx.x >> 10);
  for (int h = 0; h < hops; ++h) i = __builtin_nontemporal_load(&b->idx[i]);
  long long t1 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) { out[2 * slot] = t1 - t0; out[2 * slot + 1] = i; }
}
__global__ void k_flush(float* p, int n) {  // touch 64 MB: evict L2 / MALL between measurements
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] += 1.f;
}
int main() {
  Big h;
  for (int i = 0; i < 960; ++i) h.idx[i] = (i + 16 * 7) % 960;  // 7 lines ahead: every hop a new line for 8 hops
  Big* d;
  long long* out;
  float* junk;
  const int NJ = 128 << 20;
  hipMalloc(&d, sizeof(Big));
  hipMalloc(&out, 4096);
  hipMalloc(&junk, (size_t)NJ * 4);
  hipMemset(junk, 0, (size_t)NJ * 4);
  hipMemcpy(d, &h, sizeof(Big), hipMemcpyHostToDevice);
  const int hopsv[5] = {0, 1, 2, 4, 8};
  for (int blocks : {1, 256}) {
    for (int mode = 0; mode < 3; ++mode) {
      std::vector<double> best(5, 1e9), sum(5, 0);
      const int reps = 20;
      for (int r = 0; r < reps; ++r)
        for (int k = 0; k < 5; ++k) {
          hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, 0, junk, NJ);
          if (mode == 0) hipLaunchKernelGGL(k_kernarg, dim3(blocks), dim3(256), 0, 0, h, out, hopsv[k], k);
          else if (mode == 1) hipLaunchKernelGGL(k_devmem, dim3(blocks), dim3(256), 0, 0, d, out, hopsv[k], k);
          else hipLaunchKernelGGL(k_devmem_vec, dim3(blocks), dim3(256), 0, 0, d, out, hopsv[k], k);
          hipDeviceSynchronize();
          long long o[2];
          hipMemcpy(o, out + 2 * k, 16, hipMemcpyDeviceToHost);
          const double us = o[0] / 100.0;
          if (us < best[k]) best[k] = us;
          sum[k] += us;
        }
      printf("blocks=%3d %-22s", blocks, mode == 0 ? "kernarg by value" : mode == 1 ? "device memory (scalar)" : "device memory (vector)");
      for (int k = 0; k < 5; ++k) printf("  hops=%d: min %.2f avg %.2f us", hopsv[k], best[k], sum[k] / reps);
      printf("\n");
    }
  }
  return 0;
}
