// Operand / result layout of v_mfma_f32_4x4x4_16b_bf16 (16 independent 4x4x4 blocks per wave), checked against the hypothesis
//   block b = lane / 4;  A: lane 4 b + i holds A_b[i][k = 0..3];  B: lane 4 b + j holds B_b[k = 0..3][j];  D: lane 4 b + j, register i = D_b[i][j]
//   hipcc --offload-arch=gfx950 -O3 -o mfma4_layout mfma4_layout.hip && ./mfma4_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const s16x4* a, const s16x4* b, f32x4* o) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
  o[threadIdx.x] = acc;
}
static unsigned short bf(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)(u >> 16); }
int main() {
  float A[64][4], B[64][4];
  unsigned short ha[64][4], hb[64][4];
  srand(1);
  for (int l = 0; l < 64; ++l)
    for (int k = 0; k < 4; ++k) { A[l][k] = (float)(rand() % 7 - 3); B[l][k] = (float)(rand() % 5 - 2); ha[l][k] = bf(A[l][k]); hb[l][k] = bf(B[l][k]); }
  unsigned short *da, *db; float* dout;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&db, sizeof(hb)); hipMalloc(&dout, 64 * 16);
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof(hb), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (const s16x4*)da, (const s16x4*)db, (f32x4*)dout);
  float D[64][4];
  hipMemcpy(D, dout, sizeof(D), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 4; ++i) {
      const int b = l / 4;
      float ref = 0.f;
      for (int kk = 0; kk < 4; ++kk) ref += A[4 * b + i][kk] * B[l][kk];
      if (ref != D[l][i]) ++bad;
    }
  printf("hypothesis (block = lane/4, A row i from lane 4b+i, B column j = lane%%4, D[lane][reg i]): %d mismatches of 256\n", bad);
  return bad != 0;
}
