// K2 for SMALL batches: register-direct fp32 MFMA GEMM for the weight gradients (no LDS staging).
//
// On MI355X every kernel starts with a cold L2 (kernel boundaries write back / invalidate the per-XCD L2s), and a
// dependent global round trip measures ~2 us inside these launches (tools/ktime.py).  The staged kernel in gemm.hip pays
// one such round trip per K stage (3 for the K = 306 projection of MP3D layer 0).  For the tall-skinny problems of a
// scene-graph batch the operands of a 32x32 output tile are small enough to be requested ALL AT ONCE: every lane issues
// the loads of its whole K slice (the 4 waves of a block split K), then the MFMA chain runs from registers, the four
// partial tiles are summed through LDS in a fixed order (run-to-run identical) and every wave stores a quarter of the
// tile.  One round trip instead of K / 128.
//
// v_mfma_f32_32x32x2_f32 operand layout: lane l supplies A[m = l % 32][k = l / 32] and B[k = l / 32][n = l % 32] of a
// 2-deep k step.  For the TN form (dZ^T * [H | 1]; both operands row-contiguous over m / n, k = node) lane half h takes
// node 2 i + h and the 32 lanes of a half read 128 contiguous bytes: fully coalesced without any transposition.
// (The NT form -- both operands k-contiguous -- is NOT done this way: one row per lane touches 32 cache lines per load
// instruction and thrashes the 32 KB L1; measured 14.5 us against 11.7 us for the LDS-staged kernel.  The layer-0
// projection lives in front.hip instead.)
#include "device_fns.h"

namespace hmp {

KT_DEFINE(gemmd)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// fixed-order sum of the 4 waves' partial 32x32 tiles; wave w ends up with accumulator registers 4w .. 4w+3 summed
__device__ __forceinline__ void reduce4(f32x16& acc, float* red /* [4][16][64] */, int w, int lane, float (&out)[4]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) red[(w * 16 + i) * 64 + lane] = acc[i];
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = 4 * w + j;
    out[j] = (red[(0 * 16 + i) * 64 + lane] + red[(1 * 16 + i) * 64 + lane]) + (red[(2 * 16 + i) * 64 + lane] + red[(3 * 16 + i) * 64 + lane]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// TN, split over node chunks: slab z of C[M, N] = sum_{k in chunk z} A[k, m] * B'[k, n],  B' = [B | 1] (ones column at
// n == n_real when aug_ones).  A = dZ [nodes, ncols], B = H [nodes, F].
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TN_MAXI = 24;  // 2-k MFMA steps per wave and trip: a block covers 4 * 24 * 2 = 192 nodes per trip

// (A 32x64 tile with two accumulators per wave -- A fetched once per 64 output columns -- was measured SLOWER: 28.8 us
// against 13.0 us for the config-2 weight gradients; 72 loads per lane and trip, 2 blocks per CU.)
__global__ __launch_bounds__(256, 2) void gemm_tn_direct_kernel(const TnBatch tb) {
  __shared__ float red[4 * 16 * 64];
  const int blk = blockIdx.x;
  KT(8);
  // the whole problem table in one round trip (the search below walks one argument line per problem otherwise: common.h)
  karg_warm<(sizeof(TnBatch) + 63) / 64>(0, (int)sizeof(TnBatch));
  int pi = 0;
  while (pi + 1 < tb.n && blk >= tb.p[pi + 1].tile_start) ++pi;
  const TnProblem& P = tb.p[pi];
  const int local = blk - P.tile_start;
  // the K chunk is the fastest-varying part of the block id: blocks b and b + 8 share an XCD (speed only)
  const int z = local % P.ksplit, t = local / P.ksplit;
  const int tm = t / P.tiles_n, tn = t % P.tiles_n;
  const int m0 = tm * 32, n0 = tn * 32;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int kbeg = z * P.kchunk, kend = min(P.K, kbeg + P.kchunk);
  // the 4 waves take contiguous runs of 2-k steps
  const int n_steps = (kend - kbeg + 1) >> 1;
  const int per = (n_steps + 3) >> 2;
  const int s0 = w * per, s1 = min(n_steps, s0 + per);
  const int am = min(m0 + r, P.M - 1);
  const int bn = n0 + r;
  const bool b_real = bn < P.n_real;
  const bool b_one = P.aug_ones && bn == P.n_real;
  const int bnc = b_real ? bn : 0;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int sb0 = s0; sb0 < s1; sb0 += TN_MAXI) {  // one trip for chunks <= 192 nodes
    float av[TN_MAXI], bv[TN_MAXI];
#pragma unroll
    for (int i = 0; i < TN_MAXI; ++i) {
      const int k = kbeg + 2 * (sb0 + i) + h;
      const bool live = (sb0 + i < s1) && k < kend;
      const int kc = live ? k : kbeg;  // clamped: in range, masked below
      const float a = P.A[(int64_t)kc * P.lda + am];
      const float b = P.B[(int64_t)kc * P.ldb + bnc];
      av[i] = live ? a : 0.f;
      bv[i] = live ? (b_real ? b : (b_one ? 1.f : 0.f)) : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TN_MAXI; ++i)
      if (sb0 + i < s1) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
  }
  float out[4];
  reduce4(acc, red, w, lane, out);
  float* C = P.C + (int64_t)z * P.slab_stride;
  const int col = n0 + (lane & 31);
  if (col < P.N) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = m0 + j + 8 * w + 4 * h;
      if (row < P.M) C[(int64_t)row * P.ldc + col] = out[j];
    }
  }
  KT(9);
}

// chooses the node chunk (<= 384 nodes per block, <= max_slabs slabs per problem); returns HMP_E_ARG-free "not
// applicable" through *ok = 0 when a problem needs more slabs than allowed (the caller then uses the staged kernel)
int gemm_tn_direct_launch(TnBatch& tb, int max_slabs, int* ok, hipStream_t st) {
  int start = 0;
  *ok = 1;
  for (int i = 0; i < tb.n; ++i) {
    TnProblem& P = tb.p[i];
    P.tiles_m = cdiv(P.M, 32);
    P.tiles_n = cdiv(P.N, 32);
    // at most two trips of 192 nodes per block -- four when the output alone already gives every CU several tiles (wide GAT
    // layers: 595 tiles): half the slabs to write here and to sum in the gradient un-pack
    const int trips = (P.tiles_m * P.tiles_n >= 512) ? 4 : 2;
    int ks = cdiv(P.K, 2 * 4 * TN_MAXI * trips);
    if (ks < 1) ks = 1;
    if (ks > max_slabs) { *ok = 0; return HMP_OK; }
    // more slabs than necessary when the launch would otherwise be small: shorter MFMA chains, same round trip
    while (ks * 2 <= max_slabs && cdiv(P.K, ks * 2) >= 32 && P.tiles_m * P.tiles_n * ks < 256) ks *= 2;
    int kchunk = cdiv(cdiv(P.K, ks), 2) * 2;
    if (kchunk < 2) kchunk = 2;
    ks = P.K > 0 ? cdiv(P.K, kchunk) : 1;
    P.ksplit = ks;
    P.kchunk = kchunk;
    P.tile_start = start;
    start += P.tiles_m * P.tiles_n * ks;
  }
  tb.total_tiles = start;
  if (start == 0) return HMP_OK;
  hipLaunchKernelGGL(gemm_tn_direct_kernel, dim3(start), dim3(256), 0, st, tb);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

}  // namespace hmp
