// K2, bf16 compute variant for LARGE batches (BASELINE config 5: hidden 256, 10^6 nodes): grouped GEMM on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
//
// Storage stays fp32 (activations, gradients, weights): operands are read as fp32, rounded to bf16 (round to nearest
// even, v_cvt_pk_bf16_f32) on their way into LDS, and multiplied on the bf16 matrix pipe (2.5 PFLOP/s dense against
// 157 TFLOP/s for fp32 MFMA), so these GEMMs become HBM-bound on their fp32 operands instead of MFMA-bound.  This is an
// explicit precision mode (hmp_net_spec.compute_bf16): results differ from the fp32 path by bf16 input rounding
// (~2^-9 relative per product), so it is NOT used for the 1e-5 parity configurations.
//
// Block = 4 waves, tile 128x128, K stage 64; every wave owns a 64x64 quadrant = 2x2 MFMA tiles (64 accumulator
// registers).  LDS images are [row][k] in bf16 with a pitch of 72 elements (144 bytes: 16-byte aligned rows), so an MFMA
// operand (8 consecutive k of one row) is one ds_read_b128.  A k-contiguous operand ([row][k] in memory) is loaded as
// float4 along k (one 8-byte LDS write each) into a [row][k] image.  A row-contiguous operand ([k][row]: the weight-
// gradient form, both operands) is loaded as float4 along rows into the NATURAL [k][row] image (8-byte writes) and
// transposed by the hardware on the way out: ds_read_b64_tr_b16 hands every lane 4 consecutive k of its row, two of them
// make one MFMA operand.  (Transposing on the way IN cost 16-way bank conflicts with 2-byte column writes -- 18 ms for the
// config-5 weight gradients -- and 37 ms with 4-byte lane-per-row global loads.)  The next stage's global loads are
// issued before the current stage's MFMAs.
//
// Forms: NT (x * W^T), NN (dZ * W, optional activation-derivative epilogue), TN with split-K over node chunks
// (dZ^T * [x | 1]) -- the same GemmProblem contract as gemm.hip.
#include "kernels.h"

namespace hmp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// K stage BK (64 for the 128x128 tile, 32 for the 256x256 tile: 8 / 4 float4 slots per thread, operand and stage);
// [row][k] images have a pitch of BK + 8 bf16 elements (16-byte aligned rows)
template <int NV>
struct StageRegs {
  float4 v[NV];
};

// Block shapes <NT threads, ROWS x ROWS tile>: <256, 128> (4 waves, 2x2, 64x64 per wave) and <512, 256> (8 waves, 4x2,
// 64x128 per wave).  The big tile exists because these GEMMs are bound by L2 -> CU operand traffic (~4.7 TB/s aggregate
// measured): a 128x128 tile re-reads A once per 128 output columns and B once per 128 output rows (12 GB for the config-5
// layer-0 projection), a 256x256 tile halves both.
//  kcontig: slot q covers row q / 16, k4 = (q % 16) * 4      rcontig: slot q covers k = q / (ROWS/4), r4 = (q % (ROWS/4)) * 4
// bf_load_fast only ISSUES the stage's loads (clamped addresses, nothing reads the registers); bf_mask zeroes what lies outside
// the operand when the stage is consumed, one iteration later.  With the masks applied in the loader the compiler had to wait
// for every load right there -- before the MFMAs the prefetch was meant to overlap with (and, with the layout branch inside the
// loop, even between one load and the next).
template <int NT, int ROWS, int BKB>
__device__ __forceinline__ void bf_load_fast(StageRegs<ROWS * BKB / 4 / NT>& t, const float* __restrict__ p, int ld, int kcontig, int r0, int R,
                                             int k0, int kend) {
  constexpr int BNV = ROWS * BKB / 4 / NT;
  const int tid = threadIdx.x;
  if (kcontig) {
#pragma unroll
    for (int i = 0; i < BNV; ++i) {
      const int q = tid + i * NT;
      const int r = q / (BKB / 4), k4 = (q % (BKB / 4)) * 4;
      const int gr = r0 + r, gk = k0 + k4;
      t.v[i] = *reinterpret_cast<const float4*>(p + (int64_t)(gr < R ? gr : R - 1) * ld + (gk < kend ? gk : k0));
    }
  } else {
#pragma unroll
    for (int i = 0; i < BNV; ++i) {
      const int q = tid + i * NT;
      const int k = q / (ROWS / 4), r4 = (q % (ROWS / 4)) * 4;
      const int gk = k0 + k;
      t.v[i] = *reinterpret_cast<const float4*>(p + (int64_t)(gk < kend ? gk : k0) * ld + (r0 + r4));
    }
  }
}

template <int NT, int ROWS, int BKB>
__device__ __forceinline__ void bf_mask(StageRegs<ROWS * BKB / 4 / NT>& t, int kcontig, int r0, int R, int k0, int kend) {
  constexpr int BNV = ROWS * BKB / 4 / NT;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < BNV; ++i) {
    const int q = tid + i * NT;
    if (kcontig) {
      const int r = q / (BKB / 4), k4 = (q % (BKB / 4)) * 4;
      const int gk = k0 + k4;
      const bool rl = r0 + r < R;
      t.v[i] = make_float4(rl && gk + 0 < kend ? t.v[i].x : 0.f, rl && gk + 1 < kend ? t.v[i].y : 0.f, rl && gk + 2 < kend ? t.v[i].z : 0.f,
                           rl && gk + 3 < kend ? t.v[i].w : 0.f);
    } else {
      const int k = q / (ROWS / 4);
      if (k0 + k >= kend) t.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// Operand that already exists as bf16 (dZ written by the transposed aggregation in bf16 compute mode): 8 bytes per slot, no
// conversion.  Same slot mapping and the same issue-now / mask-later split as bf_load_fast; the 4 bf16 travel as raw bits in
// v[i].x / .y.
template <int NT, int ROWS, int BKB>
__device__ __forceinline__ void bf_load16(StageRegs<ROWS * BKB / 4 / NT>& t, const uint16_t* __restrict__ p, int ld, int kcontig, int r0, int R,
                                          int k0, int kend) {
  constexpr int BNV = ROWS * BKB / 4 / NT;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < BNV; ++i) {
    const int q = tid + i * NT;
    uint2 bits;
    if (kcontig) {
      const int r = q / (BKB / 4), k4 = (q % (BKB / 4)) * 4;
      const int gr = r0 + r, gk = k0 + k4;
      bits = *reinterpret_cast<const uint2*>(p + (int64_t)(gr < R ? gr : R - 1) * ld + (gk < kend ? gk : k0));
    } else {
      const int k = q / (ROWS / 4), r4 = (q % (ROWS / 4)) * 4;
      const int gk = k0 + k;
      bits = *reinterpret_cast<const uint2*>(p + (int64_t)(gk < kend ? gk : k0) * ld + (r0 + r4));
    }
    t.v[i].x = __uint_as_float(bits.x);
    t.v[i].y = __uint_as_float(bits.y);
  }
}

template <int NT, int ROWS, int BKB>
__device__ __forceinline__ void bf_mask16(StageRegs<ROWS * BKB / 4 / NT>& t, int kcontig, int r0, int R, int k0, int kend) {
  constexpr int BNV = ROWS * BKB / 4 / NT;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < BNV; ++i) {
    const int q = tid + i * NT;
    uint32_t x = __float_as_uint(t.v[i].x), y = __float_as_uint(t.v[i].y);
    if (kcontig) {
      const int r = q / (BKB / 4), k4 = (q % (BKB / 4)) * 4;
      const int gk = k0 + k4;
      const bool rl = r0 + r < R;
      if (!(rl && gk + 0 < kend)) x &= 0xffff0000u;
      if (!(rl && gk + 1 < kend)) x &= 0x0000ffffu;
      if (!(rl && gk + 2 < kend)) y &= 0xffff0000u;
      if (!(rl && gk + 3 < kend)) y &= 0x0000ffffu;
    } else {
      const int k = q / (ROWS / 4);
      if (k0 + k >= kend) x = y = 0u;
    }
    t.v[i].x = __uint_as_float(x);
    t.v[i].y = __uint_as_float(y);
  }
}

template <int NT, int ROWS, int BKB>
__device__ __forceinline__ void bf_store16(const StageRegs<ROWS * BKB / 4 / NT>& t, __bf16* __restrict__ s, int kcontig) {
  constexpr int BNV = ROWS * BKB / 4 / NT;
  constexpr int BPITCH = BKB + 8;
  constexpr int RP = ROWS + 8;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < BNV; ++i) {
    const int q = tid + i * NT;
    const uint2 bits = make_uint2(__float_as_uint(t.v[i].x), __float_as_uint(t.v[i].y));
    if (kcontig) {
      const int r = q / (BKB / 4), k4 = (q % (BKB / 4)) * 4;
      *reinterpret_cast<uint2*>(s + r * BPITCH + k4) = bits;
    } else {
      const int k = q / (ROWS / 4), r4 = (q % (ROWS / 4)) * 4;
      *reinterpret_cast<uint2*>(s + k * RP + r4) = bits;
    }
  }
}

// edge loader: element by element with bounds (last column tile, ones column, unaligned operands)
template <int NT, int ROWS, int BKB>
__device__ __forceinline__ void bf_load_edge(StageRegs<ROWS * BKB / 4 / NT>& t, const float* __restrict__ p, int ld, int kcontig, int r0, int R,
                                             int n_real, int aug, int k0, int kend) {
  constexpr int BNV = ROWS * BKB / 4 / NT;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < BNV; ++i) {
    const int q = tid + i * NT;
    float e[4] = {0.f, 0.f, 0.f, 0.f};
    if (kcontig) {
      const int r = q / (BKB / 4), k4 = (q % (BKB / 4)) * 4;
      const int gr = r0 + r;
      if (gr < R) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k0 + k4 + j < kend) e[j] = p[(int64_t)gr * ld + k0 + k4 + j];
      }
    } else {  // columns past n_real are zero, the ones column is virtual
      const int k = q / (ROWS / 4), r4 = (q % (ROWS / 4)) * 4;
      const int gk = k0 + k;
      if (gk < kend) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = r0 + r4 + j;
          e[j] = (c < n_real) ? p[(int64_t)gk * ld + c] : ((aug && c == n_real) ? 1.0f : 0.0f);
        }
      }
    }
    t.v[i] = make_float4(e[0], e[1], e[2], e[3]);
  }
}

template <int NT, int ROWS, int BKB>
__device__ __forceinline__ void bf_store(const StageRegs<ROWS * BKB / 4 / NT>& t, __bf16* __restrict__ s, int kcontig) {
  constexpr int BNV = ROWS * BKB / 4 / NT;
  constexpr int BPITCH = BKB + 8;
  constexpr int RP = ROWS + 8;  // bf16 elements per k row of a [k][row] image (8-byte aligned rows)
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < BNV; ++i) {
    const int q = tid + i * NT;
    bf16x4 b;
    b[0] = (__bf16)t.v[i].x; b[1] = (__bf16)t.v[i].y; b[2] = (__bf16)t.v[i].z; b[3] = (__bf16)t.v[i].w;
    if (kcontig) {
      const int r = q / (BKB / 4), k4 = (q % (BKB / 4)) * 4;
      *reinterpret_cast<bf16x4*>(s + r * BPITCH + k4) = b;
    } else {  // natural [k][row] image
      const int k = q / (ROWS / 4), r4 = (q % (ROWS / 4)) * 4;
      *reinterpret_cast<bf16x4*>(s + k * RP + r4) = b;
    }
  }
}

// MFMA operand (8 consecutive k of row `rowbase + lane % 32`, k half lane / 32) of k step ks
//  [row][k] image: one 16-byte read.   [k][row] image: two hardware-transposed reads (ds_read_b64_tr_b16): inside a
//  16-lane group, lane 4q+p supplies the address of k row q, columns 4p..4p+3 and receives column (lane % 16), 4 k rows.
//  EXEC must be all ones here (no divergence in the main loop).
typedef short s16x4 __attribute__((ext_vector_type(4)));
template <int ROWS, int BKB>
__device__ __forceinline__ bf16x8 bf_fetch(const __bf16* __restrict__ s, int kcontig, int rowbase, int ks, int lane) {
  constexpr int BPITCH = BKB + 8;
  constexpr int RP = ROWS + 8;
  if (kcontig) return *reinterpret_cast<const bf16x8*>(s + (rowbase + (lane & 31)) * BPITCH + ks * 16 + 8 * (lane >> 5));
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const int k0 = ks * 16 + 8 * (g >> 1);
  const __bf16* a0 = s + (k0 + q) * RP + rowbase + 16 * (g & 1) + 4 * p;
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a0 + 4 * RP));
  union { s16x4 h[2]; bf16x8 v; } u;
  u.h[0] = lo;
  u.h[1] = hi;
  return u.v;
}

__device__ __forceinline__ float bf_act_mask(float h, int act, bool keep, float scale) {
  if (!keep) return 0.f;
  if (act == HMP_ACT_RELU) return h > 0.f ? scale : 0.f;
  if (act == HMP_ACT_ELU) return h > 0.f ? scale : (h + scale);
  return scale;
}

// NT threads, ROWS x ROWS tile, waves WMW (along M) x WNW (along N), every wave (ROWS/WMW) x (ROWS/WNW) = MI x NI MFMA tiles
// FORM fixes the operand layouts at compile time (0: NT = x * W^T, 1: NN = dZ * W, 2: TN = dZ^T * [x | 1]; 3: per problem at run
// time).  With run-time layouts the branches sit inside the unrolled load / LDS-read loops and every access waits for the one
// before it.
template <bool ONES, int NT, int ROWS, int WMW, int WNW, int BKB, int FORM>
__global__ __launch_bounds__(NT, NT == 256 ? 2 : 1) void gemm_bf16_kernel(const GemmBatch gb) {
  constexpr int MI = ROWS / WMW / 32, NI = ROWS / WNW / 32;
  constexpr int BPITCH = BKB + 8;
  constexpr int LDSN = (ROWS * BPITCH > BKB * (ROWS + 8)) ? ROWS * BPITCH : BKB * (ROWS + 8);
  __shared__ __attribute__((aligned(16))) __bf16 As[LDSN];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[LDSN];
  int pi = 0;
  while (pi + 1 < gb.n && (int)blockIdx.x >= gb.p[pi + 1].tile_start) ++pi;
  const GemmProblem& P = gb.p[pi];
  const int local = blockIdx.x - P.tile_start;
  // XCD-aware order as in gemm.hip: K chunk fastest; row tiles grouped by 8 so that the column tiles of a row tile share an L2
  const int z = local % P.ksplit, t = local / P.ksplit;
  const int grp = t / (8 * P.tiles_n), within = t % (8 * P.tiles_n);
  const int rows_in_grp = min(8, P.tiles_m - grp * 8);
  const int m0 = (grp * 8 + within % rows_in_grp) * ROWS, n0 = (within / rows_in_grp) * ROWS;
  const int kbeg = z * P.kchunk;
  const int kend = min(P.K, kbeg + P.kchunk);
  const int a_kc = FORM == 3 ? (P.trans_a ? 0 : 1) : (FORM == 2 ? 0 : 1);
  const int b_kc = FORM == 3 ? (P.trans_b ? 1 : 0) : (FORM == 0 ? 1 : 0);
  // block-uniform loader choice
  const bool a_al = (P.lda & 3) == 0 && (reinterpret_cast<uintptr_t>(P.A) & 15) == 0;
  const bool b_al = (P.ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(P.B) & 15) == 0;
  const bool a_16 = P.a_bf16 != 0;  // host guarantees: 8-byte aligned rows, whole tiles (see GemmProblem::a_bf16)
  const bool a_fast = !a_16 && a_al && (a_kc || m0 + ROWS <= P.M);
  const bool b_16 = P.b_bf16 != 0;  // host guarantees: n_real is whole tiles, the ones column is the ONES product (GemmProblem::b_bf16)
  const bool b_fast = !b_16 && b_al && (b_kc || n0 + ROWS <= P.n_real);
  // virtual ones column of B (bias gradient = column sums of A over k): instead of a whole extra column tile for ONE column
  // (a third of the config-5 weight-gradient work), the first column tile's wn == 0 waves run one more MFMA per row tile
  // and k step against an all-ones operand; column 0 of that product is the column sum.
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w % WMW, wn = w / WMW;
  const bool ones_here = ONES && P.aug_ones && n0 == 0 && wn == 0;  // wave-uniform

  f32x16 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (P.Cadd && z == 0) {  // block-uniform: the product accumulates ON TOP of the addend (GemmProblem::Cadd)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * (MI * 32) + i * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
          const int col = n0 + wn * (NI * 32) + j * 32 + (lane & 31);
          acc[i][j][r] = (row < P.M && col < P.N) ? P.Cadd[(int64_t)row * P.ldadd + col] : 0.f;
        }
  }

  f32x16 acc1[ONES ? MI : 1];
#pragma unroll
  for (int i = 0; i < (ONES ? MI : 1); ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[i][r] = 0.f;
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;

  StageRegs<ROWS * BKB / 4 / NT> ra, rb;
  // A in two pieces (a_split): [k][row] images (weight gradient) pick the piece per row tile, [row][k] images (input
  // gradient) per K stage; the second base is moved back by a_split elements so that the loaders keep their global indices
  const uint16_t* A16 = reinterpret_cast<const uint16_t*>(P.A);
  const uint16_t* A16b = reinterpret_cast<const uint16_t*>(P.A2) - P.a_split;
  const bool a_two = a_16 && P.a_split > 0;
  const bool a_tile2 = a_two && !a_kc && m0 >= P.a_split;  // block-uniform
  auto load = [&](int k0) {
    if (a_16) {
      const bool second = a_tile2 || (a_two && a_kc && k0 >= P.a_split);
      bf_load16<NT, ROWS, BKB>(ra, second ? A16b : A16, second ? P.lda2 : P.lda, a_kc, m0, P.M, k0, kend);
    }
    else if (a_fast) bf_load_fast<NT, ROWS, BKB>(ra, P.A, P.lda, a_kc, m0, P.M, k0, kend);
    else bf_load_edge<NT, ROWS, BKB>(ra, P.A, P.lda, a_kc, m0, P.M, P.M, 0, k0, kend);
    if (b_16) bf_load16<NT, ROWS, BKB>(rb, reinterpret_cast<const uint16_t*>(P.B), P.ldb, b_kc, n0, P.n_real, k0, kend);
    else if (b_fast) bf_load_fast<NT, ROWS, BKB>(rb, P.B, P.ldb, b_kc, n0, P.n_real, k0, kend);
    else bf_load_edge<NT, ROWS, BKB>(rb, P.B, P.ldb, b_kc, n0, P.n_real, P.n_real, P.aug_ones, k0, kend);
  };
  load(kbeg);
  for (int kt = kbeg; kt < kend; kt += BKB) {
    if (a_16) bf_mask16<NT, ROWS, BKB>(ra, a_kc, m0, P.M, kt, kend);
    else if (a_fast) bf_mask<NT, ROWS, BKB>(ra, a_kc, m0, P.M, kt, kend);
    if (b_16) bf_mask16<NT, ROWS, BKB>(rb, b_kc, n0, P.n_real, kt, kend);
    else if (b_fast) bf_mask<NT, ROWS, BKB>(rb, b_kc, n0, P.n_real, kt, kend);
    if (a_16) bf_store16<NT, ROWS, BKB>(ra, As, a_kc);
    else bf_store<NT, ROWS, BKB>(ra, As, a_kc);
    if (b_16) bf_store16<NT, ROWS, BKB>(rb, Bs, b_kc);
    else bf_store<NT, ROWS, BKB>(rb, Bs, b_kc);
    __syncthreads();
    if (kt + BKB < kend) load(kt + BKB);
#pragma unroll
    for (int ks = 0; ks < BKB / 16; ++ks) {
      bf16x8 av[MI], bv[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) av[i] = bf_fetch<ROWS, BKB>(As, a_kc, wm * (MI * 32) + i * 32, ks, lane);
#pragma unroll
      for (int j = 0; j < NI; ++j) bv[j] = bf_fetch<ROWS, BKB>(Bs, b_kc, wn * (NI * 32) + j * 32, ks, lane);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
      if constexpr (ONES) {
        if (ones_here) {
#pragma unroll
          for (int i = 0; i < MI; ++i) acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], ones, acc1[i], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // D layout of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  // Epilogue without branches between memory operations: the problem's fields live in registers, the 16 activation values of
  // a 32x32 tile are requested together (clamped addresses), stores are predicated.  (Written element by element with
  // `continue`s, every H load was waited for before the next one was issued: 128 dependent round trips per thread.)
  float* C = P.C + (int64_t)z * P.slab_stride;
  const int Mrows = P.M, Ncols = P.N, ldc = P.ldc, ldh = P.ldh, act = P.act;
  const bool amask = P.epi == EPI_ACTMASK;
  const bool dropon = P.drop_on != 0;
  const bool c16 = P.c_bf16 != 0;
  const float dscale = dropon ? P.drop.scale : 1.f;
  const float* Hp = P.H;
  const bool h16 = P.h_bf16 != 0;  // block-uniform: the activations were stored as bf16 (sign of zero = keep bit survives)
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = n0 + wn * (NI * 32) + j * 32 + (lane & 31);
      const int rbase = m0 + wm * (MI * 32) + i * 32 + 4 * (lane >> 5);
      const bool cok = col < Ncols;
      const int colc = cok ? col : 0;
      float hv[16];
      if (amask && h16) {  // hoisted like c16 below: no branch between the 16 loads of either form
        const uint16_t* Hb = reinterpret_cast<const uint16_t*>(Hp);
        uint16_t hb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          hb[r] = Hb[(int64_t)(row < Mrows ? row : Mrows - 1) * ldh + colc];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[r] = __uint_as_float((uint32_t)hb[r] << 16);
      } else if (amask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          hv[r] = Hp[(int64_t)(row < Mrows ? row : Mrows - 1) * ldh + colc];
        }
      }
      float vv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r];
        if (amask) {
          const bool keep = !dropon || (__float_as_uint(hv[r]) != 0x80000000u);  // dropped elements were stored as -0.0f
          v *= bf_act_mask(hv[r], act, keep, dscale);
        }
        vv[r] = v;
      }
      if (c16) {  // block-uniform: bf16 projected rows (ldc counts elements)
        __bf16* C16 = reinterpret_cast<__bf16*>(C);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (cok && row < Mrows) C16[(int64_t)row * ldc + col] = (__bf16)vv[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          if (cok && row < Mrows) C[(int64_t)row * ldc + col] = vv[r];
        }
      }
    }
  if (ONES && ones_here && (lane & 31) == 0) {  // column 0 of the ones product -> C[:, n_real]
#pragma unroll
    for (int i = 0; i < (ONES ? MI : 1); ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (MI * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < P.M) C[(int64_t)row * P.ldc + P.n_real] = acc1[i][r];
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Weight-stationary form of the NT product C[M, N] = A[M, K] * W[N, K]^T for tall problems (the projections of a 10^6-node
// graph: M = 10^6, N = 768, K = 256) whose A is already stored as bf16 -- round 2.
//
// The tiled kernel above re-stages BOTH operands through LDS for every 256 x 256 output tile: 26 us per tile against an MFMA
// floor of 3.4 us, with the matrix pipe 10-20 % busy and 29-64 % of the LDS cycles lost to bank conflicts
// (profiles/r02_b_pmc_cfg5_mfma_lds.json).  Here a workgroup (4 waves, two workgroups per CU) owns 256 output columns for its whole life:
//   * its slice of W (256 x K, rounded to bf16) lives in REGISTERS as ready-made MFMA operands -- 128 VGPRs per wave, loaded
//     once, never staged again;
//   * A streams through two 32 KB LDS buffers, 64 rows at a time, with global_load_lds_dwordx4: memory -> LDS directly, no
//     VGPR staging, no ds_write.  The 16-byte chunks of a row are stored XOR-swizzled (position = chunk ^ (row & 31) -- each
//     lane simply REQUESTS the chunk that belongs at its position), so the operand fetch of 32 rows x one chunk is a
//     conflict-free ds_read_b128 without padding;
//   * the product is computed transposed, D = W_tile * A_tile^T (both MFMA operands have the same per-lane layout, so that
//     costs nothing): a lane then holds 4 CONSECUTIVE output columns of one output row per register quad -- 8-byte bf16
//     stores instead of 2-byte ones.
// The column slices of one row range sit on the same XCD (block ids 8 apart), so A comes out of HBM once.
// Per 64-row tile and workgroup: 32 KB of A in, 32 KB of C out, 64 MFMAs per wave.  Two workgroups per CU drift apart in phase, so
// one's MFMAs overlap the other's loads and stores (one 8-wave workgroup per CU ran its phases in lockstep: 656 us for the 10^6 x
// 768 x 256 projection against the 288 us of its MFMA phase alone).
constexpr int WS_ROWS = 64, WS_COLS = 256, WS_KMAX = 256, WS_THREADS = 256, WS_PER_CU = 2;
struct WsArgs {
  const void* A;      // bf16 or fp32 [M][lda] (lda in elements)
  const float* W;     // fp32 [N][ldb]
  void* C;            // bf16 or fp32 [M][ldc]
  int M, N, K, lda, ldb, ldc, c_bf16;
  int n_slices, groups, tiles_per_group, n_tiles;
  int dbg;  // HMP_WS_DBG (measurements only): 1 = no stores, 2 = no loads after the first tile, 4 = no MFMAs
};

// KS: K / 16 fixed at compile time (16: the K = 256 of hidden-256 layers -- no branch between the k steps, so the operand reads of
// step s + 1 are scheduled ahead of the MFMAs of step s), 0: any K <= 256 at run time
// AF32: A is stored as fp32 (the input features of layer 0, or every layer when the activations stay fp32).  Its rows go through
// the same 512-byte LDS rows in PARTS of 128 floats (a K = 256 tile is two pipeline units sharing one accumulator set); an
// operand is two 16-byte reads rounded to bf16 on the way into the MFMA -- what the tiled kernel does on the way into LDS.
template <int KS, bool AF32>
__global__ __launch_bounds__(WS_THREADS, WS_PER_CU) void gemm_bf16_ws_kernel(const WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ws_lds[];  // [2][WS_ROWS][512 bytes]
  const int b = blockIdx.x;
  // blocks b, b + 8, b + 16 .. share an XCD: they take the column slices of the same row range
  const int slice = (b >> 3) % a.n_slices;
  const int group = (b & 7) + 8 * (b / (8 * a.n_slices));
  const int n0 = slice * WS_COLS;
  const int t_begin = group * a.tiles_per_group;
  const int t_end = min(t_begin + a.tiles_per_group, a.n_tiles);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wm = 0, wn = w;  // 4 waves: all 64 rows of the tile x 64 columns each
  const int K = KS ? KS * 16 : a.K, ksteps = K >> 4;
  const int l31 = lane & 31, half = lane >> 5;

  // ---- this wave's 64 columns of W as MFMA operands: wreg[j][s] = W[n0 + wn*64 + j*32 + l31][16 s + 8 half .. + 8]
  bf16x8 wreg[2][WS_KMAX / 16];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn * 64 + j * 32 + l31;
    const float* wrow = a.W + (int64_t)min(col, a.N - 1) * a.ldb;
#pragma unroll
    for (int s = 0; s < WS_KMAX / 16; ++s) {
      const int k = min(16 * s + 8 * half, K - 8);  // steps past K are never multiplied
      const float4 lo = *reinterpret_cast<const float4*>(wrow + k);
      const float4 hi = *reinterpret_cast<const float4*>(wrow + k + 4);
      const bool live = col < a.N;
      bf16x8 v;
      v[0] = (__bf16)(live ? lo.x : 0.f); v[1] = (__bf16)(live ? lo.y : 0.f); v[2] = (__bf16)(live ? lo.z : 0.f); v[3] = (__bf16)(live ? lo.w : 0.f);
      v[4] = (__bf16)(live ? hi.x : 0.f); v[5] = (__bf16)(live ? hi.y : 0.f); v[6] = (__bf16)(live ? hi.z : 0.f); v[7] = (__bf16)(live ? hi.w : 0.f);
      wreg[j][s] = v;
    }
  }
  if (t_begin >= t_end) return;  // block-uniform

  // ---- A tile -> LDS: wave w requests rows 16 w .. 16 w + 15 of the tile, two rows (32 chunks each) per instruction
  constexpr int ESZ = AF32 ? 4 : 2;                  // bytes per element of A
  constexpr int CE = 16 / ESZ;                       // elements per 16-byte chunk
  constexpr int PMAX = AF32 ? 2 : 1;                 // pipeline units (parts of 32 chunks) per tile
  const int P = AF32 ? (KS ? KS / 8 : (K + 127) >> 7) : 1;
  const int kchunks = K / CE;                        // 16-byte chunks per row that exist
  const unsigned char* Abytes = reinterpret_cast<const unsigned char*>(a.A);
  auto request = [&](int t, int h, int buf) {
    const int m0 = t * WS_ROWS;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = (w * 8 + i) * 2 + half;          // row inside the tile (wave w: rows 16 w .. 16 w + 15)
      const int c = 32 * h + (l31 ^ (r & 31));        // the chunk that belongs at position l31 of this row
      const int64_t grow = min(m0 + r, a.M - 1);      // rows past M: a valid address, the products are never stored
      const unsigned char* src = Abytes + (grow * a.lda + (int64_t)CE * min(c, kchunks - 1)) * ESZ;
      unsigned char* dst = ws_lds + buf * (WS_ROWS * 512) + (w * 8 + i) * 1024;  // wave-uniform; lane l lands at + 16 l
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  // Order per unit: request the next unit | MFMAs of this one | wait (the next unit's requests, the previous tile's stores: both a
  // whole MFMA phase old) | barrier | stores (last part of a tile).  The wait is for EVERYTHING outstanding (on gfx9 loads and
  // stores share one counter and stores may complete out of order with loads, so a count that skips the newest stores would not
  // prove the loads landed) -- placed where everything outstanding is old.
  request(t_begin, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x16 acc[2][2];
  for (int t = t_begin; t < t_end; ++t) {
    int buf = 0;
#pragma unroll
    for (int h = 0; h < PMAX; ++h) {
      if (h >= P) break;  // block-uniform
      buf = ((t - t_begin) * P + h) & 1;
      // the other buffer: every wave left its MFMAs of the previous unit before the last barrier
      if (!(a.dbg & 2)) {
        if (h + 1 < P) request(t, h + 1, buf ^ 1);
        else if (t + 1 < t_end) request(t + 1, 0, buf ^ 1);
      }
      if (h == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
      }
      const unsigned char* Ab = ws_lds + buf * (WS_ROWS * 512);
      auto kstep = [&](int s) {  // s: k step of the tile (compile-time); the part holds steps 8 h .. 8 h + 7 when AF32
        if constexpr (AF32) {
          const int sp = s - 8 * h;
#pragma unroll
          for (int i = 0; i < 2; ++i) {  // one row tile at a time: 8 floats + their bf16 image are all the registers this needs
            const int row = i * 32 + l31;
            const float4 f0 = *reinterpret_cast<const float4*>(Ab + row * 512 + (((4 * sp + 2 * half) ^ l31) << 4));
            const float4 f1 = *reinterpret_cast<const float4*>(Ab + row * 512 + (((4 * sp + 2 * half + 1) ^ l31) << 4));
            bf16x8 av;
            av[0] = (__bf16)f0.x; av[1] = (__bf16)f0.y; av[2] = (__bf16)f0.z; av[3] = (__bf16)f0.w;
            av[4] = (__bf16)f1.x; av[5] = (__bf16)f1.y; av[6] = (__bf16)f1.z; av[7] = (__bf16)f1.w;
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[j][s], av, acc[j][i], 0, 0, 0);
          }
        } else {
          bf16x8 av[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int row = wm * 64 + i * 32 + l31;
            av[i] = *reinterpret_cast<const bf16x8*>(Ab + row * 512 + (((2 * s + half) ^ l31) << 4));
          }
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[j][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[j][s], av[i], acc[j][i], 0, 0, 0);
        }
      };
      if (!(a.dbg & 4)) {
        constexpr int SPP = AF32 ? 8 : WS_KMAX / 16;  // k steps per part
        if constexpr (KS > 0) {
#pragma unroll
          for (int s = h * SPP; s < (h + 1) * SPP && s < KS; ++s) kstep(s);
        } else {
#pragma unroll
          for (int s = h * SPP; s < (h + 1) * SPP; ++s)
            if (s < ksteps) kstep(s);  // block-uniform
        }
      }
      if (h + 1 < P) {  // more parts of this tile: hand the buffers over and go on accumulating
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- D[j][i]: lane = output row m0 + wm*64 + i*32 + l31; registers 4 g .. 4 g + 3 = output columns n0 + wn*64 + j*32 + 4 half + 8 g ..
    const int m0 = t * WS_ROWS;
    if (a.c_bf16 && !(a.dbg & 1)) {  // block-uniform
      // bf16 output: a lane's 8-byte pieces (4 columns of one row) written straight to memory touch 32 rows per instruction, 16
      // bytes each -- measured 3.2 TB/s for the 1.5 GB of a config-5 projection, the kernel's bound.  The wave turns its 64 x 64
      // block into row-major order through LDS instead and writes whole 128-byte lines, 8 rows per instruction.  Its staging
      // area is the 8 KB of the consumed A buffer that only this wave's next requests overwrite (rows 16 w .. of the tile), so
      // wave-local ordering is all the synchronisation this needs.  8-byte units are XOR-swizzled by the row.
      unsigned char* stg = ws_lds + buf * (WS_ROWS * 512) + w * 8192;  // [64 rows][128 bytes]
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int r = i * 32 + l31;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int u = j * 8 + half + 2 * g;  // 8-byte unit of the row: columns 4 u .. 4 u + 3
            bf16x4 o;
            o[0] = (__bf16)acc[j][i][4 * g + 0]; o[1] = (__bf16)acc[j][i][4 * g + 1];
            o[2] = (__bf16)acc[j][i][4 * g + 2]; o[3] = (__bf16)acc[j][i][4 * g + 3];
            *reinterpret_cast<bf16x4*>(stg + r * 128 + ((u ^ (r & 15)) << 3)) = o;
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's own writes
      const int q = lane & 7;  // 16-byte piece of the row: columns 8 q .. 8 q + 7
#pragma unroll 2
      for (int k = 0; k < 8; ++k) {
        const int r = k * 8 + (lane >> 3);
        const uint2 lo = *reinterpret_cast<const uint2*>(stg + r * 128 + (((2 * q) ^ (r & 15)) << 3));
        const uint2 hi = *reinterpret_cast<const uint2*>(stg + r * 128 + (((2 * q + 1) ^ (r & 15)) << 3));
        const int row = m0 + wm * 64 + r;
        const int col = n0 + wn * 64 + 8 * q;
        __bf16* dst = reinterpret_cast<__bf16*>(a.C) + (int64_t)row * a.ldc + col;
        if (row < a.M && col + 8 <= a.N) {
          *reinterpret_cast<uint4*>(dst) = make_uint4(lo.x, lo.y, hi.x, hi.y);
        } else if (row < a.M && col < a.N) {  // N is a multiple of 4: the first half of the piece exists
          *reinterpret_cast<uint2*>(dst) = lo;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // staged reads done before this wave's next requests reuse the area
    } else if (!(a.dbg & 1)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = m0 + wm * 64 + i * 32 + l31;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = n0 + wn * 64 + j * 32 + 4 * half + 8 * g;
          if (row < a.M && col < a.N)
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.C) + (int64_t)row * a.ldc + col) =
                make_float4(acc[j][i][4 * g + 0], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]);
        }
    }
    }
  }
}

// problems of a launch the weight-stationary kernel takes (one launch each); false: the tiled kernel
static bool ws_takes(const GemmProblem& p, bool want_split) {
  const char* v = getenv("HMP_GEMM_WS");  // 0: the tiled kernel for every problem (tests compare the two)
  const bool on = !(v && v[0] == '0');
  // an fp32 A only at K = 256 (the run-time-K form of that variant does not fit the register file: 75 spilled registers)
  if (!p.a_bf16 && p.K != 256) return false;
  return on && !want_split && !p.trans_a && p.trans_b && p.a_split == 0 && !p.b_bf16 && p.epi == EPI_NONE && !p.aug_ones &&
         p.K >= 16 && p.K <= WS_KMAX && (p.K & 15) == 0 && (p.lda & (p.a_bf16 ? 7 : 3)) == 0 && (p.ldb & 3) == 0 && (p.ldc & 3) == 0 && (p.N & 3) == 0 &&
         (reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.B) & 15) == 0 &&
         (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 && p.M >= 32768 && p.N >= 64;
}
static int ws_launch(const GemmProblem& p, hipStream_t st) {
  WsArgs a;
  a.A = p.A; a.W = p.B; a.C = p.C;
  a.M = p.M; a.N = p.N; a.K = p.K; a.lda = p.lda; a.ldb = p.ldb; a.ldc = p.ldc; a.c_bf16 = p.c_bf16;
  a.n_slices = cdiv(p.N, WS_COLS);
  a.n_tiles = cdiv(p.M, WS_ROWS);
  a.dbg = 0;
#ifdef HMP_KTIME  // measurement-only switch (wrong results): profiling build only
  {
    const char* dv = getenv("HMP_WS_DBG");
    a.dbg = dv ? atoi(dv) : 0;
  }
#endif
  static const int n_cu = [] {  // (queried once: the property call is not cheap)
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) return prop.multiProcessorCount;
    return 256;
  }();
  // groups in sets of 8 (one per XCD); as many sets as fit the chip, at least one
  int sets = WS_PER_CU * n_cu / (8 * a.n_slices);
  if (sets < 1) sets = 1;
  a.groups = 8 * sets;
  if (a.groups > a.n_tiles) a.groups = ((a.n_tiles + 7) / 8) * 8;
  a.tiles_per_group = cdiv(a.n_tiles, a.groups);
  const dim3 grid(a.groups * a.n_slices), block(WS_THREADS);
  const size_t lds = 2 * WS_ROWS * 512;  // 64 KB: the default dynamic limit
  if (p.a_bf16) {
    if (p.K == 256) hipLaunchKernelGGL((gemm_bf16_ws_kernel<16, false>), grid, block, lds, st, a);
    else hipLaunchKernelGGL((gemm_bf16_ws_kernel<0, false>), grid, block, lds, st, a);
  } else {
    hipLaunchKernelGGL((gemm_bf16_ws_kernel<16, true>), grid, block, lds, st, a);  // (ws_takes: K == 256 only for an fp32 A)
  }
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

// Weight gradients (TN, split-K over node chunks) of a large batch also take the 256x256 tile: the 128x128 tile re-reads dZ
// once per 128 input columns and H once per 128 packed rows (12 GB of operand traffic for the 768 x 257 x 10^6 problem of
// config 5, 6.5 ms); the big tile halves that, provided the launch still fills the chip: 1 block per CU, so the split aims at
// 2 x 256 blocks and needs up to ~170 slabs for 3 tiles (an earlier attempt capped at 64 slabs left a quarter of the CUs
// idle and measured slower).
// Measured on top of this kernel and rejected (config 5, 28.7 ms/step): a bf16 image of the stacked weights as B operand (half
// the bytes, the same number of requests per stage: forward 5.14 -> 5.19 ms, backward 8.25 -> 8.99 ms) and two stages of register
// prefetch instead of one (256 VGPRs + scratch spills: forward 5.50 ms, backward 13.5 ms).  The cause was elsewhere (the ISA: every load
// of a stage was waited for before the next, see bf_load_fast / the epilogue).  With that fixed the kernel moves ~7.5 TB/s through
// L2; two stages of prefetch were measured again on the clean loader (no spills for the plain products) and are still slower
// (forward 2.72 -> 2.87 ms, backward 3.75 -> 4.04 ms), and so is the bf16 weight image (2.75 / 4.03 ms): neither bytes nor
// requests explain the ~26 us per tile that remain (MFMA floor 3.4 us); LDS traffic / barriers are the untested suspects.  AGPRs are no extra budget here: 8
// waves per CU = 2 per SIMD = 256 registers per wave, VGPRs and AGPRs together.  Next candidate: a B-stationary persistent
// form (the 256 x 256 bf16 weight tile fits LDS).
int gemm_bf16_launch(GemmBatch& gb, bool want_split, int max_slabs, hipStream_t st) {
  HMP_CHECK_ARG(gb.n >= 0 && gb.n <= GEMM_MAX_PROB, "gemm_bf16: %d problems", gb.n);
  {  // backward products of the 10^6-node regime: operand-stationary kernels (gemm_bf16_bwd.hip), one launch per problem
    GemmBatch rest;
    memset(&rest, 0, sizeof(rest));
    int where[GEMM_MAX_PROB];
    int taken = 0;
    for (int i = 0; i < gb.n; ++i) {
      if (gemm_bf16_dx_takes(gb.p[i], want_split)) {
        HMP_TRY(gemm_bf16_dx_launch(gb.p[i], st));
        gb.p[i].ksplit = 1;
        ++taken;
      } else if (gemm_bf16_dw_takes(gb.p[i], want_split)) {
        int ns = 1;
        HMP_TRY(gemm_bf16_dw_launch(gb.p[i], max_slabs, &ns, st));
        gb.p[i].ksplit = ns;
        ++taken;
      } else {
        where[rest.n] = i;
        rest.p[rest.n++] = gb.p[i];
      }
    }
    if (taken) {
      if (rest.n == 0) return HMP_OK;
      HMP_TRY(gemm_bf16_launch(rest, want_split, max_slabs, st));  // (no problem of `rest` is taken here again)
      for (int i = 0; i < rest.n; ++i) gb.p[where[i]].ksplit = rest.p[i].ksplit;
      return HMP_OK;
    }
  }
  {  // tall NT products over a bf16 A: the weight-stationary kernel, one launch per problem; the rest stays with the tiled kernel
    GemmBatch rest;
    memset(&rest, 0, sizeof(rest));
    int taken = 0;
    for (int i = 0; i < gb.n; ++i) {
      if (ws_takes(gb.p[i], want_split)) {
        HMP_TRY(ws_launch(gb.p[i], st));
        ++taken;
      } else {
        rest.p[rest.n++] = gb.p[i];
      }
    }
    if (taken) {
      for (int i = 0; i < gb.n; ++i) gb.p[i].ksplit = 1;  // (no split-K in this branch: want_split is false)
      if (rest.n == 0) return HMP_OK;
      return gemm_bf16_launch(rest, want_split, max_slabs, st);  // (no problem of `rest` is taken again)
    }
  }
  // 256x256 tiles when every problem is a product with at least 4096 x 192 outputs (plain) / 192 x 192 outputs over >= 2^17
  // nodes in one problem (split-K weight gradients; narrow companions such as the 28-row last layer ride along on one tile)
  bool big = true, tn_wide = false;
  bool any_ones = false;
  for (int i = 0; i < gb.n; ++i) {
    const GemmProblem& p = gb.p[i];
    HMP_CHECK_ARG(p.M >= 0 && p.N >= 0 && p.K >= 0, "gemm_bf16: negative size");
    any_ones = any_ones || p.aug_ones != 0;
    HMP_CHECK_ARG(p.a_split == 0 || (p.a_bf16 && p.A2 && (p.a_split & 255) == 0 && (p.lda2 & 3) == 0),
                  "gemm_bf16: a two-piece A needs bf16 pieces and a split at a multiple of 256 (got %d)", p.a_split);
    HMP_CHECK_ARG(!p.b_bf16 || ((p.n_real & 255) == 0 && (p.ldb & 3) == 0 && (!p.aug_ones || want_split)),
                  "gemm_bf16: a bf16 B operand needs n_real %% 256 == 0 (got %d) and the split-K form for its ones column", p.n_real);
    if (want_split) {
      if (!(p.trans_a && !p.trans_b)) big = false;
      if (p.M >= 192 && p.n_real >= 192 && p.K >= (1 << 17)) tn_wide = true;
    } else if (p.aug_ones || p.M < 4096 || p.N < 192) big = false;
  }
  if (want_split && !tn_wide) big = false;
  if (big && !want_split) {  // the 256 x 256 tile runs one workgroup per CU: a launch with fewer tiles than CUs (the 10^4-room products of
                             // config 5: 40 x 3 tiles) leaves half the chip idle -- 128 x 128 tiles (two per CU) then
    int64_t t256 = 0;
    for (int i = 0; i < gb.n; ++i) t256 += (int64_t)cdiv(gb.p[i].M, 256) * cdiv(gb.p[i].N, 256);
    if (t256 < 256) big = false;
  }
  const int BT = big ? 256 : 128;
  const int BKB = big ? 32 : 64;
  int start = 0, all_tiles = 0;
  double work_total = 0.0;  // tiles x nodes
  for (int i = 0; i < gb.n; ++i) {
    const int t = cdiv(gb.p[i].M, BT) * cdiv(gb.p[i].aug_ones ? (gb.p[i].n_real > 0 ? gb.p[i].n_real : 1) : gb.p[i].N, BT);
    all_tiles += t;
    work_total += (double)t * gb.p[i].K;
  }
  for (int i = 0; i < gb.n; ++i) {
    GemmProblem& p = gb.p[i];
    p.tiles_m = cdiv(p.M, BT);
    p.tiles_n = cdiv(p.aug_ones ? (p.n_real > 0 ? p.n_real : 1) : p.N, BT);  // the ones column rides in the first column tile
    const int tiles = p.tiles_m * p.tiles_n;
    int ks = 1;
    if (want_split && tiles > 0) {
      // aim at 4 workgroups per CU; big tile (1 block per CU): 2 x 256 blocks shared out in proportion to the nodes each
      // problem reduces over (the 10^4-room problems of config 5 take one block per tile, the 10^6-object ones the rest)
      if (big) ks = (int)(512.0 * (double)p.K / (work_total > 0 ? work_total : 1.0));
      else ks = 1024 / (all_tiles > 0 ? all_tiles : 1);
      const int max_by_k = cdiv(p.K, BKB);
      if (ks > max_by_k) ks = max_by_k;
      if (ks > (big ? max_slabs : 64)) ks = big ? max_slabs : 64;
      if (ks < 1) ks = 1;
    }
    int kchunk = cdiv(cdiv(p.K, ks), BKB) * BKB;
    if (kchunk < BKB) kchunk = BKB;
    ks = p.K > 0 ? cdiv(p.K, kchunk) : 1;
    p.ksplit = ks;
    p.kchunk = kchunk;
    p.tile_start = start;
    start += tiles * ks;
  }
  gb.total_tiles = start;
  if (start == 0) return HMP_OK;
  // operand form shared by every problem of the launch (the executor's launches are uniform), else the run-time variant
  int form = -1;
  for (int i = 0; i < gb.n; ++i) {
    const GemmProblem& p = gb.p[i];
    const int f = (!p.trans_a && p.trans_b) ? 0 : (!p.trans_a && !p.trans_b) ? 1 : (p.trans_a && !p.trans_b) ? 2 : 3;
    form = (form == -1 || form == f) ? f : 3;
  }
#define BF_LAUNCH(ONES_, NT_, ROWS_, WM_, WN_, BK_, FORM_) \
  hipLaunchKernelGGL((gemm_bf16_kernel<ONES_, NT_, ROWS_, WM_, WN_, BK_, FORM_>), dim3(start), dim3(NT_), 0, st, gb)
#define BF_FORMS(ONES_, NT_, ROWS_, WM_, WN_, BK_)          \
  switch (form) {                                           \
    case 0: BF_LAUNCH(ONES_, NT_, ROWS_, WM_, WN_, BK_, 0); break; \
    case 1: BF_LAUNCH(ONES_, NT_, ROWS_, WM_, WN_, BK_, 1); break; \
    case 2: BF_LAUNCH(ONES_, NT_, ROWS_, WM_, WN_, BK_, 2); break; \
    default: BF_LAUNCH(ONES_, NT_, ROWS_, WM_, WN_, BK_, 3); break; \
  }
  if (big && any_ones) { BF_FORMS(true, 512, 256, 4, 2, 32) }
  else if (big) { BF_FORMS(false, 512, 256, 4, 2, 32) }
  else if (any_ones) { BF_FORMS(true, 256, 128, 2, 2, 64) }
  else { BF_FORMS(false, 256, 128, 2, 2, 64) }
#undef BF_FORMS
#undef BF_LAUNCH
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

}  // namespace hmp

extern "C" int hmp_gemm_bf16(const float* d_a, int32_t lda, int32_t trans_a, const float* d_b, int32_t ldb, int32_t trans_b,
                             float* d_c, int32_t ldc, int32_t M, int32_t N, int32_t K, void* stream) {
  using namespace hmp;
  HMP_CHECK_ARG(d_a && d_b && d_c, "hmp_gemm_bf16: null pointer");
  HMP_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "hmp_gemm_bf16: negative size");
  HMP_CHECK_ARG(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N, "hmp_gemm_bf16: leading dimension too small");
  GemmBatch gb;
  memset(&gb, 0, sizeof(gb));
  gb.n = 1;
  GemmProblem& p = gb.p[0];
  p.A = d_a; p.B = d_b; p.C = d_c;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.trans_a = trans_a; p.trans_b = trans_b;
  p.n_real = N;
  p.epi = EPI_NONE;
  if (M == 0 || N == 0) return HMP_OK;
  return gemm_bf16_launch(gb, false, 1, (hipStream_t)stream);
}

extern "C" int hmp_gemm_bf16_a16(const uint16_t* d_a, int32_t lda, const float* d_w, int32_t ldw, void* d_c, int32_t ldc, int32_t c_bf16,
                                 int32_t M, int32_t N, int32_t K, void* stream) {
  using namespace hmp;
  HMP_CHECK_ARG(d_a && d_w && d_c, "hmp_gemm_bf16_a16: null pointer");
  HMP_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && lda >= K && ldw >= K && ldc >= N, "hmp_gemm_bf16_a16: bad shape");
  HMP_CHECK_ARG((lda & 3) == 0 && (reinterpret_cast<uintptr_t>(d_a) & 7) == 0, "hmp_gemm_bf16_a16: rows of A must be 8-byte aligned");
  GemmBatch gb;
  memset(&gb, 0, sizeof(gb));
  gb.n = 1;
  GemmProblem& p = gb.p[0];
  p.A = reinterpret_cast<const float*>(d_a); p.B = d_w; p.C = reinterpret_cast<float*>(d_c);
  p.a_bf16 = 1; p.c_bf16 = c_bf16 ? 1 : 0;
  p.M = M; p.N = N; p.K = K;
  p.lda = lda; p.ldb = ldw; p.ldc = ldc;
  p.trans_a = 0; p.trans_b = 1;
  p.n_real = N;
  p.epi = EPI_NONE;
  if (M == 0 || N == 0) return HMP_OK;
  return gemm_bf16_launch(gb, false, 1, (hipStream_t)stream);
}
