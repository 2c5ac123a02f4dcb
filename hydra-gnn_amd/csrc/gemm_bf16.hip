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
