// fp32 products on the bf16 matrix pipe: every fp32 operand element is split EXACTLY into three bf16 pieces,
//     x = x1 + x2 + x3,   x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)        (8 + 8 + 8 mantissa bits, each difference exact in fp32)
// and a product a . b is evaluated as the six piece products whose weight is above 2^-24 of it,
//     a1 b1 + (a1 b2 + a2 b1) + (a2 b2 + a1 b3 + a3 b1),
// each exact in the MFMA's fp32 accumulator (8 x 8-bit mantissas).  The dropped terms (a2 b3, a3 b2, a3 b3) are below 2^-25 |a b|:
// the result has fp32 accuracy -- the same 1e-5 parity bar against the float64 oracle as the fp32-MFMA kernels of gemm.hip -- at
// 6 x v_mfma_f32_32x32x16_bf16 (6 x 32 cycles per 16 k) where v_mfma_f32_32x32x2_f32 needs 8 x 64 cycles: 2.7x the peak of the fp32 matrix
// instructions (157 TFLOP/s -> 419 TFLOP/s equivalent).  Non-finite inputs do not survive the split (inf - inf); the engine's
// operands are finite.
//
// Why (round 3): at the reference's own batch size (2048 MP3D graphs: 190 k x 306 x 192 projections) and in the GAT configuration
// (5 490 x 306 x 1 536) the fp32-MFMA kernels ran at 50-85 TFLOP/s, i.e. at 35-55 % of the fp32 matrix peak, and wider tiles did not
// move them (64 x 192 tiles: 0.395 against 0.398 ms): the instruction itself was the bound.
//
// Shape: the tiled kernel of gemm_bf16.hip with three LDS planes per operand.  Block = 4 waves, tile 128 x 128, K stage 32; a wave owns
// a 64 x 64 quadrant = 2 x 2 MFMA tiles; per 16 k it fetches 3 x (2 + 2) operand fragments (ds_read_b128, or two ds_read_b64_tr_b16
// from a natural [k][row] image) and issues 24 MFMAs.  LDS: 6 planes x 10 KB = 60 KB, two workgroups per CU.  Forms and contract as
// gemm.hip: NT (x W^T), NN (dZ W, optional activation-derivative epilogue, optional addend), TN split-K over node chunks with the
// virtual ones column (bias gradient) as one more product per row tile against an all-ones fragment.  Rows that are only 8-byte
// aligned (MP3D object features: pitch 306) load as float2 pairs.
#include "kernels.h"

namespace hmp {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int X3_BK = 32;  // K stage
// Tile shape: <256 threads, 128 x 128, 2 x 2 waves of 64 x 64>, two workgroups per CU, 60 KB of LDS each.  These products are bound
// by the operand traffic between L2 and the CUs (measured ~4.5 TB/s over the chip: the tile loads 32 KB per K stage for 48 MFMAs per
// wave -- the matrix pipe waits).  A <512 threads, 256 x 256, 4 x 2 waves of 64 x 128> shape (twice the bytes for four times the
// work) was written against the same template and does not fit the register file: 128 accumulator registers + three pieces of
// every fragment + the staging registers spill 500-1000 registers at the 256 a wave gets with 8 waves per CU -- not built.
// one operand's tile: ROWS rows x X3_BK k, staged by NT threads
template <int NT_, int ROWS_>
struct X3Side {
  static constexpr int NT = NT_, ROWS = ROWS_;
  static constexpr int NV = ROWS * X3_BK / 4 / NT;  // float4 slots per thread
  static constexpr int PITCH = X3_BK + 8;  // [row][k] image: 40 bf16 = 80 bytes per row (odd multiple of 16 bytes: 8 consecutive rows cover all banks)
  static constexpr int RP = ROWS + 8;      // [k][row] image: bf16 elements per k row
  static constexpr int PLANE = (ROWS * PITCH > X3_BK * RP) ? ROWS * PITCH : X3_BK * RP;
};
// RM x RN output tile, WMW x WNW waves; PER_CU workgroups per CU.  <256, 128, 128, 2, 2, 2>: 60 KB of LDS, 64 accumulator registers.
// <256, 128, 64, 2, 2, 3>: 45 KB, 32 accumulator registers, three workgroups per CU -- for launches whose 128 x 128 tiles would fill the
// 512 workgroup slots badly (the GAT projections: 516 tiles = one full round + 4 tiles; the GAT input gradients: 172 tiles on 256 CUs)
template <int NT_, int RM_, int RN_, int WMW_, int WNW_, int PER_CU_>
struct X3Cfg {
  static constexpr int NT = NT_, RM = RM_, RN = RN_, WMW = WMW_, WNW = WNW_, PER_CU = PER_CU_;
  using SA = X3Side<NT_, RM_>;
  using SB = X3Side<NT_, RN_>;
  static constexpr int MI = RM / WMW / 32, NI = RN / WNW / 32;  // 32x32 MFMA tiles per wave
  static constexpr int LDS_BYTES = 3 * (SA::PLANE + SB::PLANE) * 2;
};
using X3Small = X3Cfg<256, 128, 128, 2, 2, 2>;
using X3Narrow = X3Cfg<256, 128, 64, 2, 2, 3>;

template <int NV>
struct Regs {
  float4 v[NV];
};

// A (128 x 32) stage of one operand in registers: slot q of a k-contiguous operand ([row][k] in memory) covers row q / 8, k = 4 (q % 8)
// .. + 3; of a row-contiguous one ([k][row]) k = q / 32, rows 4 (q % 32) .. + 3.  VEC: 4 = one 16-byte load, 2 = two 8-byte loads (rows
// only 8-byte aligned: MP3D features, pitch 306), 1 = four scalar loads (odd pitches; partial tiles of a row-contiguous operand).
// The loader only ISSUES loads, all of them at clamped in-range addresses and without a branch in between; x3_mask zeroes what lies
// outside the operand when the stage is consumed, one iteration later (a select right here would make the compiler wait for each load
// before the MFMAs it is meant to overlap with: gemm.hip).
template <class C, int VEC>
__device__ __forceinline__ void x3_load(Regs<C::NV>& t, const float* __restrict__ p, int ld, int kcontig, int r0, int R, int k0, int kend) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < C::NV; ++i) {
    const int q = tid + i * C::NT;
    const float* src;
    int o1 = 1, o2 = 2, o3 = 3;
    if (kcontig) {
      const int r = q / (X3_BK / 4), k4 = (q % (X3_BK / 4)) * 4;
      const int gr = r0 + r, gk = k0 + k4;
      const int gkc = (gk < kend) ? gk : k0;  // a slot past the operand re-reads the start of the stage
      src = p + (int64_t)(gr < R ? gr : R - 1) * ld + gkc;
      // VEC 4: ld is a multiple of 4 and gkc one too, so the vector ends inside the row.  VEC 2: the first pair ends at gkc + 1 <= ld - 1
      // (ld even), the second may start at / after kend.  VEC 1: every element on its own
      if (VEC == 2) o2 = (gkc + 2 < kend) ? 2 : 0;
      if (VEC == 1) { o1 = (gkc + 1 < kend) ? 1 : 0; o2 = (gkc + 2 < kend) ? 2 : 0; o3 = (gkc + 3 < kend) ? 3 : 0; }
    } else {
      const int k = q / (C::ROWS / 4), r4 = (q % (C::ROWS / 4)) * 4;
      const int gk = k0 + k, c = r0 + r4;
      const float* row = p + (int64_t)(gk < kend ? gk : k0) * ld;
      if (VEC == 1) {  // partial tile: every column clamped into the operand
        src = row + (c < R ? c : R - 1);
        o1 = (c + 1 < R) ? 1 : 0; o2 = (c + 2 < R) ? 2 : 0; o3 = (c + 3 < R) ? 3 : 0;
        if (c >= R) o1 = o2 = o3 = 0;
      } else {
        src = row + c;  // whole tile inside the operand (the caller's choice of VEC)
      }
    }
    if (VEC == 4) {
      t.v[i] = *reinterpret_cast<const float4*>(src);
    } else if (VEC == 2) {
      const float2 a = *reinterpret_cast<const float2*>(src);
      const float2 b = *reinterpret_cast<const float2*>(src + o2);
      t.v[i] = make_float4(a.x, a.y, b.x, b.y);
    } else {
      t.v[i] = make_float4(src[0], src[o1], src[o2], src[o3]);
    }
  }
}

template <class C>
__device__ __forceinline__ void x3_mask(Regs<C::NV>& t, int kcontig, int r0, int R, int k0, int kend) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < C::NV; ++i) {
    const int q = tid + i * C::NT;
    if (kcontig) {
      const int r = q / (X3_BK / 4), k4 = (q % (X3_BK / 4)) * 4;
      const int gk = k0 + k4;
      const bool rl = r0 + r < R;
      t.v[i] = make_float4(rl && gk + 0 < kend ? t.v[i].x : 0.f, rl && gk + 1 < kend ? t.v[i].y : 0.f, rl && gk + 2 < kend ? t.v[i].z : 0.f,
                           rl && gk + 3 < kend ? t.v[i].w : 0.f);
    } else {
      const int k = q / (C::ROWS / 4), c = r0 + (q % (C::ROWS / 4)) * 4;
      const bool kl = k0 + k < kend;
      t.v[i] = make_float4(kl && c + 0 < R ? t.v[i].x : 0.f, kl && c + 1 < R ? t.v[i].y : 0.f, kl && c + 2 < R ? t.v[i].z : 0.f,
                           kl && c + 3 < R ? t.v[i].w : 0.f);
    }
  }
}

// the exact three-way split of four consecutive elements -> one 8-byte write per plane
template <class C>
__device__ __forceinline__ void x3_store(const Regs<C::NV>& t, __bf16* __restrict__ s, int kcontig) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < C::NV; ++i) {
    const int q = tid + i * C::NT;
    const float x[4] = {t.v[i].x, t.v[i].y, t.v[i].z, t.v[i].w};
    bf16x4 p1, p2, p3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      p1[j] = (__bf16)x[j];
      const float r1 = x[j] - (float)p1[j];
      p2[j] = (__bf16)r1;
      const float r2 = r1 - (float)p2[j];
      p3[j] = (__bf16)r2;
    }
    int o;
    if (kcontig) {
      const int r = q / (X3_BK / 4), k4 = (q % (X3_BK / 4)) * 4;
      o = r * C::PITCH + k4;
    } else {  // natural [k][row] image
      const int k = q / (C::ROWS / 4), r4 = (q % (C::ROWS / 4)) * 4;
      o = k * C::RP + r4;
    }
    *reinterpret_cast<bf16x4*>(s + o) = p1;
    *reinterpret_cast<bf16x4*>(s + C::PLANE + o) = p2;
    *reinterpret_cast<bf16x4*>(s + 2 * C::PLANE + o) = p3;
  }
}

// MFMA operand (8 consecutive k of row `rowbase + lane % 32`, k half lane / 32) of k step ks from one plane (see gemm_bf16.hip: bf_fetch)
template <class C>
__device__ __forceinline__ bf16x8 x3_fetch(const __bf16* __restrict__ s, int kcontig, int rowbase, int ks, int lane) {
  if (kcontig) return *reinterpret_cast<const bf16x8*>(s + (rowbase + (lane & 31)) * C::PITCH + ks * 16 + 8 * (lane >> 5));
  const int g = lane >> 4, li = lane & 15, q = li >> 2, p = li & 3;
  const int k0 = ks * 16 + 8 * (g >> 1);
  const __bf16* a0 = s + (k0 + q) * C::RP + rowbase + 16 * (g & 1) + 4 * p;
  typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(a0 + 4 * C::RP));
  union { s16x4 h[2]; bf16x8 v; } u;
  u.h[0] = lo;
  u.h[1] = hi;
  return u.v;
}

__device__ __forceinline__ float x3_act_mask(float h, int act, bool keep, float scale) {
  if (!keep) return 0.f;
  if (act == HMP_ACT_RELU) return h > 0.f ? scale : 0.f;
  if (act == HMP_ACT_ELU) return h > 0.f ? scale : (h + scale);
  return scale;
}

// The K loop of one output tile.  AV / BV (the loaders' vector widths) are TEMPLATE arguments: a run-time `if (vec == 4) .. else ..`
// around the loads is a control-flow join after every operand's loads, and hipcc drains the memory counter at a join -- every
// stage then waited for its own prefetch right where it was issued (ISA: global_load x 4, s_waitcnt vmcnt(0), global_load x 8,
// s_waitcnt vmcnt(0)), which is what held the first version of this kernel at the fp32-MFMA kernel's speed.
template <class C, bool ONES, int AV, int BV>
__device__ __forceinline__ void x3_loop(const GemmProblem& P, int a_kc, int b_kc, int m0, int n0, int kbeg, int kend, __bf16* As, __bf16* Bs,
                                        f32x16 (&acc)[C::MI][C::NI], f32x16 (&acc1)[ONES ? C::MI : 1], bool ones_here, int wm, int wn, int lane) {
  constexpr int MI = C::MI, NI = C::NI;
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;
  // DEPTH register sets, loads DEPTH stages ahead (2 where the registers allow it: a stage is ~1 500 cycles of MFMAs, shorter than a
  // trip to HBM under load; the ones-column forms and the big tile hold more accumulator tiles and keep one set)
  constexpr int DEPTH = (ONES || MI * NI > 4) ? 1 : 2;
  using SA = typename C::SA;
  using SB = typename C::SB;
  Regs<SA::NV> ra0, ra1;
  Regs<SB::NV> rb0, rb1;
  auto load = [&](Regs<SA::NV>& ra, Regs<SB::NV>& rb, int k0) {
    x3_load<SA, AV>(ra, P.A, P.lda, a_kc, m0, P.M, k0, kend);
    x3_load<SB, BV>(rb, P.B, P.ldb, b_kc, n0, P.n_real, k0, kend);
  };
  auto stage = [&](Regs<SA::NV>& ra, Regs<SB::NV>& rb, int kt) {
    x3_mask<SA>(ra, a_kc, m0, P.M, kt, kend);
    x3_mask<SB>(rb, b_kc, n0, P.n_real, kt, kend);
    x3_store<SA>(ra, As, a_kc);
    x3_store<SB>(rb, Bs, b_kc);
    __syncthreads();
    if (kt + DEPTH * X3_BK < kend) load(ra, rb, kt + DEPTH * X3_BK);  // this set's next turn
#pragma unroll
    for (int ks = 0; ks < X3_BK / 16; ++ks) {
      // the A fragments of the k step stay in registers (3 pieces x MI), the B fragments come one column tile at a time
      bf16x8 av[3][MI];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < MI; ++i) av[p][i] = x3_fetch<SA>(As + p * SA::PLANE, a_kc, wm * (MI * 32) + i * 32, ks, lane);
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        bf16x8 bv[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) bv[p] = x3_fetch<SB>(Bs + p * SB::PLANE, b_kc, wn * (NI * 32) + j * 32, ks, lane);
        // smallest terms first: (3,1) (1,3) (2,2) (2,1) (1,2) (1,1)
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int tp = 0; tp < 6; ++tp)
#pragma unroll
          for (int i = 0; i < MI; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[PA[tp]][i], bv[PB[tp]], acc[i][j], 0, 0, 0);
      }
      if constexpr (ONES) {
        if (ones_here) {  // column sums of A: every piece against ones
#pragma unroll
          for (int p = 2; p >= 0; --p)
#pragma unroll
            for (int i = 0; i < MI; ++i) acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[p][i], ones, acc1[i], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  };
  load(ra0, rb0, kbeg);
  if constexpr (DEPTH == 2) {
    if (kbeg + X3_BK < kend) load(ra1, rb1, kbeg + X3_BK);
    for (int kt = kbeg; kt < kend; kt += 2 * X3_BK) {
      stage(ra0, rb0, kt);
      if (kt + X3_BK < kend) stage(ra1, rb1, kt + X3_BK);
    }
  } else {
    for (int kt = kbeg; kt < kend; kt += X3_BK) stage(ra0, rb0, kt);
  }
}

// FORM: 0 NT, 1 NN, 2 TN, 3 per problem at run time (mixed launches)
template <class C, bool ONES, int FORM>
__global__ __launch_bounds__(C::NT, C::PER_CU) void gemm_x3_kernel(const GemmBatch gb) {
  constexpr int MI = C::MI, NI = C::NI;
  extern __shared__ __attribute__((aligned(16))) unsigned char x3_lds[];
  __bf16* As = reinterpret_cast<__bf16*>(x3_lds);
  __bf16* Bs = As + 3 * C::SA::PLANE;
  int pi = 0;
  while (pi + 1 < gb.n && (int)blockIdx.x >= gb.p[pi + 1].tile_start) ++pi;
  const GemmProblem& P = gb.p[pi];
  const int local = blockIdx.x - P.tile_start;
  // XCD-aware order as in gemm.hip: K chunk fastest; row tiles grouped by 8 so that the column tiles of a row tile share an L2
  const int z = local % P.ksplit, t = local / P.ksplit;
  const int grp = t / (8 * P.tiles_n), within = t % (8 * P.tiles_n);
  const int rows_in_grp = min(8, P.tiles_m - grp * 8);
  const int m0 = (grp * 8 + within % rows_in_grp) * C::RM, n0 = (within / rows_in_grp) * C::RN;
  const int kbeg = z * P.kchunk;
  const int kend = min(P.K, kbeg + P.kchunk);
  const int a_kc = FORM == 3 ? (P.trans_a ? 0 : 1) : (FORM == 2 ? 0 : 1);
  const int b_kc = FORM == 3 ? (P.trans_b ? 1 : 0) : (FORM == 0 ? 1 : 0);
  // block-uniform loader choice: 16-byte rows, 8-byte rows, or element by element; a row-contiguous operand's vectors run ALONG its
  // rows, so its partial tiles (and odd pitches) go element by element
  auto vec_of = [](const float* p, int ld) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    return ((ld & 3) == 0 && (a & 15) == 0) ? 4 : (((ld & 1) == 0 && (a & 7) == 0) ? 2 : 1);
  };
  const int a_vec = (a_kc || m0 + C::RM <= P.M) ? vec_of(P.A, P.lda) : 1;
  const int b_vec = (b_kc || n0 + C::RN <= P.n_real) ? vec_of(P.B, P.ldb) : 1;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w % C::WMW, wn = w / C::WMW;
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

  // the K loop, instantiated per loader pair: the choice must not sit INSIDE the loop (see x3_loop)
  const int mode = a_vec * 8 + b_vec;
#define X3_LOOP(AV_, BV_) x3_loop<C, ONES, AV_, BV_>(P, a_kc, b_kc, m0, n0, kbeg, kend, As, Bs, acc, acc1, ones_here, wm, wn, lane)
  switch (mode) {
    case 4 * 8 + 4: X3_LOOP(4, 4); break;
    case 2 * 8 + 4: X3_LOOP(2, 4); break;
    case 4 * 8 + 2: X3_LOOP(4, 2); break;
    case 4 * 8 + 1: X3_LOOP(4, 1); break;
    case 1 * 8 + 4: X3_LOOP(1, 4); break;
    default: X3_LOOP(1, 1); break;  // (element loads are right for every alignment)
  }
#undef X3_LOOP

  // D layout of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); epilogue as gemm.hip (no branch
  // between memory operations)
  float* Cp = P.C + (int64_t)z * P.slab_stride;
  const int Mrows = P.M, Ncols = P.N, ldc = P.ldc, ldh = P.ldh, act = P.act;
  const bool amask = P.epi == EPI_ACTMASK;
  const bool dropon = P.drop_on != 0;
  const float dscale = dropon ? P.drop.scale : 1.f;
  const float* Hp = P.H;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = n0 + wn * (NI * 32) + j * 32 + (lane & 31);
      const int rbase = m0 + wm * (MI * 32) + i * 32 + 4 * (lane >> 5);
      const bool cok = col < Ncols && !(ONES && P.aug_ones && col >= P.n_real);  // (the ones column comes from acc1)
      const int colc = col < Ncols ? col : 0;
      float hv[16];
      if (amask) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          hv[r] = Hp[(int64_t)(row < Mrows ? row : Mrows - 1) * ldh + colc];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        float v = acc[i][j][r];
        if (amask) {
          const bool keep = !dropon || (__float_as_uint(hv[r]) != 0x80000000u);  // dropped elements were stored as -0.0f
          v *= x3_act_mask(hv[r], act, keep, dscale);
        }
        if (cok && row < Mrows) Cp[(int64_t)row * ldc + col] = v;
      }
    }
  if (ONES && ones_here && (lane & 31) == 0) {  // column 0 of the ones product -> C[:, n_real]
#pragma unroll
    for (int i = 0; i < (ONES ? MI : 1); ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (MI * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < P.M) Cp[(int64_t)row * P.ldc + P.n_real] = acc1[i][r];
      }
  }
}

template <class C>
int x3_launch_cfg(GemmBatch& gb, bool want_split, int max_slabs, hipStream_t st) {
  bool any_ones = false;
  int start = 0, all_tiles = 0;
  for (int i = 0; i < gb.n; ++i) {
    const GemmProblem& p = gb.p[i];
    any_ones = any_ones || p.aug_ones != 0;
    all_tiles += cdiv(p.M, C::RM) * cdiv(p.aug_ones ? (p.n_real > 0 ? p.n_real : 1) : p.N, C::RN);
  }
  // workgroups over the whole launch when split-K supplies them: one per CU and resident slot -- the kernel time is flat from 256 to 2048
  // at config 3, and every slab is written here and read again by the gradient un-pack
  const int target = 256 * C::PER_CU;
  for (int i = 0; i < gb.n; ++i) {
    GemmProblem& p = gb.p[i];
    p.tiles_m = cdiv(p.M, C::RM);
    p.tiles_n = cdiv(p.aug_ones ? (p.n_real > 0 ? p.n_real : 1) : p.N, C::RN);  // the ones column rides in the first column tile
    const int tiles = p.tiles_m * p.tiles_n;
    int ks = 1;
    if (want_split && tiles > 0) {  // every slab gets at least one K stage
      ks = target / (all_tiles > 0 ? all_tiles : 1);
      const int max_by_k = cdiv(p.K, X3_BK);
      if (ks > max_by_k) ks = max_by_k;
      if (ks > max_slabs) ks = max_slabs;
      if (ks < 1) ks = 1;
    }
    int kchunk = cdiv(cdiv(p.K, ks), X3_BK) * X3_BK;
    if (kchunk < X3_BK) kchunk = X3_BK;
    ks = p.K > 0 ? cdiv(p.K, kchunk) : 1;
    p.ksplit = ks;
    p.kchunk = kchunk;
    p.tile_start = start;
    start += tiles * ks;
  }
  gb.total_tiles = start;
  if (start == 0) return HMP_OK;
  int form = -1;
  for (int i = 0; i < gb.n; ++i) {
    const GemmProblem& p = gb.p[i];
    const int f = (!p.trans_a && p.trans_b) ? 0 : (!p.trans_a && !p.trans_b) ? 1 : (p.trans_a && !p.trans_b) ? 2 : 3;
    form = (form == -1 || form == f) ? f : 3;
  }
  static bool attr_done = false;  // (per configuration: one static per template instance)
  if (!attr_done) {
#define X3_ATTR(ONES_, FORM_) \
  HMP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_kernel<C, ONES_, FORM_>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES))
    X3_ATTR(true, 2); X3_ATTR(true, 3); X3_ATTR(false, 0); X3_ATTR(false, 1); X3_ATTR(false, 2); X3_ATTR(false, 3);
#undef X3_ATTR
    attr_done = true;
  }
#define X3_LAUNCH(ONES_, FORM_) hipLaunchKernelGGL((gemm_x3_kernel<C, ONES_, FORM_>), dim3(start), dim3(C::NT), C::LDS_BYTES, st, gb)
  if (any_ones) {
    if (form == 2) X3_LAUNCH(true, 2);
    else X3_LAUNCH(true, 3);
  } else {
    switch (form) {
      case 0: X3_LAUNCH(false, 0); break;
      case 1: X3_LAUNCH(false, 1); break;
      case 2: X3_LAUNCH(false, 2); break;
      default: X3_LAUNCH(false, 3); break;
    }
  }
#undef X3_LAUNCH
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

}  // namespace

// Launches of >= 10^9 multiply-adds whose problems are plain fp32 (no bf16-stored operand, no two-piece A, no gather-add term).
// HMP_GEMM_X3=0: the fp32-MFMA kernels of gemm.hip (tests compare the two).
// Which launches: plain fp32 problems (no bf16-stored operand, no two-piece A, no gather-add term) of >= 10^9 multiply-adds in total.
// Split-K weight gradients only over FEWER than 32 768 nodes (GAT at its batch size: 1 548 x 513 outputs over 5 490 nodes -- 0.341 ->
// 0.280 ms for the step's backward GEMMs against the direct fp32 TN kernel, whose 32 x 32 tiles re-read the operands 10x from HBM);
// over more nodes the outputs are at most 192 columns wide and the tall fp32 kernel of gemm.hip is faster (0.52 against 0.65 ms at
// batch 2048).  HMP_GEMM_X3=0: never; =2: every split-K launch.
static bool x3_takes_problems(const GemmProblem* ps, int n, bool want_split) {
  const char* v = getenv("HMP_GEMM_X3");
  if (v && v[0] == '0') return false;
  if (n <= 0) return false;
  const bool all_split = v && v[0] == '2';
  double work = 0.0;
  for (int i = 0; i < n; ++i) {
    const GemmProblem& p = ps[i];
    if (p.a_bf16 || p.b_bf16 || p.c_bf16 || p.h_bf16 || p.a_split || p.g_rowptr) return false;
    if (p.M < 0 || p.N < 0 || p.K < 0) return false;
    if (want_split && !all_split && p.K >= 32768) return false;
    work += (double)p.M * p.N * p.K;
  }
  return work >= 1e9;
}
bool gemm_x3_takes(const GemmBatch& gb, bool want_split) { return x3_takes_problems(gb.p, gb.n, want_split); }
// the executor's question before it picks the register-direct TN kernel for a step's weight gradients: would one of the launches of
// these problems (GEMM_MAX_PROB at a time, in order) come here?
bool gemm_x3_split_takes(const GemmProblem* ps, int n) {
  for (int base = 0; base < n; base += GEMM_MAX_PROB)
    if (x3_takes_problems(ps + base, n - base < GEMM_MAX_PROB ? n - base : GEMM_MAX_PROB, true)) return true;
  return false;
}

int gemm_x3_launch(GemmBatch& gb, bool want_split, int max_slabs, hipStream_t st) {
  HMP_CHECK_ARG(gb.n >= 0 && gb.n <= GEMM_MAX_PROB, "gemm_x3: %d problems", gb.n);
  for (int i = 0; i < gb.n; ++i) {
    const GemmProblem& p = gb.p[i];
    HMP_CHECK_ARG(!p.aug_ones || (p.trans_a && !p.trans_b && p.N == p.n_real + 1), "gemm_x3: the ones column belongs to the TN form with N = n_real + 1");
  }
  // 128 x 64 tiles at three workgroups per CU where the square tile wastes the chip: (a) fewer square tiles than workgroup slots
  // (the GAT input gradients: 172 tiles on 512 slots -- 0.338 -> 0.324 ms), (b) output widths that fill the last square column
  // tile badly (N = 192 at batch 2048: 256 padded columns against 192 -- projections 0.307 -> 0.268 ms, input gradients 0.52 -> 0.48).
  // Not where neither holds: the narrow tile re-reads the row operand once per 64 instead of 128 columns, and these products are
  // bound by operand traffic (GAT projections, N = 1536: 0.116 ms square, 0.130 ms narrow)
  if (!want_split) {
    int64_t t_sq = 0;
    double pad_sq = 0.0, pad_nr = 0.0;
    for (int i = 0; i < gb.n; ++i) {
      t_sq += (int64_t)cdiv(gb.p[i].M, 128) * cdiv(gb.p[i].N, 128);
      pad_sq += (double)gb.p[i].M * (cdiv(gb.p[i].N, 128) * 128);
      pad_nr += (double)gb.p[i].M * (cdiv(gb.p[i].N, 64) * 64);
    }
    const char* tv = getenv("HMP_GEMM_X3_TILE");  // 128 / 64: pin the square / the narrow tile (tests, measurements)
    const bool narrow = tv ? (tv[0] == '6') : (t_sq < 410 || pad_nr <= 0.8 * pad_sq);
    if (narrow) return x3_launch_cfg<X3Narrow>(gb, want_split, max_slabs, st);
  }
  return x3_launch_cfg<X3Small>(gb, want_split, max_slabs, st);
}

}  // namespace hmp
