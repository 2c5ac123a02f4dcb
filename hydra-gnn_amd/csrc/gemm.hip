// K2 -- grouped fp32 GEMM on the gfx950 matrix cores.
//
// v_mfma_f32_32x32x2_f32 is a k-ordered fp32 FMA chain (no TF32-style truncation exists on CDNA4), so the
// projection stays inside the 1e-5 parity budget of the reference's F.linear.  Scene-graph batches give
// tall-skinny problems (M = nodes: 10^2..10^6, N,K = feature widths: 6..512), and several of them per
// layer (one per node type), so the kernel is GROUPED: one launch walks a table of problems.
//
// Block = 4 wavefronts.  Two shapes:
//   <2,2,32>  64x64 tile, 2x2 waves, K stage of 32: throughput shape for big problems (many blocks per CU hide latency).
//   <1,1,128> 32x32 tile, the 4 waves split the K range of each staged tile (partials summed through LDS in a fixed
//             order => run-to-run identical), K stage of 128: LATENCY shape for the small problems of MP3D batches --
//             4x the workgroups, and each barrier interval carries 8 independent 16-byte loads per lane and 16 MFMAs
//             per wave instead of 2 and 4, so a K = 306 projection is 3 load round trips deep instead of 10.
// Operands are staged global -> registers -> LDS in [k][m] / [k][n] order, which makes every MFMA operand fetch a
// conflict-free ds_read_b32 (lanes 0-31 read 32 consecutive floats of one k row, lanes 32-63 the next k row); the
// next stage's global loads are issued before the current stage's MFMAs.
//
// Forms: NT (x * W^T, nn.Linear), NN (dZ * W, input gradient, optional activation-derivative epilogue),
// TN with split-K over node chunks (dZ^T * [x | 1], weight + bias gradient; slabs are reduced later).
#include <cstdlib>

#include "kernels.h"

namespace hmp {

KT_DEFINE(gemm)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// A (ROWS x BK) tile in registers; logical element (r, k).
//  kcontig = 1: memory is [r][k] (k contiguous)  -> per thread NV float4 along k
//  kcontig = 0: memory is [k][r] (r contiguous)  -> per thread NV float4 along r
template <int ROWS, int BK>
struct TileRegs {
  static constexpr int NV = ROWS * BK / 4 / 256;  // float4 per thread
  float4 v[NV];
};

// Fast path: NO control flow between the loads (a branchy loader makes hipcc drain vmcnt between slots, i.e. one load in
// flight at a time).  Out-of-range rows / k are CLAMPED to an in-range address and zeroed by selects afterwards, so any
// K stage of a k-contiguous operand (also the tail stage) and any K stage of an r-contiguous operand whose tile width is
// fully in range take this path.  A 16/8-byte vector never leaves the row: ld is a multiple of the vector width and the
// vector starts below kend <= ld.
template <int ROWS, int BK, int VEC>
__device__ __forceinline__ void tile_load_fast(TileRegs<ROWS, BK>& t, const float* __restrict__ p, int ld, int kcontig, int r0,
                                               int R, int k0, int kend) {
  // ISSUES the loads only (clamped addresses); tile_mask zeroes what lies outside the operand when the stage is consumed, one
  // iteration later -- a select on a loaded value right here makes the compiler wait for the load before the MFMAs it was
  // meant to overlap with.
  constexpr int NV = TileRegs<ROWS, BK>::NV;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int q = tid + i * 256;
    const float* src;
    int o1 = 1, o2 = 2, o3 = 3;  // element offsets (VEC < 4: clamped so that no access leaves [.., kend))
    if (kcontig) {
      const int r = q / (BK / 4), k4 = (q % (BK / 4)) * 4;
      const int gr = r0 + r, gk = k0 + k4;
      // clamp: row to the last row; k to the start of the stage (always < kend) when the slot starts out of range
      const int gkc = (gk < kend) ? gk : k0;
      src = p + (int64_t)(gr < R ? gr : R - 1) * ld + gkc;
      if (VEC == 1) { o1 = (gkc + 1 < kend) ? 1 : 0; o2 = (gkc + 2 < kend) ? 2 : 0; o3 = (gkc + 3 < kend) ? 3 : 0; }
      if (VEC == 2) { o2 = (gkc + 2 < kend) ? 2 : 0; }  // the second pair may start at / after kend: re-read the first pair then
    } else {
      const int k = q / (ROWS / 4), r4 = (q % (ROWS / 4)) * 4;
      const int gk = k0 + k;
      src = p + (int64_t)(gk < kend ? gk : k0) * ld + (r0 + r4);
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

template <int ROWS, int BK>
__device__ __forceinline__ void tile_mask(TileRegs<ROWS, BK>& t, int kcontig, int r0, int R, int k0, int kend) {
  constexpr int NV = TileRegs<ROWS, BK>::NV;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int q = tid + i * 256;
    if (kcontig) {
      const int r = q / (BK / 4), k4 = (q % (BK / 4)) * 4;
      const int gk = k0 + k4;
      const bool rlive = r0 + r < R;
      t.v[i] = make_float4(rlive && gk + 0 < kend ? t.v[i].x : 0.f, rlive && gk + 1 < kend ? t.v[i].y : 0.f,
                           rlive && gk + 2 < kend ? t.v[i].z : 0.f, rlive && gk + 3 < kend ? t.v[i].w : 0.f);
    } else {
      const int k = q / (ROWS / 4);
      if (k0 + k >= kend) t.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// Edge path (last K stage, last column tile, ones column): bounds-checked element by element.
// vec: 4 = 16-byte loads, 2 = 8-byte loads (rows only 8-byte aligned, e.g. ld = 306), 1 = scalar
template <int ROWS, int BK>
__device__ __forceinline__ void tile_load_edge(TileRegs<ROWS, BK>& t, const float* __restrict__ p, int ld, int kcontig, int r0,
                                               int R, int n_real, int aug, int k0, int kend, int vec) {
  constexpr int NV = TileRegs<ROWS, BK>::NV;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int q = tid + i * 256;
    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kcontig) {
      const int r = q / (BK / 4), k4 = (q % (BK / 4)) * 4;
      const int gr = r0 + r, gk = k0 + k4;
      if (gr < R && gk < kend) {
        const float* src = p + (int64_t)gr * ld + gk;
        if (vec == 4 && gk + 3 < kend) {
          val = *reinterpret_cast<const float4*>(src);
        } else if (vec == 2 && gk + 3 < kend) {
          const float2 a = *reinterpret_cast<const float2*>(src);
          const float2 b = *reinterpret_cast<const float2*>(src + 2);
          val = make_float4(a.x, a.y, b.x, b.y);
        } else {
          val.x = src[0];
          if (gk + 1 < kend) val.y = src[1];
          if (gk + 2 < kend) val.z = src[2];
          if (gk + 3 < kend) val.w = src[3];
        }
      }
    } else {
      const int k = q / (ROWS / 4), r4 = (q % (ROWS / 4)) * 4;
      const int gk = k0 + k, gr = r0 + r4;
      if (gk < kend) {
        const float* src = p + (int64_t)gk * ld + gr;
        if (vec == 4 && gr + 3 < n_real) {
          val = *reinterpret_cast<const float4*>(src);
        } else if (vec == 2 && gr + 3 < n_real) {
          const float2 a = *reinterpret_cast<const float2*>(src);
          const float2 b = *reinterpret_cast<const float2*>(src + 2);
          val = make_float4(a.x, a.y, b.x, b.y);
        } else {
          float e[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int c = gr + j;
            e[j] = (c < n_real) ? src[j] : ((aug && c == n_real) ? 1.0f : 0.0f);
          }
          val = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
    }
    t.v[i] = val;
  }
}

template <int ROWS, int BK>
__device__ __forceinline__ void tile_load(TileRegs<ROWS, BK>& t, const float* __restrict__ p, int ld, int kcontig, int r0,
                                          int R, int n_real, int aug, int k0, int kend, int vec) {
  // all conditions are block-uniform
  const bool fast = kcontig || (r0 + ROWS <= n_real);
  if (fast) {
    if (vec == 4) tile_load_fast<ROWS, BK, 4>(t, p, ld, kcontig, r0, R, k0, kend);
    else if (vec == 2) tile_load_fast<ROWS, BK, 2>(t, p, ld, kcontig, r0, R, k0, kend);
    else tile_load_fast<ROWS, BK, 1>(t, p, ld, kcontig, r0, R, k0, kend);
  } else {
    tile_load_edge<ROWS, BK>(t, p, ld, kcontig, r0, R, n_real, aug, k0, kend, vec);
  }
}

// LDS image is [k][LD] with LD = ROWS + 4.  A k-contiguous operand is transposed on the way in (each lane holds 4
// consecutive k of one row): written plainly, 32 lanes (k = 0,4,8,..) would hit 2 banks (4*LD = 16 mod 32: 16-way
// conflict), so row r of k-row k is ROTATED to column (r + k/4) mod ROWS: the 32 lanes then land on 32 different banks
// (bank = 17*(k/4) + r + const), and the MFMA operand fetch (32 consecutive r of one k) stays conflict-free.
// column of the rotated image: x in [0, 2 ROWS); ROWS need not be a power of two (192-column tiles)
template <int ROWS>
__device__ __forceinline__ int wrap_rows(int x) {
  if constexpr ((ROWS & (ROWS - 1)) == 0) return x & (ROWS - 1);
  else return x >= ROWS ? x - ROWS : x;
}
template <int ROWS, int BK>
__device__ __forceinline__ void tile_store(const TileRegs<ROWS, BK>& t, float* __restrict__ s, int kcontig) {
  constexpr int NV = TileRegs<ROWS, BK>::NV;
  constexpr int LD = ROWS + 4;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int q = tid + i * 256;
    if (kcontig) {
      const int r = q / (BK / 4), k4 = (q % (BK / 4)) * 4;
      const int rr = wrap_rows<ROWS>(r + (k4 >> 2));
      s[(k4 + 0) * LD + rr] = t.v[i].x;
      s[(k4 + 1) * LD + rr] = t.v[i].y;
      s[(k4 + 2) * LD + rr] = t.v[i].z;
      s[(k4 + 3) * LD + rr] = t.v[i].w;
    } else {
      const int k = q / (ROWS / 4), r4 = (q % (ROWS / 4)) * 4;
      *reinterpret_cast<float4*>(&s[k * LD + r4]) = t.v[i];
    }
  }
}

__device__ __forceinline__ float act_mask_factor(float h, int act, bool keep, float scale) {
  // d out / d pre for out = dropout(act(pre)) given the stored out value h
  if (!keep) return 0.f;
  if (act == HMP_ACT_RELU) return h > 0.f ? scale : 0.f;
  if (act == HMP_ACT_ELU) return h > 0.f ? scale : (h + scale);  // elu'(pre) = elu(pre) + 1 for pre <= 0
  return scale;
}

__device__ __forceinline__ int vec_mode(const float* p, int ld, bool k_ok) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  if (!k_ok) return 1;
  if ((ld & 3) == 0 && (a & 15) == 0) return 4;
  if ((ld & 1) == 0 && (a & 7) == 0) return 2;
  return 1;
}

// FORM fixes the operand layouts at compile time (0: NT = x * W^T, 1: NN = dZ * W, 2: TN = dZ^T * [x | 1]; 3: per problem at run
// time): with run-time layouts the layout branch sits inside the unrolled load loops and the loads stop overlapping.
// MI x NI 32x32 accumulator tiles per wave (round 2): <2,2,32,.,2,2> = a 128x128 tile, every operand fragment fetched from LDS
// feeds two MFMAs instead of one and the tile re-reads A / B from L2 half as often -- the 64x64 form ran the big problems
// (GAT projections, batch-2048 weight gradients) at 34-60 TFLOP/s, bound by L2 -> CU operand traffic at 16 flop per byte.
template <int WM, int WN, int BK, int FORM, int MI = 1, int NI = 1>
// second launch-bound = waves per SIMD = blocks per CU for 256-thread blocks: >= 3 so that the ~600 workgroups of an MP3D
// layer-0 projection are all resident at once (one round instead of two)
__global__ __launch_bounds__(256, (MI * NI > 1) ? 2 : 3) void gemm_kernel(const GemmBatch gb) {
  constexpr int BM = 32 * WM * MI, BN = 32 * WN * NI, KW = 4 / (WM * WN);
  static_assert(MI * NI == 1 || KW == 1, "register tiling only with one K group");
  constexpr int LDA = BM + 4, LDB = BN + 4;
  constexpr int STAGE = BK * LDA + BK * LDB;
  constexpr int RED = (KW > 1) ? (KW - 1) * 16 * 64 : 0;
  constexpr int SMEM = STAGE > RED ? STAGE : RED;
  __shared__ __attribute__((aligned(16))) float smem[SMEM];
  float* As = smem;
  float* Bs = smem + BK * LDA;

  int pi = 0;
  while (pi + 1 < gb.n && (int)blockIdx.x >= gb.p[pi + 1].tile_start) ++pi;
  const GemmProblem& P = gb.p[pi];
  const int local = blockIdx.x - P.tile_start;
  const int tiles_mn = P.tiles_m * P.tiles_n;
  // XCD-aware order (blocks b and b+8 share an XCD and its 4 MB L2; speed only, never correctness):
  //  * split-K: the K chunk index is the fastest-varying part of the block id, so with ksplit = 8 (or 16) all tiles
  //    of one chunk run on one XCD and its A/B panels are fetched into that L2 once;
  //  * otherwise row tiles are grouped by 8 and the column tiles of one row tile are 8 block ids apart, so the row
  //    panel of A (the big operand: nodes x features) is fetched once per XCD instead of once per column tile.
  const int z = local % P.ksplit, t = local / P.ksplit;
  (void)tiles_mn;
  const int grp = t / (8 * P.tiles_n), within = t % (8 * P.tiles_n);
  const int rows_in_grp = min(8, P.tiles_m - grp * 8);
  const int m0 = (grp * 8 + within % rows_in_grp) * BM, n0 = (within / rows_in_grp) * BN;
  const int kbeg = z * P.kchunk;
  const int kend = min(P.K, kbeg + P.kchunk);

  const int a_kcontig = FORM == 3 ? (P.trans_a ? 0 : 1) : (FORM == 2 ? 0 : 1);
  const int b_kcontig = FORM == 3 ? (P.trans_b ? 1 : 0) : (FORM == 0 ? 1 : 0);
  // fast (clamp + deferred mask) or edge (bounds-checked, final values) loader per operand; block-uniform
  const bool a_fast = a_kcontig || (m0 + BM <= P.M);
  const bool b_fast = b_kcontig || (n0 + BN <= P.n_real);
  // kbeg is a multiple of BK (>= 32), so k offsets keep the row alignment class
  const int a_vec = vec_mode(P.A, P.lda, true);
  const int b_vec = vec_mode(P.B, P.ldb, true);

  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wm = w % WM, wn = (w / WM) % WN, kw = w / (WM * WN);

  f32x16 acc_t[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc_t[i][j][r] = 0.f;
  if (P.Cadd && kw == 0 && z == 0) {  // block-uniform per K group: the product accumulates ON TOP of the addend (GemmProblem::Cadd)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * 32 * MI + 32 * i + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
          const int col = n0 + wn * 32 * NI + 32 * j + (lane & 31);
          acc_t[i][j][r] = (row < P.M && col < P.N) ? P.Cadd[(int64_t)row * P.ldadd + col] : 0.f;
        }
  }
  f32x16& acc = acc_t[0][0];

  TileRegs<BM, BK> ra;
  TileRegs<BN, BK> rb;
  KT(0);
  // for A the "row" dimension is M; for B it is N (n_real real columns, optional virtual ones column)
  tile_load<BM, BK>(ra, P.A, P.lda, a_kcontig, m0, P.M, P.M, 0, kbeg, kend, a_vec);
  tile_load<BN, BK>(rb, P.B, P.ldb, b_kcontig, n0, P.n_real, P.n_real, P.aug_ones, kbeg, kend, b_vec);

  int kti = 0;
  (void)kti;
  for (int kt = kbeg; kt < kend; kt += BK) {
    if (a_fast) tile_mask<BM, BK>(ra, a_kcontig, m0, P.M, kt, kend);
    if (b_fast) tile_mask<BN, BK>(rb, b_kcontig, n0, P.n_real, kt, kend);
    tile_store<BM, BK>(ra, As, a_kcontig);
    tile_store<BN, BK>(rb, Bs, b_kcontig);
    __syncthreads();
    KT(1 + (kti++));
    if (kt + BK < kend) {
      tile_load<BM, BK>(ra, P.A, P.lda, a_kcontig, m0, P.M, P.M, 0, kt + BK, kend, a_vec);
      tile_load<BN, BK>(rb, P.B, P.ldb, b_kcontig, n0, P.n_real, P.n_real, P.aug_ones, kt + BK, kend, b_vec);
    }
    constexpr int KS = BK / KW;
    const int klen = min(BK, kend - kt);  // tail stage: skip the k rows that are all zero
    const int ma = wm * 32 * MI + (lane & 31), mb = wn * 32 * NI + (lane & 31);
    // KS is a multiple of 4 and kk is even, so (k >> 2) does not depend on the lane half: the rotation is uniform
    auto mfma_pair = [&](int kk) {
      const int k = kw * KS + kk + (lane >> 5);
      const int rot = (kw * KS + kk) >> 2;
      float av[MI], bv[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) av[i] = As[k * LDA + (a_kcontig ? wrap_rows<BM>(ma + 32 * i + rot) : ma + 32 * i)];  // undo the store rotation
#pragma unroll
      for (int j = 0; j < NI; ++j) bv[j] = Bs[k * LDB + (b_kcontig ? wrap_rows<BN>(mb + 32 * j + rot) : mb + 32 * j)];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc_t[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc_t[i][j], 0, 0, 0);
    };
    if (klen == BK) {
      // full stage: straight-line code, so the scheduler can run the LDS operand reads ahead of the MFMAs
#pragma unroll
      for (int kk = 0; kk < KS; kk += 2) mfma_pair(kk);
    } else {
      // tail stage: skip groups of 8 k rows that lie entirely in the zero padding
#pragma unroll
      for (int g = 0; g < KS; g += 8) {
        if (kw * KS + g < klen) {
#pragma unroll
          for (int kk = g; kk < g + 8; kk += 2) mfma_pair(kk);
        }
      }
    }
    __syncthreads();
  }

  KT(8);
  if (KW > 1) {  // fixed-order cross-wave reduction through LDS
    float* red = smem;
    if (kw > 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) red[((kw - 1) * 16 + i) * 64 + lane] = acc[i];
    }
    __syncthreads();
    if (kw == 0) {
#pragma unroll
      for (int q = 0; q < KW - 1; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] += red[(q * 16 + i) * 64 + lane];
    }
  }
  if (kw != 0) return;

  // A = [i][k] supplies the rows of C, B = [k][j] its columns;
  // D layout: col j = lane & 31, row i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
  // No branch between memory operations: fields in registers, the 16 activation values requested together (clamped addresses),
  // predicated stores.  (With a `continue` per element the compiler waited for every H load before issuing the next one.)
  float* C = P.C + (int64_t)z * P.slab_stride;
  const int Mrows = P.M, Ncols = P.N, ldc = P.ldc, ldh = P.ldh, act = P.act;
  const bool amask = P.epi == EPI_ACTMASK;
  const bool dropon = P.drop_on != 0;
  const float dscale = dropon ? P.drop.scale : 1.f;
  const float* Hp = P.H;
#pragma unroll
  for (int ti = 0; ti < MI; ++ti)
#pragma unroll
    for (int tj = 0; tj < NI; ++tj) {
      const int col = n0 + wn * 32 * NI + 32 * tj + (lane & 31);
      const bool cok = col < Ncols;
      const int colc = cok ? col : 0;
      const int rbase = m0 + wm * 32 * MI + 32 * ti + 4 * (lane >> 5);
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
        float v = acc_t[ti][tj][r];
        if (amask) {
          // the forward stored dropped elements as -0.0f: the keep bit is the sign of a zero, no RNG replay needed
          const bool keep = !dropon || (__float_as_uint(hv[r]) != 0x80000000u);
          v *= act_mask_factor(hv[r], act, keep, dscale);
        }
        if (cok && row < Mrows) C[(int64_t)row * ldc + col] = v;
      }
    }
  KT(9);
}

template <int WM, int WN, int BK, int MI = 1, int NI = 1>
static int launch_cfg(GemmBatch& gb, bool want_split, int max_slabs, hipStream_t st) {
  constexpr int BM = 32 * WM * MI, BN = 32 * WN * NI;
  int start = 0;
  int all_tiles = 0;
  for (int i = 0; i < gb.n; ++i) all_tiles += cdiv(gb.p[i].M, BM) * cdiv(gb.p[i].N, BN);
  for (int i = 0; i < gb.n; ++i) {
    GemmProblem& p = gb.p[i];
    p.tiles_m = cdiv(p.M, BM);
    p.tiles_n = cdiv(p.N, BN);
    int tiles = p.tiles_m * p.tiles_n;
    int ks = 1;
    if (want_split && tiles > 0) {
      // aim at ~1024 workgroups over the whole launch; every slab gets >= 1 full K stage
      ks = 1024 / (all_tiles > 0 ? all_tiles : 1);
      int max_by_k = cdiv(p.K, BK);
      if (ks > max_by_k) ks = max_by_k;
      if (ks > max_slabs) ks = max_slabs;
      if (ks < 1) ks = 1;
      while (ks & (ks - 1)) ks &= ks - 1;  // power of two (XCD affinity of the K chunks, see the kernel)
    }
    int kchunk = cdiv(cdiv(p.K, ks), BK) * BK;
    if (kchunk < BK) kchunk = BK;
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
  switch (form) {
    case 0: hipLaunchKernelGGL((gemm_kernel<WM, WN, BK, 0, MI, NI>), dim3(start), dim3(256), 0, st, gb); break;
    case 1: hipLaunchKernelGGL((gemm_kernel<WM, WN, BK, 1, MI, NI>), dim3(start), dim3(256), 0, st, gb); break;
    case 2: hipLaunchKernelGGL((gemm_kernel<WM, WN, BK, 2, MI, NI>), dim3(start), dim3(256), 0, st, gb); break;
    default: hipLaunchKernelGGL((gemm_kernel<WM, WN, BK, 3, MI, NI>), dim3(start), dim3(256), 0, st, gb); break;
  }
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Weight gradients of a LARGE batch of narrow layers: C[M, N] = A[K, M]^T * [B | 1][K, N] with K = 10^5 .. 10^6 nodes, M <= 192
// stacked output columns, N = input width + 1 (the reference's own batch size: 2048 MP3D graphs = 176 k objects, M = 192, N = 307).
// The 64x64-tile split-K kernel above covers such an output with 15 tiles, i.e. every K chunk of A is read 5 times and of B 3
// times through L2, for 15 MFMAs of 32x32x2 per 4 LDS operand reads: 0.80 ms = 37 TFLOP/s at batch 2048, 40 % of the step.
// Here a workgroup (4 waves) takes one K chunk and the WHOLE M extent for an 80-column slice of N: A is staged once per chunk
// and slice (4 slices at N = 307), B once; a wave owns 3 x 5 tiles of 16x16 (v_mfma_f32_16x16x4_f32: 15 MFMAs per 8 operand
// reads, 60 accumulator registers).  Both operands are row-contiguous in memory ([k][m], [k][n]) and keep that layout in LDS
// (pitch = 16 mod 32 floats: the four k rows of a step land on different banks), two stage buffers of 32 nodes, one barrier per
// stage.  Slabs as before: block (chunk z, slice q) writes its partial product into slab z; the gradient un-pack sums the slabs
// in fixed order.
constexpr int TT_M = 192, TT_NS = 80, TT_BK = 32;
constexpr int TT_LDA = TT_M + 16, TT_LDB = TT_NS;  // both = 16 mod 32
constexpr int TT_STAGE = TT_BK * (TT_LDA + TT_LDB);  // floats per stage buffer
typedef float f32x4t __attribute__((ext_vector_type(4)));
struct TallProblem {
  const float* A;
  const float* B;
  float* C;
  int64_t slab_stride;
  int M, N, K, lda, ldb, ldc, n_real, aug_ones;
  int ksplit, kchunk, nslices, block_start;
};
struct TallBatch {
  int n;
  TallProblem p[GEMM_MAX_PROB];
};

__global__ __launch_bounds__(256, 2) void gemm_tn_tall_kernel(const TallBatch tb) {
  extern __shared__ __attribute__((aligned(16))) float tt_lds[];  // [2][TT_STAGE]
  int pi = 0;
  while (pi + 1 < tb.n && (int)blockIdx.x >= tb.p[pi + 1].block_start) ++pi;
  const TallProblem& P = tb.p[pi];
  const int local = (int)blockIdx.x - P.block_start;
  const int z = local / P.nslices, q = local - z * P.nslices;
  const int n0 = q * TT_NS;
  const int kbeg = z * P.kchunk, kend = min(P.K, kbeg + P.kchunk);
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kq = lane >> 4;

  f32x4t acc[3][5];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 5; ++c) acc[r][c] = (f32x4t){0.f, 0.f, 0.f, 0.f};

  // stage loads: A 32 x 192 floats = 1536 float4 (6 per thread), B 32 x 80 floats = 1280 float2 (5 per thread)
  float4 ra[6];
  float2 rb[5];
  auto load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int s = tid + i * 256;
      const int k = s / 48, m4 = (s - k * 48) * 4;
      const int gk = min(k0 + k, kend - 1);               // clamped: masked when stored
      const int mc = m4 < P.M ? m4 : 0;                   // M is a multiple of 4 (checked by the host)
      ra[i] = *reinterpret_cast<const float4*>(P.A + (int64_t)gk * P.lda + mc);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int s = tid + i * 256;
      const int k = s / 40, n2 = (s - k * 40) * 2;
      const int gk = min(k0 + k, kend - 1);
      const int col = n0 + n2;
      const int cc = col + 1 < P.n_real ? col : 0;        // a pair that reaches past the real columns is rebuilt below
      rb[i] = *reinterpret_cast<const float2*>(P.B + (int64_t)gk * P.ldb + cc);
    }
  };
  auto put = [&](int k0, float* buf) {
    float* As = buf;
    float* Bs = buf + TT_BK * TT_LDA;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int s = tid + i * 256;
      const int k = s / 48, m4 = (s - k * 48) * 4;
      const bool live = k0 + k < kend && m4 < P.M;
      *reinterpret_cast<float4*>(As + k * TT_LDA + m4) = live ? ra[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int s = tid + i * 256;
      const int k = s / 40, n2 = (s - k * 40) * 2;
      const bool klive = k0 + k < kend;
      const int col = n0 + n2;
      float2 v = rb[i];
      if (col + 1 >= P.n_real) {  // the last real column, the ones column, padding
        const float x0 = col < P.n_real ? P.B[(int64_t)min(k0 + k, kend - 1) * P.ldb + col] : ((P.aug_ones && col == P.n_real) ? 1.f : 0.f);
        const float x1 = (P.aug_ones && col + 1 == P.n_real) ? 1.f : 0.f;
        v = make_float2(x0, x1);
      }
      *reinterpret_cast<float2*>(Bs + k * TT_LDB + n2) = klive ? v : make_float2(0.f, 0.f);
    }
  };
  const int nst = (kend - kbeg + TT_BK - 1) / TT_BK;
  if (nst > 0) {
    load(kbeg);
    put(kbeg, tt_lds);
  }
  __syncthreads();
  for (int s = 0; s < nst; ++s) {
    float* cur = tt_lds + (s & 1) * TT_STAGE;
    if (s + 1 < nst) load(kbeg + (s + 1) * TT_BK);
    const float* As = cur + kq * TT_LDA + w * 48 + i16;        // wave w: rows 48 w .. 48 w + 47 of the output
    const float* Bs = cur + TT_BK * TT_LDA + kq * TT_LDB + i16;
#pragma unroll
    for (int ks = 0; ks < TT_BK / 4; ++ks) {
      float av[3], bv[5];
#pragma unroll
      for (int r = 0; r < 3; ++r) av[r] = As[ks * 4 * TT_LDA + r * 16];
#pragma unroll
      for (int c = 0; c < 5; ++c) bv[c] = Bs[ks * 4 * TT_LDB + c * 16];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 5; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r], bv[c], acc[r][c], 0, 0, 0);
    }
    if (s + 1 < nst) put(kbeg + (s + 1) * TT_BK, tt_lds + ((s + 1) & 1) * TT_STAGE);
    __syncthreads();
  }
  // D layout of a 16x16 tile: column = lane & 15, row = 4 * (lane >> 4) + reg
  float* C = P.C + (int64_t)z * P.slab_stride;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const int col = n0 + c * 16 + i16;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int row = w * 48 + r * 16 + 4 * kq + e;
        if (row < P.M && col < P.N) C[(int64_t)row * P.ldc + col] = acc[r][c][e];
      }
    }
}

// problems of a split-K launch the tall kernel takes
static bool tall_takes(const GemmProblem& p) {
  const char* v = getenv("HMP_GEMM_TALL");  // 0: the tiled split-K kernel for every weight gradient (tests compare the two)
  if (v && v[0] == '0') return false;
  return p.trans_a && !p.trans_b && p.M > 0 && p.M <= TT_M && (p.M & 3) == 0 && p.N > 0 && p.K >= 32768 && (p.lda & 3) == 0 && (p.ldb & 1) == 0 &&
         (reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.B) & 7) == 0 && p.epi == EPI_NONE &&
         (p.n_real & 1) == 0;
}

int gemm_launch(GemmBatch& gb, bool want_split, int max_slabs, hipStream_t st) {
  HMP_CHECK_ARG(gb.n >= 0 && gb.n <= GEMM_MAX_PROB, "gemm: %d problems", gb.n);
  // launches of >= 10^9 multiply-adds: fp32 accuracy from six bf16 piece products per element product (gemm_x3.hip)
  if (gemm_x3_takes(gb, want_split)) return gemm_x3_launch(gb, want_split, max_slabs, st);
  if (want_split) {  // weight gradients over >= 32768 nodes with <= 192 stacked columns: the tall kernel (one launch for all of them)
    TallBatch tb;
    memset(&tb, 0, sizeof(tb));
    GemmBatch rest;
    memset(&rest, 0, sizeof(rest));
    int idx_t[GEMM_MAX_PROB], idx_r[GEMM_MAX_PROB];
    double work_total = 0.0;
    for (int i = 0; i < gb.n; ++i)
      if (tall_takes(gb.p[i])) work_total += (double)gb.p[i].K * cdiv(gb.p[i].N, TT_NS);
    for (int i = 0; i < gb.n; ++i) {
      const GemmProblem& g = gb.p[i];
      if (!tall_takes(g)) {
        idx_r[rest.n] = i;
        rest.p[rest.n++] = g;
        continue;
      }
      TallProblem& P = tb.p[tb.n];
      idx_t[tb.n++] = i;
      P.A = g.A; P.B = g.B; P.C = g.C; P.slab_stride = g.slab_stride;
      P.M = g.M; P.N = g.N; P.K = g.K; P.lda = g.lda; P.ldb = g.ldb; P.ldc = g.ldc; P.n_real = g.n_real; P.aug_ones = g.aug_ones;
      P.nslices = cdiv(g.N, TT_NS);
      // ~2 workgroups per CU over the whole launch, shared out in proportion to chunks x slices; <= GEMM_TALL_SLABS slabs
      // (measured at batch 2048: 256 / 512 / 768 / 1024 workgroups -> 0.80 / 0.51 / 0.58 / 0.52 ms for the backward GEMMs)
      int ks = (int)(512.0 * ((double)g.K * P.nslices / work_total) / P.nslices + 0.5);
      if (ks > GEMM_TALL_SLABS) ks = GEMM_TALL_SLABS;
      if (ks < 1) ks = 1;
      int kchunk = cdiv(cdiv(g.K, ks), TT_BK) * TT_BK;
      ks = cdiv(g.K, kchunk);
      P.ksplit = ks; P.kchunk = kchunk;
    }
    if (tb.n > 0) {
      int start = 0;
      for (int i = 0; i < tb.n; ++i) {
        tb.p[i].block_start = start;
        start += tb.p[i].ksplit * tb.p[i].nslices;
        gb.p[idx_t[i]].ksplit = tb.p[i].ksplit;
        gb.p[idx_t[i]].kchunk = tb.p[i].kchunk;
      }
      static const int rc = [] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_tall_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   2 * TT_STAGE * (int)sizeof(float)) == hipSuccess ? 0 : 1;
      }();
      HMP_CHECK_ARG(rc == 0, "gemm: could not raise the dynamic LDS limit of the tall weight-gradient kernel");
      hipLaunchKernelGGL(gemm_tn_tall_kernel, dim3(start), dim3(256), 2 * TT_STAGE * sizeof(float), st, tb);
      HMP_LAUNCH_CHECK();
      if (rest.n == 0) return HMP_OK;
      const int rc2 = gemm_launch(rest, want_split, max_slabs, st);  // (tall_takes is false for every problem of `rest`)
      for (int i = 0; i < rest.n; ++i) {
        gb.p[idx_r[i]].ksplit = rest.p[i].ksplit;
        gb.p[idx_r[i]].kchunk = rest.p[i].kchunk;
      }
      return rc2;
    }
  }
  int64_t tiles64 = 0;
  double work = 0.0;
  int max_k = 0;
  for (int i = 0; i < gb.n; ++i) {
    const GemmProblem& p = gb.p[i];
    HMP_CHECK_ARG(p.M >= 0 && p.N >= 0 && p.K >= 0, "gemm: negative size");
    tiles64 += (int64_t)cdiv(p.M, 64) * cdiv(p.N, 64);
    work += (double)p.M * p.N * p.K;
    max_k = p.K > max_k ? p.K : max_k;
  }
  // big problems (many tiles, or few tiles over a very deep K: the weight gradients of a 10^6-node graph): 64x64 tiles;
  // small ones: 32x32 tiles with in-block K split and deep K stages
  // 128x128 tiles (2x2 accumulator tiles per wave) once they still fill the chip: >= 256 of them, or a split-K launch (the
  // K chunks supply the workgroups); HMP_GEMM_BIG=0 keeps the 64x64 form (tests compare the two)
  if (tiles64 >= 1024 || work >= 1e9) {
    int64_t tiles128 = 0;
    for (int i = 0; i < gb.n; ++i) tiles128 += (int64_t)cdiv(gb.p[i].M, 128) * cdiv(gb.p[i].N, 128);
    const char* gv = getenv("HMP_GEMM_BIG");
    const bool big_ok = !(gv && gv[0] == '0');
    // ... and only for wide outputs (every N >= 256): measured on MI355X, the GAT backward GEMMs (N = 512 .. 1100) gain 18 %
    // (0.448 -> 0.366 ms per step, config 3), but at N = 192 / 65 (config 2 at batch 2048) the half-empty second column tile and
    // two workgroups per CU instead of three LOSE 20 % (gemm 1.33 -> 1.67 ms)
    int min_n = 1 << 30;
    for (int i = 0; i < gb.n; ++i) min_n = gb.p[i].N < min_n ? gb.p[i].N : min_n;
    if (big_ok && min_n >= 256 && work >= 1e9 && (tiles128 >= 256 || (want_split && max_k >= 8192)))
      return launch_cfg<2, 2, 32, 2, 2>(gb, want_split, max_slabs, st);
    // round 3: outputs of 129 .. 192 columns (the three stacked 64-column blocks of an MP3D layer at the reference's batch size:
    // 190 k rows x 192) take the WHOLE width in one workgroup -- 64 x 192 tiles, 1 x 3 accumulator tiles per wave: the node rows
    // are read once instead of three times and an A fragment feeds three MFMAs (246 VGPRs, no spill; the 128 x 192 form with
    // 2 x 3 tiles per wave spills 96 registers at two workgroups per CU and was not kept).  HMP_GEMM_WIDE=0: the 64x64 form
    int max_n = 0;
    for (int i = 0; i < gb.n; ++i) max_n = gb.p[i].N > max_n ? gb.p[i].N : max_n;
    const char* wv = getenv("HMP_GEMM_WIDE");
    if (!(wv && wv[0] == '0') && !want_split && min_n > 128 && max_n <= 192 && work >= 1e9)
      return launch_cfg<2, 2, 32, 1, 3>(gb, want_split, max_slabs, st);
    return launch_cfg<2, 2, 32>(gb, want_split, max_slabs, st);
  }
  if (max_k <= 64) return launch_cfg<1, 1, 64>(gb, want_split, max_slabs, st);
  return launch_cfg<1, 1, 128>(gb, want_split, max_slabs, st);
}

}  // namespace hmp

extern "C" int hmp_gemm_f32(const float* d_a, int32_t lda, int32_t trans_a, const float* d_b, int32_t ldb, int32_t trans_b,
                            float* d_c, int32_t ldc, int32_t M, int32_t N, int32_t K, void* stream) {
  using namespace hmp;
  HMP_CHECK_ARG(d_a && d_b && d_c, "hmp_gemm_f32: null pointer");
  HMP_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "hmp_gemm_f32: negative size");
  HMP_CHECK_ARG(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N, "hmp_gemm_f32: leading dimension too small");
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
  return gemm_launch(gb, false, 1, (hipStream_t)stream);
}
