// Front kernel of the small-batch step: everything that has to exist before the first aggregation, in ONE launch.
//
// Three independent jobs precede layer 0's aggregation: the graph plan (CSR/CSC of every edge type), the parameter pack
// (stacked weights of layers >= 1, bias sums) and the layer-0 projection Z[0][s] = x_s * Wp[0][s]^T.  Run one after the
// other they cost ~45 us of a ~130 us MP3D step although each keeps only a fraction of the chip busy: a plan part has
// to read its whole edge list through ONE compute unit (~25 GB/s of fetch bandwidth per CU -> ~18 us for 17k edges),
// while the projection is a few hundred 64x64 tiles.  Here they are ROLES of one 1024-thread launch (<= one block per
// CU, all co-resident):
//   [0, gemm_blocks)   projection tiles: 64x64, 16 waves = 4 K-groups x (2x2 sub-tiles), K staged through LDS in 128-deep
//                      stages whose global loads are ALL issued before the first stage is consumed (one memory round
//                      trip for K <= 384 instead of one per stage); the four K-group partials are summed through LDS in
//                      a fixed order.  The B operand is read straight from the flat parameter buffer (a stacked row =
//                      sum of <= 6 parameter rows), so the projection does not wait for the pack role.
//   then plan parts    plan_small_part (plan_small.h), unchanged
//   then pack blocks   16 (packed row, 64-column chunk) items per block
// The step's critical path sees max(projection, plan) ~ 20 us instead of their sum.
#include "device_fns.h"
#ifdef HMP_KTIME
// phase stamps of the FIRST plan part (job 0, by destination, part 0) into this file's stamp buffer
#define PS_KT(i)                                                                           \
  do {                                                                                     \
    if (threadIdx.x == 0 && part == 0 && dir == 0 && J.ei == front_kt_ei) front_kt_slot(i); \
  } while (0)
namespace hmp {
__device__ const void* front_kt_ei;
__device__ void front_kt_slot(int i);
}
#endif
#include "plan_small.h"

namespace hmp {

KT_DEFINE(front)
#ifdef HMP_KTIME
__device__ void front_kt_slot(int i) { kt_buf[i] = wall_clock64(); }
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int FT = 64;         // tile edge
constexpr int FBK = 128;       // K stage
constexpr int FLD = FBK + 2;    // LDS pitch (floats) of a [row][k] image: 8-byte aligned rows, row m starts in bank 2m mod 32
constexpr int FSTAGES = 3;     // K <= 384
constexpr int F_STAGE_FLOATS = 2 * 2 * FT * FLD;  // two buffers (the stages alternate) of an A and a B image
constexpr int F_RED_FLOATS = 16 * 16 * 64;
constexpr int F_GEMM_FLOATS = F_STAGE_FLOATS > F_RED_FLOATS ? F_STAGE_FLOATS : F_RED_FLOATS;
constexpr int F_SEG_OFF = F_GEMM_FLOATS * 4;  // byte offset of the segment table copy (behind the stage / reduction area)
constexpr int F_GEMM_BYTES = F_SEG_OFF + FR_MAX_SEG * (int)sizeof(FrontSeg);
constexpr int F_LDS_BYTES = PS_LDS_BYTES > F_GEMM_BYTES ? PS_LDS_BYTES : F_GEMM_BYTES;

template <int VEC>
__device__ __forceinline__ float4 fr_ld4(const float* p, bool second_ok) {
  if (VEC == 4) return *reinterpret_cast<const float4*>(p);
  const float2 a = *reinterpret_cast<const float2*>(p);
  const float2 b = *reinterpret_cast<const float2*>(p + (second_ok ? 2 : 0));  // the second pair may start at/after K
  return make_float4(a.x, a.y, b.x, b.y);
}

template <int VEC, int NSRC>
__device__ __forceinline__ void front_gemm_tile(const FrontProb& P, const FrontSeg* __restrict__ sseg, const float* __restrict__ params,
                                                int tm, int tn, float* lds) {
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = tm * FT, n0 = tn * FT;
  // slot geometry: slot q = tid + i * 1024 (i < 2) covers row q >> 5, k4 = (q & 31) * 4 of a 64 x 128 stage
  const int k4 = (tid & 31) * 4;
  const float* arow[2];
  const float* brow[2][NSRC];
  int bn[2];  // live sources of the B row (0: padding row -> zeros)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (tid >> 5) + 32 * i;
    arow[i] = P.A + (int64_t)min(m0 + row, P.M - 1) * P.lda;
    const int c = n0 + row;
    int si = 0;
    while (si + 1 < P.n_seg && c >= sseg[si + 1].col0) ++si;  // table in LDS: the index differs per lane
    const FrontSeg& S = sseg[si];
    const int br = c - S.col0;
    const bool blive = c < P.N && br >= 0 && br < S.rows;
    bn[i] = blive ? S.nsrc : 0;
#pragma unroll
    for (int q = 0; q < NSRC; ++q) brow[i][q] = params + S.off[q < S.nsrc ? q : 0] + (int64_t)(blive ? br : 0) * S.ld;
  }
  const int n_stage = (P.K + FBK - 1) / FBK;
  // ---- all global loads of the tile, branch-free (clamped addresses, masked values) ----------------------------
  float4 ra[FSTAGES][2], rb[FSTAGES][2];
#pragma unroll
  for (int s = 0; s < FSTAGES; ++s) {
    const int gk = s * FBK + k4;
    const bool klive = s < n_stage && gk < P.K;
    const int kc = klive ? gk : 0;
    const bool sec = kc + 2 < P.K;
    const bool m1 = klive && gk + 1 < P.K, m2 = klive && gk + 2 < P.K, m3 = klive && gk + 3 < P.K;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float4 a = fr_ld4<VEC>(arow[i] + kc, sec);
      float4 b = fr_ld4<VEC>(brow[i][0] + kc, sec);
#pragma unroll
      for (int q = 1; q < NSRC; ++q) {
        const float4 t = fr_ld4<VEC>(brow[i][q] + kc, sec);
        if (q < bn[i]) { b.x += t.x; b.y += t.y; b.z += t.z; b.w += t.w; }
      }
      const bool bl = bn[i] > 0;
      ra[s][i] = make_float4(klive ? a.x : 0.f, m1 ? a.y : 0.f, m2 ? a.z : 0.f, m3 ? a.w : 0.f);
      rb[s][i] = make_float4(klive && bl ? b.x : 0.f, m1 && bl ? b.y : 0.f, m2 && bl ? b.z : 0.f, m3 && bl ? b.w : 0.f);
    }
  }
  KT(2);
  KTW(3);
  // ---- stages through LDS -----------------------------------------------------------------------------------------
  // [row][k] images, two buffers: stage s + 1 is written while the MFMAs of stage s run (one barrier per stage; written and read
  // in turn -- write, barrier, MFMA, barrier -- a stage cost 2.8 us of which the MFMAs were 1.0).  A thread's four k values of a
  // row are two 8-byte writes; an MFMA operand (row m = lane & 31, k + (lane >> 5)) is a 4-byte read at a constant offset from
  // the lane's row base (pitch 130 floats: rows 16 apart share a bank, a 2-way conflict on the 32 reads of a stage).
  const int kg = w >> 2, sub = w & 3, wm = sub & 1, wn = sub >> 1;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  auto put = [&](int s) {
    float* As = lds + (s & 1) * (2 * FT * FLD);
    float* Bs = As + FT * FLD;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (tid >> 5) + 32 * i;
      float2* pa = reinterpret_cast<float2*>(As + row * FLD + k4);
      float2* pb = reinterpret_cast<float2*>(Bs + row * FLD + k4);
      pa[0] = make_float2(ra[s][i].x, ra[s][i].y); pa[1] = make_float2(ra[s][i].z, ra[s][i].w);
      pb[0] = make_float2(rb[s][i].x, rb[s][i].y); pb[1] = make_float2(rb[s][i].z, rb[s][i].w);
    }
  };
  put(0);
  KT(5);
  __syncthreads();
  KT(6);
#pragma unroll
  for (int s = 0; s < FSTAGES; ++s) {
    if (s >= n_stage) break;  // block-uniform
    if (s + 1 < FSTAGES && s + 1 < n_stage) put(s + 1);
    const float* As = lds + (s & 1) * (2 * FT * FLD);
    const float* Bs = As + FT * FLD;
    const int klen = min(FBK, P.K - s * FBK);
    // k steps of this wave's K-group inside the stage (wave-uniform; a ragged last stage leaves some groups short or idle)
    const int nk = min(max(klen - kg * 32, 0), 32);
    const float* ap = As + (wm * 32 + (lane & 31)) * FLD + kg * 32 + (lane >> 5);
    const float* bp = Bs + (wn * 32 + (lane & 31)) * FLD + kg * 32 + (lane >> 5);
    if (nk == 32) {
      // full group: operand reads in batches of 8 steps ahead of the MFMAs that consume them (a branch per step, as the ragged
      // form below needs, makes every step wait for its own two LDS reads)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float av[8], bv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          av[i] = ap[16 * h + 2 * i];
          bv[i] = bp[16 * h + 2 * i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
      }
    } else {
      for (int kk = 0; kk < nk; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[kk], bp[kk], acc, 0, 0, 0);
    }
    if (s == 0) KT(7);
    if (s == 1) KT(14);
    __syncthreads();
    if (s == 0) KT(13);
    if (s == 1) KT(15);
  }
  KT(4);
  // ---- fixed-order sum over the 4 K-groups; wave (kg, sub) finishes accumulator registers 4 kg .. 4 kg + 3 -----------
  float* red = lds;
#pragma unroll
  for (int i = 0; i < 16; ++i) red[(w * 16 + i) * 64 + lane] = acc[i];
  __syncthreads();
  const int col = n0 + wn * 32 + (lane & 31);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = 4 * kg + j;
    const float v = (red[((0 * 4 + sub) * 16 + i) * 64 + lane] + red[((1 * 4 + sub) * 16 + i) * 64 + lane]) +
                    (red[((2 * 4 + sub) * 16 + i) * 64 + lane] + red[((3 * 4 + sub) * 16 + i) * 64 + lane]);
    // D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int row = m0 + wm * 32 + j + 8 * kg + 4 * (lane >> 5);
    if (row < P.M && col < P.N) P.C[(int64_t)row * P.ldc + col] = v;
  }
}

template <int VEC>
__device__ __forceinline__ void front_gemm_v(const FrontProb& P, const FrontSeg* sseg, const float* params, int tm, int tn, float* lds) {
  switch (P.max_nsrc) {  // block-uniform
    case 1: front_gemm_tile<VEC, 1>(P, sseg, params, tm, tn, lds); break;
    case 2: front_gemm_tile<VEC, 2>(P, sseg, params, tm, tn, lds); break;
    case 3: front_gemm_tile<VEC, 3>(P, sseg, params, tm, tn, lds); break;
    default: front_gemm_tile<VEC, FR_MAX_SRC>(P, sseg, params, tm, tn, lds); break;
  }
}

__global__ __launch_bounds__(1024) void front_kernel(const FrontArgs a) {
  __shared__ __attribute__((aligned(16))) char lds[F_LDS_BYTES];
  const int blk = blockIdx.x;
  if (blk < a.gemm_blocks) {
    KT(0);
    int pi = 0;
    while (pi + 1 < a.n_prob && blk >= a.prob_start[pi + 1]) ++pi;
    karg_warm<(sizeof(FrontProb) + 63) / 64 + 1>((int)offsetof(FrontArgs, prob) + pi * (int)sizeof(FrontProb), (int)sizeof(FrontProb));
    const FrontProb& P = a.prob[pi];
    const int local = blk - P.blk_start;
    // column tiles of one row tile are adjacent block ids (same A rows: L2 / MALL reuse)
    const int tm = local / P.tiles_n, tn = local % P.tiles_n;
    FrontSeg* sseg = reinterpret_cast<FrontSeg*>(lds + F_SEG_OFF);
    if ((int)threadIdx.x < P.n_seg) sseg[threadIdx.x] = P.seg[threadIdx.x];
    __syncthreads();
    if (P.vec == 4) front_gemm_v<4>(P, sseg, a.params, tm, tn, reinterpret_cast<float*>(lds));
    else front_gemm_v<2>(P, sseg, a.params, tm, tn, reinterpret_cast<float*>(lds));
    KT(1);
    return;
  }
  if (blk < a.gemm_blocks + a.plan_blocks) {
    const int pb = blk - a.gemm_blocks;
    int jd = 0;
    while (jd + 1 < 2 * a.n_jobs && pb >= a.part_start[jd + 1]) ++jd;
    const FrontJob& F = a.job[jd >> 1];
    int* wsi = reinterpret_cast<int*>(a.ws);
    PlanJob J;
    J.ei = F.ei; J.E = F.E; J.n_src = F.n_src; J.n_dst = F.n_dst;
    J.rowptr = wsi + F.rowptr; J.col = wsi + F.col; J.eid = wsi + F.eid;
    J.t_rowptr = wsi + F.t_rowptr; J.t_col = wsi + F.t_col; J.t_pos = nullptr;
    J.degf = reinterpret_cast<float*>(wsi + F.degf);
    J.gp_dst = F.gp_dst; J.gp_src = F.gp_src; J.gp_edge = F.gp_edge; J.n_graphs = F.n_graphs;
    J.ell = F.ell ? wsi + F.ell : nullptr;
    J.t_ell = F.t_ell ? wsi + F.t_ell : nullptr;
    J.cnt_in = J.cnt_out = J.cur_in = J.cur_out = nullptr;
    J.tmpc_in = J.tmpc_out = nullptr;
    J.tmp_in = wsi + F.tmp_in; J.tmp_out = wsi + F.tmp_out; J.t_eid = wsi + F.t_eid; J.pos_of_eid = wsi + F.pos_of_eid;
#ifdef HMP_KTIME
    if (threadIdx.x == 0 && pb == 0) front_kt_ei = a.job[0].ei;
#endif
    const bool last = pb + 1 == a.part_start[jd + 1];
    if (a.plan_rc) plan_small_part<true>(J, jd & 1, pb - a.part_start[jd], a.rows_per_part[jd], last, a.need_tpos, a.status, lds);
    else plan_small_part<false>(J, jd & 1, pb - a.part_start[jd], a.rows_per_part[jd], last, a.need_tpos, a.status, lds);
    return;
  }
  // pack role: 16 items per block; the block -> (segment, first item) map is a static device table
  const int kb = blk - a.gemm_blocks - a.plan_blocks;
  if (a.step_ctr && kb == 0 && threadIdx.x == 0) {
    const int t = *a.step_ctr + 1;
    *a.step_ctr = t;
    if (a.step_mirror) *a.step_mirror = t;
  }
  const int2 m = a.pack_map[kb];
  pack_item(a.segs[m.x], m.y + (int)(threadIdx.x >> 6), a.params, a.packed);
}

int front_launch(FrontArgs& a, hipStream_t st) {
  int start = 0;
  for (int i = 0; i < a.n_prob; ++i) {
    FrontProb& P = a.prob[i];
    HMP_CHECK_ARG(P.K >= 4 && P.K <= FSTAGES * FBK && (P.K & 1) == 0 && P.M > 0 && P.N > 0, "front: projection K %d unsupported", P.K);
    HMP_CHECK_ARG(P.n_seg >= 1 && P.n_seg <= FR_MAX_SEG, "front: %d segments", P.n_seg);
    const uintptr_t pa = reinterpret_cast<uintptr_t>(P.A), pp = reinterpret_cast<uintptr_t>(a.params);
    HMP_CHECK_ARG((pa & 7) == 0 && (P.lda & 1) == 0 && (pp & 15) == 0, "front: operands must be 8-byte aligned with even leading dimensions");
    int vec = ((pa & 15) == 0 && (P.lda & 3) == 0) ? 4 : 2;
    P.max_nsrc = 1;
    for (int s = 0; s < P.n_seg; ++s) {
      const FrontSeg& S = P.seg[s];
      HMP_CHECK_ARG(S.nsrc >= 1 && S.nsrc <= FR_MAX_SRC && (S.ld & 1) == 0, "front: segment with %d sources / ld %d", S.nsrc, S.ld);
      P.max_nsrc = S.nsrc > P.max_nsrc ? S.nsrc : P.max_nsrc;
      if (S.ld & 3) vec = 2;
      for (int q = 0; q < S.nsrc; ++q) {
        HMP_CHECK_ARG((S.off[q] & 1) == 0, "front: parameter offset not 8-byte aligned");
        if (S.off[q] & 3) vec = 2;
      }
    }
    P.vec = vec;
    P.tiles_m = cdiv(P.M, FT);
    P.tiles_n = cdiv(P.N, FT);
    P.blk_start = start;
    a.prob_start[i] = start;
    start += P.tiles_m * P.tiles_n;
  }
  a.gemm_blocks = start;
  const int total = a.gemm_blocks + a.plan_blocks + a.pack_blocks;
  if (total == 0) return HMP_OK;
  hipLaunchKernelGGL(front_kernel, dim3(total), dim3(1024), 0, st, a);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

}  // namespace hmp
