// Backward GEMMs of the 10^6-node regime (BASELINE config 5, bf16 compute mode), round 3.  Both are operand-stationary forms of
// what the tiled kernel of gemm_bf16.hip computes, built with what made the forward weight-stationary kernel work: operands go
// memory -> LDS by global_load_lds_dwordx4 (no VGPR staging, no ds_write), 16-byte chunks XOR-swizzled by the row through the
// SOURCE address so that every operand fetch is conflict-free, persistent workgroups that keep one operand in registers.
// The tiled kernel ran them with 58-64 % of its LDS cycles lost to bank conflicts and the matrix pipe 10-20 % busy
// (profiles/r02_e_pmc_cfg5_mfma_lds.json).
//
//   gemm_bf16_dx_kernel   input gradient   G[M, N] = act'(H) . (dZ[M, K] * Wp[K, N])      (reference: the backward of
//                         SAGEConv.lin_l / lin_r, [PyG] sage_conv.py, reached from models/utils.py:14)
//                         M = 10^6 rows, K = 256 / 512 / 768 stacked columns (dZ in one or two bf16 pieces: its root block is the
//                         output gradient itself), N = 256.  WEIGHT-stationary: a workgroup (4 waves, two per CU) owns 128 output
//                         columns; its K x 128 slice of Wp sits in registers as MFMA operands (K / 4 VGPRs per wave); dZ streams 64
//                         rows x 256 columns at a time through two 32 KB LDS buffers; the product is computed transposed so that a
//                         lane holds 4 consecutive output columns of one row; the activation / dropout mask comes from H through a
//                         wave-private LDS tile, the bf16 result leaves through LDS as whole 64-byte row segments.
//   gemm_bf16_dw_kernel   weight gradient  dWp[Mw, F + 1] = dZ[nodes, Mw]^T * [H | 1][nodes, F]   (split over node ranges: slabs)
//                         OUTPUT-stationary: a workgroup (8 waves, one per CU) owns a 256 x 256 tile of dWp in registers (128
//                         accumulator VGPRs per wave) for a contiguous range of nodes; both operands are node-major, i.e. strided
//                         along the MFMA's k: they keep their natural [node][column] image in LDS (two sub-images of 256-byte rows,
//                         chunk ^= ((row & 3) << 2) | ((row >> 2) & 3)) and are transposed on the way out by ds_read_b64_tr_b16,
//                         conflict-free on that image.  The bias column (row sums of dZ^T) is one extra MFMA per wave and k step
//                         against an all-ones operand.  An fp32 H (the input features of layer 0) is converted in registers and
//                         written into the same image (8-byte ds_writes) one chunk ahead.
#include "kernels.h"

namespace hmp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bwd_act_mask(float h, int act, bool keep, float scale) {
  if (!keep) return 0.f;
  if (act == HMP_ACT_RELU) return h > 0.f ? scale : 0.f;
  if (act == HMP_ACT_ELU) return h > 0.f ? scale : (h + scale);
  return scale;
}

// ---------------------------------------------------------------------------------------------------------------------------
// input gradient
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int DX_ROWS = 64, DX_COLS = 128, DX_THREADS = 256;
constexpr int DX_UNIT = DX_ROWS * 256;            // pipeline unit: 64 rows x 128 columns of dZ (bf16) = 16 KB
constexpr int DX_HTILE = DX_ROWS * DX_COLS * 2;   // the H tile (4 KB per wave)
constexpr int dx_lds_bytes(int nbuf) { return nbuf * DX_UNIT + DX_HTILE; }

struct DxArgs {
  const uint16_t* A;   // dZ, bf16 [M][lda]: columns [0, a_split) (all of them when a_split == 0)
  const uint16_t* A2;  // columns [a_split, K): A2[m * lda2 + (k - a_split)]
  int lda, lda2, a_split;
  const float* W;      // Wp fp32 [K][ldb]
  int ldb;
  uint16_t* C;         // bf16 [M][ldc]
  int ldc;
  const uint16_t* H;   // bf16 [M][ldh] or null (no mask)
  int ldh, act, drop_on;
  float dscale;
  // gather-add (aggregate-first convs): C += sum over the CSC entries k of row m of gadd[t_col[k]][.] * gdeg[t_col[k]], before the mask
  const int* g_rowptr;  // null: none
  const int* g_col;
  const float* g_deg;
  const float* g_rows;  // fp32 [.][g_ld]
  int g_ld;
  int M, N, K;
  int n_slices, groups, tiles_per_group, n_tiles;
};

#define HMP_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
// LDS accesses the compiler must not see: next to a pending LDS-DMA (global_load_lds) hipcc orders every LDS access it cannot prove
// disjoint from the DMA's destination -- and every intrinsic LDS read (ds_read_b64_tr_b16 has no memory operand) -- behind
// s_waitcnt vmcnt(0), which drains the whole request pipeline.  These go through inline asm; the kernel waits by hand (counted
// vmcnt for the DMA, lgkmcnt for the reads, followed by a sched_barrier: the compiler may otherwise hoist the consumers).
typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lds_ptr_t)p; }
__device__ __forceinline__ uint2 lds_read_b64(uint32_t addr) {
  uint2 r;
  asm volatile("ds_read_b64 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}
__device__ __forceinline__ void lds_write_b64(uint32_t addr, uint2 v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_wait_all() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// workgroup barrier that does NOT drain the vector-memory counter (__syncthreads() waits vmcnt(0) while an LDS-DMA is in flight):
// what crosses waves here is DMA data, ordered by each wave's own counted vmcnt wait in front of the barrier
__device__ __forceinline__ void bwd_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// KU = K / 128 pipeline units per 64-row tile; NBUF unit buffers, NBUF - 1 units requested ahead (the wait in front of unit v leaves
// the NBUF - 2 younger units in flight).  NBUF = 4: two workgroups per CU (80 KB of LDS, 256 registers); K = 768 needs 192 registers
// of Wp per wave and runs ONE workgroup per CU with the whole register file and NBUF = 8.
template <int KU, int NBUF>
__global__ __launch_bounds__(DX_THREADS, (NBUF > 4 ? 1 : 2)) void gemm_bf16_dx_kernel(const DxArgs a) {
  constexpr int D = NBUF - 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char dx_lds[];
  unsigned char* const h_base = dx_lds + NBUF * DX_UNIT;
  const int b = blockIdx.x;
  // blocks b, b + 8, .. share an XCD: they take the column slices of the same row range, so dZ leaves HBM once
  const int slice = (b >> 3) % a.n_slices;
  const int group = (b & 7) + 8 * (b / (8 * a.n_slices));
  const int n0 = slice * DX_COLS;
  const int t_begin = group * a.tiles_per_group;
  const int t_end = min(t_begin + a.tiles_per_group, a.n_tiles);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int l31 = lane & 31, half = lane >> 5;

  // ---- this wave's 32 output columns of Wp as MFMA operands of the TRANSPOSED product D^T = Wp^T_tile * dZ_tile^T:
  // wreg[s] = Wp[16 s + 8 half .. + 8][n0 + 32 w + l31]   (32 lanes read 128 contiguous bytes per k)
  bf16x8 wreg[KU * 8];
  {
    const float* wcol = a.W + n0 + 32 * w + l31;
#pragma unroll
    for (int s = 0; s < KU * 8; ++s) {
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (__bf16)wcol[(int64_t)(16 * s + 8 * half + j) * a.ldb];
      wreg[s] = v;
      // 2 k steps (16 loads, each with its own 64-bit address) per round trip: without the fence all K x 8 loads are hoisted ahead
      // of the first conversion and the prologue alone needs more registers than the main loop (the empty asm pins the converted
      // operand here and, as a memory clobber, keeps the later loads behind it)
      if ((s & 1) == 1) {
        asm volatile("" : "+v"(wreg[s - 1]), "+v"(wreg[s]) : : "memory");
      }
    }
  }
  if (t_begin >= t_end) return;  // block-uniform
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the counted waits below count LDS-DMA requests only

  // ---- unit v = (tile t_begin + v / KU, columns 128 (v % KU) ..) -> buffer v % NBUF.  Wave w requests rows 16 w .. 16 w + 15, four
  // rows (16 chunks of 16 bytes each) per instruction; the lane at position `pos` of a row requests chunk pos ^ (row & 15): the
  // swizzle lives in the source address, the LDS image stays lane-linear
  const int total = (t_end - t_begin) * KU;
  const int rlo = lane >> 4, pos = lane & 15;
  auto request = [&](int v) {
    const int t = t_begin + v / KU, p = v % KU, buf = v % NBUF;
    const int m0 = t * DX_ROWS, kb = 128 * p;
    const bool second = a.a_split > 0 && kb >= a.a_split;  // block-uniform
    // (kept in scalar registers: as a per-lane select the two 64-bit bases were spilled, and a scratch reload in this loop is a
    // vector-memory operation whose vmcnt(0) drains the whole request pipeline)
    const uint64_t bb = reinterpret_cast<uint64_t>(second ? a.A2 + (kb - a.a_split) : a.A + kb);
    // (the builtin returns a SIGNED int: widen through uint32_t, or a low half with bit 31 set sign-extends into the high half)
    const uint32_t bb_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(bb >> 32));
    const uint32_t bb_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)bb);
    const uint16_t* base = reinterpret_cast<const uint16_t*>(((uint64_t)bb_hi << 32) | (uint64_t)bb_lo);
    const int ld = __builtin_amdgcn_readfirstlane(second ? a.lda2 : a.lda);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 16 * w + 4 * i + rlo;
      const int64_t grow = min(m0 + r, a.M - 1);  // rows past M: a valid address, the products are never stored
      const uint16_t* src = base + grow * ld + 8 * (pos ^ (r & 15));
      unsigned char* dst = dx_lds + buf * DX_UNIT + w * 4096 + i * 1024;  // wave-uniform; lane l lands at + 16 l
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };
  // the wave's own 64 x 32 piece of H: four requests of 16 rows x 64 bytes; image [q][chunk][row & 15][16 bytes]
  auto request_h = [&](int t) {
    const int m0 = t * DX_ROWS;
    const int ch = lane >> 4, rr = lane & 15;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t grow = min(m0 + 16 * q + rr, a.M - 1);
      const uint16_t* src = a.H + grow * a.ldh + n0 + 32 * w + 8 * ch;
      unsigned char* dst = h_base + w * 4096 + q * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  for (int v = 0; v < D && v < total; ++v) request(v);
  f32x16 acc[2];
  int v = 0;
  for (int t = t_begin; t < t_end; ++t) {
#pragma unroll
    for (int p = 0; p < KU; ++p, ++v) {
      // this wave's share of unit v has landed once at most the D - 1 younger units are outstanding (anything else issued since --
      // H, the previous tile's stores -- only makes the count stricter); at the tail, where fewer units follow, wait for everything
      if (v + D <= total) HMP_VMCNT(4 * (D - 1));
      else HMP_VMCNT(0);
      bwd_barrier();  // every wave's share landed; every wave left its reads of unit v - 1
      if (v + D < total) request(v + D);  // into the buffer of unit v - 1
      if (p == 0) {
        if (a.H) request_h(t);  // wave-private tile: the previous tile's reads of it were waited for (lgkmcnt) in its epilogue
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      }
      const unsigned char* Ab = dx_lds + (v % NBUF) * DX_UNIT;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        bf16x8 av[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const bf16x8*>(Ab + (i * 32 + l31) * 256 + (((2 * s + half) ^ (l31 & 15)) << 4));
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[p * 8 + s], av[i], acc[i], 0, 0, 0);
      }
    }
    // ---- epilogue.  D^T: lane = output row m0 + i*32 + l31; registers 4 g .. 4 g + 3 = columns n0 + 32 w + 8 g + 4 half ..
    // H was requested in front of this tile's first unit: younger than it are the KU - 1 unit requests issued since
    if (a.H) HMP_VMCNT(4 * (KU - 1));
    const int m0 = t * DX_ROWS;
    // The wave's 4 KB of the H tile doubles as the staging area of its output: it is the one region no other wave touches (the
    // unit buffers are read as MFMA operands by every wave until the next barrier).  All H values are read into registers first;
    // a wave's LDS operations execute in order, so the staging writes below cannot overtake those reads.
    unsigned char* stg = h_base + w * 4096;
    const unsigned char* hq = h_base + w * 4096;
    uint2 hball[2][4];
    if (a.H) {  // block-uniform
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int m = i * 32 + l31;
          hball[i][g] = lds_read_b64(lds_addr(hq + (m >> 4) * 1024 + ((g * 16 + (m & 15)) << 4) + 8 * half));
        }
      lds_wait_all();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = i * 32 + l31;
      float vv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) vv[r] = acc[i][r];
      if (a.g_rowptr) {  // block-uniform: gather-add of the aggregate-first convs' gradient rows (a lane = one output row)
        const int row = min(m0 + m, a.M - 1);
        const int kb = a.g_rowptr[row], ke = a.g_rowptr[row + 1];
        for (int k = kb; k < ke; ++k) {
          const int j = a.g_col[k];
          const float wd = a.g_deg[j];
          const float* gr = a.g_rows + (int64_t)j * a.g_ld + n0 + 32 * w + 4 * half;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4 x = *reinterpret_cast<const float4*>(gr + 8 * g);
            vv[4 * g + 0] += wd * x.x; vv[4 * g + 1] += wd * x.y; vv[4 * g + 2] += wd * x.z; vv[4 * g + 3] += wd * x.w;
          }
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (a.H) {  // block-uniform
          const uint2 hb = hball[i][g];
          const uint32_t hw[4] = {hb.x << 16, hb.x & 0xffff0000u, hb.y << 16, hb.y & 0xffff0000u};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool keep = !a.drop_on || hw[e] != 0x80000000u;  // dropped elements were stored as -0
            vv[4 * g + e] *= bwd_act_mask(__uint_as_float(hw[e]), a.act, keep, a.dscale);
          }
        }
        union { bf16x4 b; uint2 u; } o;
        o.b[0] = (__bf16)vv[4 * g + 0]; o.b[1] = (__bf16)vv[4 * g + 1]; o.b[2] = (__bf16)vv[4 * g + 2]; o.b[3] = (__bf16)vv[4 * g + 3];
        const int u8 = 2 * g + half;  // 8-byte unit of the wave's 64-byte row segment
        lds_write_b64(lds_addr(stg + m * 64 + ((u8 ^ (m & 7)) << 3)), o.u);
      }
    }
    lds_wait_all();  // this wave's own writes
    const int q4 = lane & 3;
    uint2 slo[4], shi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = k * 16 + (lane >> 2);
      slo[k] = lds_read_b64(lds_addr(stg + r * 64 + (((2 * q4) ^ (r & 7)) << 3)));
      shi[k] = lds_read_b64(lds_addr(stg + r * 64 + (((2 * q4 + 1) ^ (r & 7)) << 3)));
    }
    lds_wait_all();  // staged / H reads done before this wave's next requests reuse the area
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int row = m0 + k * 16 + (lane >> 2);
      if (row < a.M) *reinterpret_cast<uint4*>(a.C + (int64_t)row * a.ldc + n0 + 32 * w + 8 * q4) = make_uint4(slo[k].x, slo[k].y, shi[k].x, shi[k].y);
    }
  }
}

static bool env_off(const char* name) {
  const char* v = getenv(name);
  return v && v[0] == '0';
}

bool gemm_bf16_dx_takes(const GemmProblem& p, bool want_split) {
  if (env_off("HMP_GEMM_DX")) return false;  // 0: the tiled kernel (tests compare the two)
  // K = 768 (one workgroup per CU, one wave per SIMD, every LDS read waited for in front of its MFMA) measured 903 us against the
  // tiled kernel's 796 us at 10^6 rows: only on request (HMP_GEMM_DX=2, the unit test); with the objects -> rooms conv evaluated
  // aggregate-first the stacked operand of config 5 is 512 columns
  if (p.K > 512) {
    const char* v = getenv("HMP_GEMM_DX");
    if (!(v && v[0] == '2')) return false;
  }
  return !want_split && !p.trans_a && !p.trans_b && p.a_bf16 && p.c_bf16 && !p.b_bf16 && !p.aug_ones &&
         (p.epi == EPI_NONE || p.h_bf16) && (p.K == 256 || p.K == 512 || p.K == 768) && (p.N % DX_COLS) == 0 && p.N >= DX_COLS &&
         p.M >= 32768 && (p.lda & 7) == 0 && (p.a_split == 0 || ((p.a_split & 255) == 0 && (p.lda2 & 7) == 0 && p.A2 &&
         (reinterpret_cast<uintptr_t>(p.A2) & 15) == 0)) && (p.ldc & 7) == 0 && (p.epi == EPI_NONE || (p.ldh & 7) == 0) &&
         (reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 &&
         (p.epi == EPI_NONE || (reinterpret_cast<uintptr_t>(p.H) & 15) == 0) &&
         (!p.g_rowptr || ((p.g_ld & 3) == 0 && (reinterpret_cast<uintptr_t>(p.g_rows) & 15) == 0));
}

static int n_cus() {
  static const int n = [] {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) return prop.multiProcessorCount;
    return 256;
  }();
  return n;
}

int gemm_bf16_dx_launch(const GemmProblem& p, hipStream_t st) {
  DxArgs a;
  memset(&a, 0, sizeof(a));
  a.A = reinterpret_cast<const uint16_t*>(p.A); a.A2 = reinterpret_cast<const uint16_t*>(p.A2);
  a.lda = p.lda; a.lda2 = p.lda2; a.a_split = p.a_split;
  a.W = p.B; a.ldb = p.ldb;
  a.C = reinterpret_cast<uint16_t*>(p.C); a.ldc = p.ldc;
  a.H = p.epi == EPI_ACTMASK ? reinterpret_cast<const uint16_t*>(p.H) : nullptr;
  a.ldh = p.ldh; a.act = p.act; a.drop_on = p.drop_on;
  a.dscale = p.drop_on ? p.drop.scale : 1.f;
  a.g_rowptr = p.g_rowptr; a.g_col = p.g_col; a.g_deg = p.g_deg; a.g_rows = p.g_rows; a.g_ld = p.g_ld;
  a.M = p.M; a.N = p.N; a.K = p.K;
  a.n_slices = p.N / DX_COLS;
  a.n_tiles = cdiv(p.M, DX_ROWS);
  const int per_cu = p.K > 512 ? 1 : 2;
  int sets = per_cu * n_cus() / (8 * a.n_slices);
  if (sets < 1) sets = 1;
  a.groups = 8 * sets;
  if (a.groups > a.n_tiles) a.groups = ((a.n_tiles + 7) / 8) * 8;
  a.tiles_per_group = cdiv(a.n_tiles, a.groups);
  const dim3 grid(a.groups * a.n_slices), block(DX_THREADS);
  static const int attr_rc = [] {
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dx_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, dx_lds_bytes(4));
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dx_kernel<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, dx_lds_bytes(4));
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dx_kernel<6, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, dx_lds_bytes(8));
    return rc;
  }();
  HMP_CHECK_ARG(attr_rc == 0, "gemm_bf16_dx: LDS size refused");
  if (p.K == 256) hipLaunchKernelGGL((gemm_bf16_dx_kernel<2, 4>), grid, block, dx_lds_bytes(4), st, a);
  else if (p.K == 512) hipLaunchKernelGGL((gemm_bf16_dx_kernel<4, 4>), grid, block, dx_lds_bytes(4), st, a);
  else hipLaunchKernelGGL((gemm_bf16_dx_kernel<6, 8>), grid, block, dx_lds_bytes(8), st, a);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int DW_TILE = 256, DW_THREADS = 512;
constexpr int dw_lds_bytes(int un, int nbuf) { return nbuf * 2 * un * 512; }  // per unit: A image + B image, [2 sub-images][un nodes][256 bytes] each

struct DwArgs {
  const uint16_t* A;   // dZ bf16 [nodes][lda]: columns [0, a_split) (all when a_split == 0)
  const uint16_t* A2;  // columns [a_split, Mw)
  int lda, lda2, a_split;
  const void* B;       // H: bf16 [nodes][ldb] (b_bf16) or fp32 [nodes][ldb]; 256 columns
  int ldb, b_bf16;
  float* C;            // slabs: slab z = C + z * slab_stride, fp32 [Mw][ldc], column n_real = row sums when aug_ones
  int ldc, n_real, aug_ones;
  int64_t slab_stride;
  int K;               // nodes
  int tiles_m, groups, nodes_per_group;
};

__device__ __forceinline__ s16x4 lds_read_tr16_b64(uint32_t addr) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}
// byte offset of 16-byte chunk ch (0..15) of row `row` inside a sub-image of 256-byte rows
__device__ __forceinline__ int dw_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// UN nodes per pipeline unit, NBUF unit buffers, NBUF - 1 units requested ahead.
//   bf16 H: <false, 32, 4> -- both operands by LDS-DMA, three units (96 KB) in flight per CU;
//   fp32 H: <true, 64, 2>  -- dZ by LDS-DMA, H through registers (converted, 8-byte ds_writes) one unit ahead.
template <bool BF32, int UN, int NBUF>
__global__ __launch_bounds__(DW_THREADS, 2) void gemm_bf16_dw_kernel(const DwArgs a) {
  constexpr int D = NBUF - 1;
  constexpr int IMG = UN * 512, UNIT = 2 * IMG, SUB = UN * 256;
  constexpr int RQ = UN / 16;                 // LDS-DMA requests per wave, operand and unit
  constexpr int RQ_ALL = BF32 ? RQ : 2 * RQ;  // ... per wave and unit
  constexpr int NB = BF32 ? UN / 8 : 1;       // float4 registers per thread and unit of an fp32 H
  static_assert(!BF32 || D == 1, "the register-staged operand is written one unit ahead");
  extern __shared__ __attribute__((aligned(16))) unsigned char dw_lds[];
  const int b = blockIdx.x;
  // the tiles of one node range run on one XCD (block ids 8 apart): H leaves HBM once
  const int idx = b >> 3;
  const int tile = idx % a.tiles_m;
  const int group = (b & 7) + 8 * (idx / a.tiles_m);
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wm = w >> 2, wn = w & 3;
  const int n_begin = (int)min((int64_t)group * a.nodes_per_group, (int64_t)a.K);
  const int n_end = (int)min((int64_t)n_begin + a.nodes_per_group, (int64_t)a.K);
  const int units = (n_end - n_begin + UN - 1) / UN;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x16 acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc1[r] = 0.f;
  bf16x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;

  // ---- unit v -> buffer v % NBUF: UN / 2 requests per bf16 operand (4 rows x 256 bytes of one sub-image each), RQ per wave
  const int lrow = lane >> 4, pch = lane & 15;
  auto request = [&](int v) {
    const int nb = n_begin + v * UN;
    unsigned char* ub = dw_lds + (v % NBUF) * UNIT;
#pragma unroll
    for (int ii = 0; ii < RQ; ++ii) {
      const int id = w * RQ + ii;
      const int h2 = id / (UN / 4), rq = id % (UN / 4);
      const int row = 4 * rq + lrow;
      const int ch = pch ^ (((row & 3) << 2) | ((row >> 2) & 3));
      const int64_t node = min(nb + row, a.K - 1);  // rows past the range: a valid address; the B operand is zeroed there
      {
        const int colg = tile * DW_TILE + 128 * h2;
        const bool second = a.a_split > 0 && colg >= a.a_split;  // wave-uniform
        const uint16_t* src = (second ? a.A2 + node * a.lda2 + (colg - a.a_split) : a.A + node * a.lda + colg) + 8 * ch;
        unsigned char* dst = ub + h2 * SUB + rq * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
      if constexpr (!BF32) {
        const uint16_t* src = reinterpret_cast<const uint16_t*>(a.B) + node * a.ldb + 128 * h2 + 8 * ch;
        unsigned char* dst = ub + IMG + h2 * SUB + rq * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
  };
  // fp32 H: NB float4 per thread and unit, issued one unit ahead, converted and written behind the MFMAs of the current one
  float4 breg[NB];
  auto load_b = [&](int v) {
    if constexpr (BF32) {
      const int nb = n_begin + v * UN;
#pragma unroll
      for (int e = 0; e < NB; ++e) {
        const int q = (int)threadIdx.x + DW_THREADS * e;
        const int row = q >> 6, c4 = q & 63;
        const int64_t node = min(nb + row, a.K - 1);
        breg[e] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.B) + node * a.ldb + 4 * c4);
      }
    }
  };
  auto store_b = [&](int v) {
    if constexpr (BF32) {
      unsigned char* ub = dw_lds + (v % NBUF) * UNIT + IMG;
#pragma unroll
      for (int e = 0; e < NB; ++e) {
        const int q = (int)threadIdx.x + DW_THREADS * e;
        const int row = q >> 6, c4 = q & 63;
        const int h2 = c4 >> 5, c4l = c4 & 31;
        bf16x4 o;
        o[0] = (__bf16)breg[e].x; o[1] = (__bf16)breg[e].y; o[2] = (__bf16)breg[e].z; o[3] = (__bf16)breg[e].w;
        *reinterpret_cast<bf16x4*>(ub + h2 * SUB + dw_off(row, c4l >> 1) + 8 * (c4l & 1)) = o;
      }
    }
  };

  for (int v = 0; v < D && v < units; ++v) request(v);
  if (BF32 && units > 0) {
    load_b(0);
    store_b(0);
  }
  // transposed operand reads: 16-lane group gi = lane >> 4 reads the block of 4 nodes x 16 columns; lane 4 q + p supplies node
  // row q, columns 4 p .. 4 p + 3 and receives column (lane & 15) of the 4 nodes
  const int gi = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  for (int v = 0; v < units; ++v) {
    // this wave's share of unit v has landed once at most the D - 1 younger units are outstanding; fewer follow at the tail
    if (D > 1 && v + D <= units) HMP_VMCNT(RQ_ALL * (D - 1));
    else HMP_VMCNT(0);
    bwd_barrier();  // every wave's share (and its ds_writes of an fp32 H) landed; every wave left its reads of unit v - 1
    if (v + D < units) {
      request(v + D);  // into the buffer of unit v - 1
      load_b(v + D);
    }
    const unsigned char* Ai = dw_lds + (v % NBUF) * UNIT + wm * SUB;
    const unsigned char* Bi = dw_lds + (v % NBUF) * UNIT + IMG + (wn >> 1) * SUB;
    const int valid = n_end - (n_begin + v * UN);  // nodes of this unit that exist (>= UN: all)
#pragma unroll
    for (int ks = 0; ks < UN / 16; ++ks) {
      const int r0 = 16 * ks + 8 * (gi >> 1);
      bf16x8 av[4], bv[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c0 = 4 * i + 2 * (gi & 1) + (pp >> 1);
        union { s16x4 h[2]; bf16x8 v; } uu;
        uu.h[0] = lds_read_tr16_b64(lds_addr(Ai + dw_off(r0 + q, c0) + 8 * (pp & 1)));
        uu.h[1] = lds_read_tr16_b64(lds_addr(Ai + dw_off(r0 + 4 + q, c0) + 8 * (pp & 1)));
        av[i] = uu.v;
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c0 = 8 * (wn & 1) + 4 * j + 2 * (gi & 1) + (pp >> 1);
        union { s16x4 h[2]; bf16x8 v; } uu;
        uu.h[0] = lds_read_tr16_b64(lds_addr(Bi + dw_off(r0 + q, c0) + 8 * (pp & 1)));
        uu.h[1] = lds_read_tr16_b64(lds_addr(Bi + dw_off(r0 + 4 + q, c0) + 8 * (pp & 1)));
        bv[j] = uu.v;
      }
      lds_wait_all();
      bf16x8 on = ones;
      if (valid < UN) {  // block-uniform: last unit of the range -- a lane's element e is node 16 ks + 8 (lane >> 5) + e
        const int kb = 16 * ks + 8 * (lane >> 5);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool live = kb + e < valid;
          bv[0][e] = live ? bv[0][e] : (__bf16)0.f;
          bv[1][e] = live ? bv[1][e] : (__bf16)0.f;
          on[e] = live ? on[e] : (__bf16)0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
      // row sums (bias gradient): row tile i = wn of this wave's half -> every wave runs ONE extra MFMA per k step
      if (a.aug_ones) {  // block-uniform.  The operand is fetched again by ADDRESS (row tile wn): selecting among av[0..3] by a
                         // run-time index sends the array to scratch
        const int c0 = 4 * wn + 2 * (gi & 1) + (pp >> 1);
        union { s16x4 h[2]; bf16x8 v; } uu;
        uu.h[0] = lds_read_tr16_b64(lds_addr(Ai + dw_off(r0 + q, c0) + 8 * (pp & 1)));
        uu.h[1] = lds_read_tr16_b64(lds_addr(Ai + dw_off(r0 + 4 + q, c0) + 8 * (pp & 1)));
        lds_wait_all();
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(uu.v, on, acc1, 0, 0, 0);
      }
    }
    if (BF32 && v + 1 < units) store_b(v + 1);  // D == 1: the buffer of unit v - 1, free since the barrier above
  }
  // ---- slab of this node range.  D: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  float* Cz = a.C + (int64_t)group * a.slab_stride;
  const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = tile * DW_TILE + 128 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
        Cz[(int64_t)row * a.ldc + 64 * wn + 32 * j + l31] = acc[i][j][r];
      }
  if (a.aug_ones && l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = tile * DW_TILE + 128 * wm + 32 * wn + (r & 3) + 8 * (r >> 2) + 4 * half;
      Cz[(int64_t)row * a.ldc + a.n_real] = acc1[r];
    }
  }
}

bool gemm_bf16_dw_takes(const GemmProblem& p, bool want_split) {
  if (env_off("HMP_GEMM_DW")) return false;  // 0: the tiled kernel (tests compare the two)
  return want_split && p.trans_a && !p.trans_b && p.a_bf16 && (p.M % DW_TILE) == 0 && p.M >= DW_TILE && p.n_real == 256 &&
         p.N == p.n_real + (p.aug_ones ? 1 : 0) && p.K >= 65536 && (p.lda & 7) == 0 && (reinterpret_cast<uintptr_t>(p.A) & 15) == 0 &&
         (p.a_split == 0 || ((p.a_split & 255) == 0 && (p.lda2 & 7) == 0 && p.A2 && (reinterpret_cast<uintptr_t>(p.A2) & 15) == 0)) &&
         (reinterpret_cast<uintptr_t>(p.B) & 15) == 0 && (p.ldb & (p.b_bf16 ? 7 : 3)) == 0 && p.ldc >= p.N && !p.c_bf16;
}

// *n_slabs: the number of slabs written (one per node range)
int gemm_bf16_dw_launch(const GemmProblem& p, int max_slabs, int* n_slabs, hipStream_t st) {
  DwArgs a;
  memset(&a, 0, sizeof(a));
  a.A = reinterpret_cast<const uint16_t*>(p.A); a.A2 = reinterpret_cast<const uint16_t*>(p.A2);
  a.lda = p.lda; a.lda2 = p.lda2; a.a_split = p.a_split;
  a.B = p.B; a.ldb = p.ldb; a.b_bf16 = p.b_bf16;
  a.C = p.C; a.ldc = p.ldc; a.n_real = p.n_real; a.aug_ones = p.aug_ones; a.slab_stride = p.slab_stride;
  a.K = p.K;
  a.tiles_m = p.M / DW_TILE;
  int groups = 8 * (n_cus() / (8 * a.tiles_m));  // one workgroup per CU, node ranges in sets of 8 (one per XCD)
  if (groups < 8) groups = 8;
  while (groups > 8 && groups > max_slabs) groups -= 8;
  HMP_CHECK_ARG(groups <= max_slabs, "gemm_bf16_dw: %d slabs needed, %d available", groups, max_slabs);
  a.groups = groups;
  a.nodes_per_group = cdiv(cdiv(p.K, groups), 64) * 64;
  *n_slabs = groups;
  const dim3 grid(groups * a.tiles_m), block(DW_THREADS);
  static const int attr_rc = [] {
    int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dw_kernel<false, 32, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, dw_lds_bytes(32, 4));
    rc |= (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_dw_kernel<true, 64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, dw_lds_bytes(64, 2));
    return rc;
  }();
  HMP_CHECK_ARG(attr_rc == 0, "gemm_bf16_dw: LDS size refused");
  if (p.b_bf16) hipLaunchKernelGGL((gemm_bf16_dw_kernel<false, 32, 4>), grid, block, dw_lds_bytes(32, 4), st, a);
  else hipLaunchKernelGGL((gemm_bf16_dw_kernel<true, 64, 2>), grid, block, dw_lds_bytes(64, 2), st, a);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

}  // namespace hmp

// ---- unit-test entries (include/hydra_mp.h section 3): the same GemmProblem the executor builds, through gemm_bf16_launch, so
// that HMP_GEMM_DX=0 / HMP_GEMM_DW=0 send the identical call to the tiled kernel
extern "C" int hmp_gemm_bf16_dx(const uint16_t* d_dz, int32_t lddz, const uint16_t* d_dz2, int32_t lddz2, int32_t split, const float* d_w,
                                int32_t ldw, const uint16_t* d_h, int32_t ldh, int32_t act, int32_t drop_on, float drop_scale,
                                uint16_t* d_g, int32_t ldg, int32_t M, int32_t N, int32_t K, void* stream) {
  using namespace hmp;
  HMP_CHECK_ARG(d_dz && d_w && d_g && M >= 0 && N > 0 && K > 0, "hmp_gemm_bf16_dx: bad argument");
  HMP_CHECK_ARG(split == 0 || (d_dz2 && split > 0 && split < K && (split & 255) == 0), "hmp_gemm_bf16_dx: split %d of K = %d", split, K);
  HMP_CHECK_ARG(lddz >= (split ? split : K) && (split == 0 || lddz2 >= K - split) && ldw >= N && ldg >= N && (!d_h || ldh >= N), "hmp_gemm_bf16_dx: leading dimension too small");
  GemmBatch gb;
  memset(&gb, 0, sizeof(gb));
  gb.n = 1;
  GemmProblem& p = gb.p[0];
  p.A = reinterpret_cast<const float*>(d_dz); p.lda = lddz; p.a_bf16 = 1;
  if (split) { p.A2 = reinterpret_cast<const float*>(d_dz2); p.lda2 = lddz2; p.a_split = split; }
  p.B = d_w; p.ldb = ldw;
  p.C = reinterpret_cast<float*>(d_g); p.ldc = ldg; p.c_bf16 = 1;
  p.M = M; p.N = N; p.K = K; p.n_real = N;
  p.epi = EPI_NONE;
  if (d_h) {
    p.epi = EPI_ACTMASK;
    p.H = reinterpret_cast<const float*>(d_h); p.ldh = ldh; p.h_bf16 = 1; p.act = act; p.drop_on = drop_on; p.drop.scale = drop_scale;
  }
  if (M == 0) return HMP_OK;
  return gemm_bf16_launch(gb, false, 1, (hipStream_t)stream);
}

extern "C" int hmp_gemm_bf16_dw(const uint16_t* d_dz, int32_t lddz, const uint16_t* d_dz2, int32_t lddz2, int32_t split, const void* d_h,
                                int32_t ldh, int32_t h_bf16, float* d_slabs, int32_t ldc, int64_t slab_stride, int32_t max_slabs,
                                int32_t* n_slabs, int32_t Mw, int32_t F, int32_t nodes, void* stream) {
  using namespace hmp;
  HMP_CHECK_ARG(d_dz && d_h && d_slabs && n_slabs && Mw > 0 && F > 0 && nodes > 0 && max_slabs >= 1, "hmp_gemm_bf16_dw: bad argument");
  HMP_CHECK_ARG(split == 0 || (d_dz2 && split > 0 && split < Mw && (split & 255) == 0), "hmp_gemm_bf16_dw: split %d of %d columns", split, Mw);
  HMP_CHECK_ARG(ldc >= F + 1 && slab_stride >= (int64_t)Mw * ldc, "hmp_gemm_bf16_dw: slab geometry");
  GemmBatch gb;
  memset(&gb, 0, sizeof(gb));
  gb.n = 1;
  GemmProblem& p = gb.p[0];
  p.A = reinterpret_cast<const float*>(d_dz); p.lda = lddz; p.a_bf16 = 1; p.trans_a = 1;
  if (split) { p.A2 = reinterpret_cast<const float*>(d_dz2); p.lda2 = lddz2; p.a_split = split; }
  p.B = reinterpret_cast<const float*>(d_h); p.ldb = ldh; p.b_bf16 = h_bf16 ? 1 : 0; p.trans_b = 0;
  p.C = d_slabs; p.ldc = ldc; p.slab_stride = slab_stride;
  p.M = Mw; p.N = F + 1; p.K = nodes; p.n_real = F; p.aug_ones = 1;
  p.epi = EPI_NONE;
  HMP_TRY(gemm_bf16_launch(gb, true, max_slabs, (hipStream_t)stream));
  *n_slabs = gb.p[0].ksplit;
  return HMP_OK;
}
