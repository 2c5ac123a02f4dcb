// K1 -- CSR gather / segment reduce, and the fused per-layer SAGE aggregation built on it.
//
// Work decomposition: one ROW GROUP of GS lanes per destination row (GS = 8/16/32/64, chosen from the
// row width so that each lane owns one float4 of the row; several rows share a wavefront when rows are
// narrower than 256 floats).  A group walks its CSR segment four neighbours at a time: the neighbour
// ids are read with one group-uniform load each, the four source rows are fetched as independent
// 16-byte-per-lane loads (GS*16 contiguous bytes per row: one fully coalesced request), then added in
// edge order, so the sum has the same association as a sequential scatter_add over the edge list.
// No atomics anywhere: the backward pass gathers over the transposed lists (CSC) instead of scattering.
//
// HBM/L2 traffic per row: deg * F * 4 bytes of source rows + 4*deg + 8 bytes of indices + F*4 out.
#include <type_traits>

#include "kernels.h"

namespace hmp {

KT_DEFINE(agg)
KT_BLOCKS_DEFINE(agg)

__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }
[[maybe_unused]] constexpr int WIN_THREADS_CHAIN = 512;   // graph-local chain kernel: CHAIN_GROUPS groups of 256 threads (1024 threads would cap the
[[maybe_unused]] constexpr int CHAIN_GROUPS = WIN_THREADS_CHAIN / 256;  // tile routines at 128 VGPRs: measured 1 201 spilled registers, 0.49 ms)

// raw buffer descriptor over [base, base + bytes): loads past the end return 0 without touching memory
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// ----- row access helpers -----------------------------------------------------------------------
// VEC = 4: 16-byte accesses (pointer and ld 16-byte aligned); VEC = 1: scalar fall-back for arbitrary ld.
template <int VEC>
struct Acc;
template <>
struct Acc<4> {
  float4 v;
  __device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
  __device__ __forceinline__ void add(const Acc& o) { v.x += o.v.x; v.y += o.v.y; v.z += o.v.z; v.w += o.v.w; }
  // mean of a segment: sum * (1 / deg), the reciprocal computed ONCE (an IEEE division; wave-uniform where the row is) and applied
  // by explicit fused multiply-adds -- the same form in every kernel, so all of them stay bit-identical to each other (the
  // backward pass multiplies by the same reciprocal: hmp plan's degf).  x / deg per element cost 4 divisions per lane and edge type.
  __device__ __forceinline__ void add_div(const Acc& o, float d) {
    const float r = 1.0f / d;
    v.x = fmaf(o.v.x, r, v.x); v.y = fmaf(o.v.y, r, v.y); v.z = fmaf(o.v.z, r, v.z); v.w = fmaf(o.v.w, r, v.w);
  }
  // explicit fused multiply-adds (two v_pk_fma_f32): left to the compiler, an 8-edge batch became 2 packed multiplies + 2 packed adds per
  // edge and a 2-edge batch 2 packed fmas -- a third more vector instructions, and a rounding that depended on the batch shape
  __device__ __forceinline__ void add_mul(const Acc& o, float r) {
    v.x = fmaf(o.v.x, r, v.x); v.y = fmaf(o.v.y, r, v.y); v.z = fmaf(o.v.z, r, v.z); v.w = fmaf(o.v.w, r, v.w);
  }
  __device__ __forceinline__ void div(float d) { v.x /= d; v.y /= d; v.z /= d; v.w /= d; }
  __device__ __forceinline__ float& at(int i) { return (&v.x)[i]; }
};
template <>
struct Acc<1> {
  float v;
  __device__ __forceinline__ void zero() { v = 0.f; }
  __device__ __forceinline__ void load(const float* p) { v = *p; }
  __device__ __forceinline__ void store(float* p) const { *p = v; }
  __device__ __forceinline__ void add(const Acc& o) { v += o.v; }
  __device__ __forceinline__ void add_div(const Acc& o, float d) { v = fmaf(o.v, 1.0f / d, v); }
  __device__ __forceinline__ void add_mul(const Acc& o, float r) { v = fmaf(o.v, r, v); }
  __device__ __forceinline__ void div(float d) { v /= d; }
  __device__ __forceinline__ float& at(int) { return v; }
};

// Projected rows stored as bf16 (bf16 compute mode at 10^6 rows: halves the projection's write and the gather's read traffic):
// ZB = true reads 4 bf16 (8 bytes) at ELEMENT index `idx` of the buffer and widens them; the buffer is typed float* like the
// fp32 one, offsets and leading dimensions count elements either way.
template <bool ZB>
__device__ __forceinline__ void load_z(Acc<4>& a, const float* base, int64_t idx) {
  if constexpr (ZB) {
    const uint2 b = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + idx);
    a.v = make_float4(__uint_as_float(b.x << 16), __uint_as_float(b.x & 0xffff0000u), __uint_as_float(b.y << 16),
                      __uint_as_float(b.y & 0xffff0000u));
  } else {
    a.load(base + idx);
  }
}

// counterpart for buffers WRITTEN as bf16 (round to nearest even): 4 elements at element index `idx`
template <bool ZB>
__device__ __forceinline__ void store_z(const Acc<4>& a, float* base, int64_t idx) {
  if constexpr (ZB) {
    typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
    bf4 b;
    b[0] = (__bf16)a.v.x; b[1] = (__bf16)a.v.y; b[2] = (__bf16)a.v.z; b[3] = (__bf16)a.v.w;
    *reinterpret_cast<bf4*>(reinterpret_cast<uint16_t*>(base) + idx) = b;
  } else {
    a.store(base + idx);
  }
}

// sum_{k in [b,e)} x[col[k]][c .. c+VEC)   in edge order; NV column chunks per lane (stride GS*VEC).
// Neighbours are processed in batches of UB = 8 (4 for wide rows) with NO tail loop: ids beyond the row are clamped to the
// last valid entry (same address => cache hit) and masked at the add.  A row of degree <= 8 therefore costs three
// dependent memory round trips (extent, ids, rows) instead of one per tail neighbour.
template <int GS, int NV, int VEC, bool ZB = false>
__device__ __forceinline__ void gather_sum(Acc<VEC> (&acc)[NV], const float* __restrict__ x, int ld, const int* __restrict__ col,
                                           int b, int e, int c0, int F) {
  constexpr int UB = (NV == 1) ? 8 : 4;
  for (int k = b; k < e; k += UB) {
    int j[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) j[u] = col[min(k + u, e - 1)];
    Acc<VEC> v[UB][NV];
#pragma unroll
    for (int u = 0; u < UB; ++u)
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int c = c0 + q * GS * VEC;
        if (c < F) {
          if constexpr (ZB && VEC == 4) load_z<true>(v[u][q], x, (int64_t)j[u] * ld + c);
          else v[u][q].load(x + (int64_t)j[u] * ld + c);
        }
      }
    const int cnt = e - k;
#pragma unroll
    for (int u = 0; u < UB; ++u)
      if (u < cnt) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int c = c0 + q * GS * VEC;
          if (c < F) acc[q].add(v[u][q]);
        }
      }
  }
}

// sum_k g[tcol[k]][c..] * rdeg(tcol[k])  (backward of the mean: every edge carries 1 / max(deg(dst), 1)).
// `rdeg` (the reciprocal per destination, written by the plan) removes the two dependent rowptr loads per edge AND the
// per-edge fp32 divisions (4 per lane and edge: ~1 ms of the config-5 transposed aggregation); without it (unit entry
// point) the reciprocal is derived from rowptr.  Same clamped batches as gather_sum.
template <int GS, int NV, int VEC, bool GB = false>
__device__ __forceinline__ void gather_sum_w(Acc<VEC> (&acc)[NV], const float* __restrict__ g, int ld, const int* __restrict__ tcol,
                                             const int* __restrict__ rowptr, const float* __restrict__ rdeg, int mean, int b, int e,
                                             int c0, int F) {
  constexpr int UB = (NV == 1) ? 8 : 4;
  for (int k = b; k < e; k += UB) {
    int i[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) i[u] = tcol[min(k + u, e - 1)];
    float d[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      d[u] = 1.f;
      if (mean) {
        if (rdeg) {
          d[u] = rdeg[i[u]];
        } else {
          const int deg = rowptr[i[u] + 1] - rowptr[i[u]];
          d[u] = 1.f / (float)(deg > 1 ? deg : 1);
        }
      }
    }
    Acc<VEC> v[UB][NV];
#pragma unroll
    for (int u = 0; u < UB; ++u)
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int c = c0 + q * GS * VEC;
        if (c < F) {
          if constexpr (GB && VEC == 4) load_z<true>(v[u][q], g, (int64_t)i[u] * ld + c);
          else v[u][q].load(g + (int64_t)i[u] * ld + c);
        }
      }
    const int cnt = e - k;
#pragma unroll
    for (int u = 0; u < UB; ++u)
      if (u < cnt) {
#pragma unroll
        for (int q = 0; q < NV; ++q) {
          const int c = c0 + q * GS * VEC;
          if (c < F) {
            if (mean) acc[q].add_mul(v[u][q], d[u]); else acc[q].add(v[u][q]);
          }
        }
      }
  }
}

// ----- K1 unit kernels ----------------------------------------------------------------------------
template <int GS, int NV, int VEC>
__global__ __launch_bounds__(256) void segment_mean_fwd_kernel(const float* __restrict__ x, int ldx, int F, const int* __restrict__ rowptr,
                                                               const int* __restrict__ col, int n_rows, float* __restrict__ out, int ldo) {
  const int rpb = 256 / GS;
  const int row = blockIdx.x * rpb + threadIdx.x / GS;
  if (row >= n_rows) return;
  const int c0 = (threadIdx.x % GS) * VEC;
  Acc<VEC> acc[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) acc[q].zero();
  const int b = rowptr[row], e = rowptr[row + 1];
  gather_sum<GS, NV, VEC>(acc, x, ldx, col, b, e, c0, F);
  const float d = (float)((e - b) > 1 ? (e - b) : 1);
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int c = c0 + q * GS * VEC;
    if (c < F) { acc[q].div(d); acc[q].store(out + (int64_t)row * ldo + c); }
  }
}

template <int GS, int NV, int VEC>
__global__ __launch_bounds__(256) void segment_mean_bwd_kernel(const float* __restrict__ g, int ldg, int F, const int* __restrict__ t_rowptr,
                                                               const int* __restrict__ t_col, const int* __restrict__ rowptr, int n_rows,
                                                               float* __restrict__ gx, int ldgx) {
  const int rpb = 256 / GS;
  const int row = blockIdx.x * rpb + threadIdx.x / GS;
  if (row >= n_rows) return;
  const int c0 = (threadIdx.x % GS) * VEC;
  Acc<VEC> acc[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) acc[q].zero();
  gather_sum_w<GS, NV, VEC>(acc, g, ldg, t_col, rowptr, nullptr, 1, t_rowptr[row], t_rowptr[row + 1], c0, F);
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int c = c0 + q * GS * VEC;
    if (c < F) acc[q].store(gx + (int64_t)row * ldgx + c);
  }
}

// ----- neighbour ids from the ELL table (kernels.h: ELL_W) ---------------------------------------------
// The first 8 (PRE: 16) ids of `row`, requested by address arithmetic on the row alone: they travel in the same round trip as the
// row extent.  Slots past the degree n were never written: they are replaced by the first id (a line the gather touches anyway;
// the adds are masked by the callers exactly as for the clamped CSR loads), so every later load has a valid address.
template <bool PRE>
struct EllRow {
  int4 h[PRE ? 4 : 2];
  __device__ __forceinline__ void fetch(const int* __restrict__ ell, int row) {
    const int4* p = reinterpret_cast<const int4*>(ell) + (int64_t)row * (ELL_W / 4);
#pragma unroll
    for (int q = 0; q < (PRE ? 4 : 2); ++q) h[q] = p[q];
  }
  __device__ __forceinline__ int word(int u) const {
    const int4& v = h[u >> 2];
    return (u & 3) == 0 ? v.x : (u & 3) == 1 ? v.y : (u & 3) == 2 ? v.z : v.w;
  }
  __device__ __forceinline__ void ids(int n, int (&j)[8], int (&jt)[PRE ? 8 : 1]) const {
    const int first = n > 0 ? h[0].x : 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      j[u] = u < n ? word(u) : first;
      if constexpr (PRE) jt[u] = 8 + u < n ? word(8 + u) : first;
    }
  }
};

// ----- fused SAGE layer aggregation -----------------------------------------------------------------
// out[t][i] = dropout(act( zroot[i] + bias + sum_e mean_{k in N_e(i)} z_e[col_k] ))   (one row group, result also in `tot`)
template <int GS, int NV, bool ZB = false, bool HB = false>
__device__ __forceinline__ void agg_row(const AggDst& D, int mean, int row, int c0, Acc<4> (&tot)[NV]) {
  constexpr int VEC = 4;
  // Everything whose address depends on the row alone is requested here, and nothing of it is consumed before the gathers
  // (a use inside one of these branches would put an s_waitcnt in it): the device step counter of the dropout coordinates, the
  // root row and the bias, the row extents of every incoming edge type, the ELL ids of the first pair -- ONE round trip.
  KT(3);
  DropCfg dcfg = D.drop;
  uint32_t step_add = 0;
  if (D.drop_on) step_add = drop_step_vload(D.drop.step_dev);
  Acc<VEC> bs[NV];
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int c = c0 + q * GS * VEC;
    tot[q].zero();
    bs[q].zero();
    if (c < D.F) {
      if (D.zroot) load_z<ZB>(tot[q], D.zroot, (int64_t)row * D.ldzr + D.roff + c);
      if (D.bias) bs[q].load(D.bias + c);
    }
  }
  int rb[AGG_MAX_IN], re[AGG_MAX_IN];
#pragma unroll
  for (int ii = 0; ii < AGG_MAX_IN; ++ii) {
    rb[ii] = re[ii] = 0;
    if (ii < D.n_in) { rb[ii] = D.in[ii].rowptr[row]; re[ii] = D.in[ii].rowptr[row + 1]; }
  }
  constexpr bool ELLP = NV == 1 && GS < 64;  // the pair path below
  EllRow<GS <= 16> hA, hB;
  bool ell0 = false;
  if constexpr (ELLP) {
    const AggIn& A0 = D.in[0];
    const AggIn& A1 = D.in[D.n_in > 1 ? 1 : 0];
    ell0 = D.n_in > 0 && A0.ell && A1.ell;  // block-uniform
    if (ell0) {
      hA.fetch(A0.ell, row);
      hB.fetch(A1.ell, row);
    }
  }
  KTW(4);
#pragma unroll
  for (int q = 0; q < NV; ++q) tot[q].add(bs[q]);  // (root + bias) first, as the sequential code
  if constexpr (NV == 1 && GS == 64) {
    // One wavefront per row (the caller made `row` wave-uniform): the neighbour ids of BOTH edge types of a pair are scalar
    // loads issued together, then 16 rows of the first type in flight, then 16 of the second through the same registers --
    // extents, ids, rows, rows: 4 dependent round trips for in-degrees <= 16 (the 8 + 8 pair batch needed a tail pass from 9
    // neighbours on) and no clamped duplicate loads for the short list of a pair (rooms -> objects: one neighbour).
    constexpr int UB = 16;
    const bool cin = c0 < D.F;
    const int cc = cin ? c0 : 0;
#pragma unroll
    for (int ii = 0; ii < AGG_MAX_IN; ii += 2) {
      if (ii >= D.n_in) break;
      const bool has2 = ii + 1 < D.n_in;
      const AggIn& I0 = D.in[ii];
      const AggIn& I1 = D.in[has2 ? ii + 1 : ii];
      const int b0 = rb[ii], e0 = re[ii];
      const int b1 = has2 ? rb[ii + 1] : b0, e1 = has2 ? re[ii + 1] : b0;
      int j0[UB], j1[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        j0[u] = I0.col[e0 > b0 ? min(b0 + u, e0 - 1) : 0];
        j1[u] = I1.col[e1 > b1 ? min(b1 + u, e1 - 1) : 0];
      }
      Acc<VEC> v[UB], a0[1], a1[1];
      a0[0].zero();
      a1[0].zero();
      if (e0 > b0) {  // wave-uniform
#pragma unroll
        for (int u = 0; u < UB; ++u) load_z<ZB>(v[u], I0.z, I0.coff + (int64_t)j0[u] * I0.ldz + cc);
#pragma unroll
        for (int u = 0; u < UB; ++u)
          if (b0 + u < e0) a0[0].add(v[u]);
      }
      if (e1 > b1) {
#pragma unroll
        for (int u = 0; u < UB; ++u) load_z<ZB>(v[u], I1.z, I1.coff + (int64_t)j1[u] * I1.ldz + cc);
#pragma unroll
        for (int u = 0; u < UB; ++u)
          if (b1 + u < e1) a1[0].add(v[u]);
      }
      // (ZB: the column offset goes into the element index, the base pointer stays the buffer start)
      if (e0 - b0 > UB) gather_sum<GS, 1, VEC, ZB>(a0, ZB ? I0.z : I0.z + I0.coff, I0.ldz, I0.col, b0 + UB, e0, ZB ? c0 + I0.coff : c0, ZB ? D.F + I0.coff : D.F);
      if (e1 - b1 > UB) gather_sum<GS, 1, VEC, ZB>(a1, ZB ? I1.z : I1.z + I1.coff, I1.ldz, I1.col, b1 + UB, e1, ZB ? c0 + I1.coff : c0, ZB ? D.F + I1.coff : D.F);
      if (cin) {
        if (e0 > b0) tot[0].add_div(a0[0], mean ? (float)(e0 - b0) : 1.f);
        if (e1 > b1) tot[0].add_div(a1[0], mean ? (float)(e1 - b1) : 1.f);
      }
    }
  } else if constexpr (NV == 1) {
    // Incoming edge types in PAIRS: the neighbour ids of both types travel together, then the 16 neighbour rows -- three
    // dependent round trips (extents, ids, rows) for two edge types instead of five.  A missing partner aliases the
    // first type with an empty extent (loads hit the same lines, adds are masked).  Sums keep the edge order per type
    // and the type order of the sequential code.
    constexpr int UB = 8;
#pragma unroll
    for (int ii = 0; ii < AGG_MAX_IN; ii += 2) {
      if (ii >= D.n_in) break;
      const bool has2 = ii + 1 < D.n_in;
      const AggIn& I0 = D.in[ii];
      const AggIn& I1 = D.in[has2 ? ii + 1 : ii];
      const int b0 = rb[ii], e0 = re[ii];
      const int b1 = has2 ? rb[ii + 1] : b0, e1 = has2 ? re[ii + 1] : b0;
      // GS <= 16 (rows of <= 64 floats, the MP3D hidden width): ids of neighbours 0..7 AND 8..15 of both types in the same round
      // trip -- a row of 9..16 neighbours (most 16-row blocks of a scene-graph batch hold one) then needs one more round trip
      // for its second batch of rows instead of two.  Wider rows keep the plain tail: the 16 extra registers cost them a wave
      // per SIMD (config 4: 0.140 -> 0.151 ms).
      constexpr bool PRE = GS <= 16;
      int j0[UB], j1[UB], j0t[PRE ? UB : 1], j1t[PRE ? UB : 1];
      if (ii == 0 ? ell0 : (I0.ell && I1.ell)) {  // block-uniform: ids by row address, no dependence on the extents
        if (ii != 0) {
          hA.fetch(I0.ell, row);
          hB.fetch(I1.ell, row);
        }
        hA.ids(e0 - b0, j0, j0t);
        hB.ids(e1 - b1, j1, j1t);
      } else {
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        j0[u] = I0.col[e0 > b0 ? min(b0 + u, e0 - 1) : 0];
        j1[u] = I1.col[e1 > b1 ? min(b1 + u, e1 - 1) : 0];
        if constexpr (PRE) {
          j0t[u] = I0.col[e0 > b0 ? min(b0 + UB + u, e0 - 1) : 0];
          j1t[u] = I1.col[e1 > b1 ? min(b1 + UB + u, e1 - 1) : 0];
        }
      }
      }
      Acc<VEC> v0[UB], v1[UB];
      const bool cin = c0 < D.F;
      const int cc = cin ? c0 : 0;
      if (ii == 0) KTW(5);
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        v0[u].load(I0.z + I0.coff + (int64_t)j0[u] * I0.ldz + cc);
        v1[u].load(I1.z + I1.coff + (int64_t)j1[u] * I1.ldz + cc);
      }
      if (ii == 0) KTW(6);
      Acc<VEC> a0[1], a1[1];
      a0[0].zero();
      a1[0].zero();
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (b0 + u < e0) a0[0].add(v0[u]);
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (b1 + u < e1) a1[0].add(v1[u]);
      int done = UB;
      if constexpr (PRE) {
        if (e0 - b0 > UB || e1 - b1 > UB) {  // second batch through the same registers (ids already here)
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            v0[u].load(I0.z + I0.coff + (int64_t)j0t[u] * I0.ldz + cc);
            v1[u].load(I1.z + I1.coff + (int64_t)j1t[u] * I1.ldz + cc);
          }
#pragma unroll
          for (int u = 0; u < UB; ++u)
            if (b0 + UB + u < e0) a0[0].add(v0[u]);
#pragma unroll
          for (int u = 0; u < UB; ++u)
            if (b1 + UB + u < e1) a1[0].add(v1[u]);
        }
        done = 2 * UB;
      }
      if (e0 - b0 > done) gather_sum<GS, 1, VEC>(a0, I0.z + I0.coff, I0.ldz, I0.col, b0 + done, e0, c0, D.F);
      if (e1 - b1 > done) gather_sum<GS, 1, VEC>(a1, I1.z + I1.coff, I1.ldz, I1.col, b1 + done, e1, c0, D.F);
      if (cin) {
        if (e0 > b0) tot[0].add_div(a0[0], mean ? (float)(e0 - b0) : 1.f);
        if (e1 > b1) tot[0].add_div(a1[0], mean ? (float)(e1 - b1) : 1.f);
      }
    }
  } else {
#pragma unroll
  for (int ii = 0; ii < AGG_MAX_IN; ++ii) {
    if (ii >= D.n_in) break;
    const AggIn& I = D.in[ii];
    const int b = rb[ii], e = re[ii];
    if (e == b) continue;
    Acc<VEC> acc[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) acc[q].zero();
    gather_sum<GS, NV, VEC>(acc, I.z + I.coff, I.ldz, I.col, b, e, c0, D.F);
    const float d = mean ? (float)(e - b) : 1.f;
#pragma unroll
    for (int q = 0; q < NV; ++q) tot[q].add_div(acc[q], d);
  }
  }
  KTW(7);
  dcfg.step += step_add;
  dcfg.step_dev = nullptr;
#pragma unroll
  for (int q = 0; q < NV; ++q) {
    const int c = c0 + q * GS * VEC;
    if (c >= D.F) continue;
    bool keep[4] = {true, true, true, true};
    if (D.drop_on) drop_keep4(dcfg, (uint32_t)row * (uint32_t)(D.ldo >> 2) + (uint32_t)(c >> 2), keep);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = tot[q].at(i);
      if (D.act == HMP_ACT_RELU) v = v > 0.f ? v : 0.f;
      else if (D.act == HMP_ACT_ELU) v = v > 0.f ? v : expm1f(v);
      // a dropped element is stored as -0.0f (a kept one that happens to be zero as +0.0f): numerically both are 0 for
      // every consumer, and the backward pass reads the keep bit off the sign instead of regenerating the draws
      if (D.drop_on) v = keep[i] ? (v * D.drop.scale + 0.0f) : -0.0f;  // "+ 0.0f": a kept -0.0 becomes +0.0
      tot[q].at(i) = v;
    }
    if constexpr (HB && VEC == 4) store_z<true>(tot[q], D.out, (int64_t)row * D.ldo + c);
    else tot[q].store(D.out + (int64_t)row * D.ldo + c);
  }
}

// Masked cross entropy of one output row held by its row group (lane = 4 consecutive logits): models/utils.py:143-148
// with mask = label != ignored; writes the gradient of the SUM loss and the row's {loss, valid} pair.  The group
// reductions are xor butterflies below GS, i.e. inside the (GS-aligned) row group.
template <int GS>
__device__ __forceinline__ void ce_rowgroup(const AggDst& D, NetState* state, int row, int c0, Acc<4>& t, int64_t y) {
  const int nc = D.ce_classes;
  float v[4];
  bool in[4];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    in[i] = c0 + i < nc;
    v[i] = in[i] ? t.at(i) : -INFINITY;
    m = fmaxf(m, v[i]);
  }
#pragma unroll
  for (int o = GS / 2; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += in[i] ? expf(v[i] - m) : 0.f;
#pragma unroll
  for (int o = GS / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float lse = m + logf(s);
  const bool valid = (y != D.ce_ignored);
  const bool bad = valid && (y < 0 || y >= nc);
  const bool use = valid && !bad;
  float ly = 0.f;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool hit = use && (int64_t)(c0 + i) == y;
    if (hit) ly = v[i];
    if (use && in[i]) (&g.x)[i] = expf(v[i] - lse) - (hit ? 1.f : 0.f);
  }
#pragma unroll
  for (int o = GS / 2; o > 0; o >>= 1) ly += __shfl_xor(ly, o);
  if (c0 < D.ce_ldg) *reinterpret_cast<float4*>(D.ce_grad + (int64_t)row * D.ce_ldg + c0) = g;
  if ((threadIdx.x % GS) == 0) {
    D.ce_row_lv[2 * row] = use ? (lse - ly) : 0.f;
    D.ce_row_lv[2 * row + 1] = use ? 1.f : 0.f;
    if (bad && state) atomicOr(&state->status, 2);
  }
}

template <int GS, int NV, bool ZB = false, bool HB = false>
__global__ __launch_bounds__(256) void agg_fwd_kernel(const AggArgs a) {
  int ti = 0;
  while (ti + 1 < a.n && (int)blockIdx.x >= a.bstart[ti + 1]) ++ti;
  karg_warm<9>((int)offsetof(AggArgs, d) + ti * (int)sizeof(AggDst), (int)sizeof(AggDst));
  const AggDst& D = a.d[ti];
  // rows per workgroup: 256 / GS, or 8 for an entry of heavy rows in a small launch (AggDst::tile_rows; the other row groups idle)
  const int rpb = (D.tile_rows == 8 && 256 / GS > 8) ? 8 : 256 / GS;
  int local = blockIdx.x - D.block_start;
  // workgroups go to the 8 XCDs round robin by block id: give XCD x the x-th contiguous eighth of the rows, so that the
  // neighbour rows a run of consecutive destinations shares (scene graphs: the same room) are fetched into ONE L2, not 8
  if (a.xcd) local = (local & 7) * ((cdiv_dev(D.n_rows, rpb) + 7) >> 3) + (local >> 3);
  if ((int)threadIdx.x / GS >= rpb) return;
  int row = local * rpb + threadIdx.x / GS;
  // GS = 64: the wavefront IS the row group, so the row (and with it every extent / neighbour id) is wave-uniform; saying so
  // moves those loads and the address arithmetic to the scalar unit
  if (GS == 64) row = __builtin_amdgcn_readfirstlane(row);
  if (row >= D.n_rows) return;
  Acc<4> tot[NV];
  const int c0 = (threadIdx.x % GS) * 4;
  // the label is requested before the aggregation (it depends on nothing): one round trip less behind the last gather
  int64_t y = 0;
  if (NV == 1 && D.ce_labels) y = D.ce_labels[row];
  agg_row<GS, NV, ZB, HB>(D, a.mean, row, c0, tot);
  if constexpr (NV == 1) {
    if (D.ce_labels) ce_rowgroup<GS>(D, a.state, row, c0, tot[0], y);
  }
}

// ----- aggregation of layer l FUSED with the projection of layer l+1 --------------------------------------------
// The projection Z[l+1][t] = H[l+1][t] * Wp[l+1][t]^T is row-local, so the block that has just aggregated 16 rows of
// H[l+1][t] keeps them in LDS ([k][row] image, LD 17) and multiplies them on the matrix cores right away
// (v_mfma_f32_16x16x4_f32: 16 rows x 16 packed columns per accumulator, exact fp32).  One kernel (and one ~4 us launch
// floor, one cross-XCD hand-off of H) less per layer; H is still written to HBM for the backward pass.
// Wp is read straight from L2 (it is <= 200 KB and shared by every block): lane (n, kq) loads 16 bytes
// Wp[n0 + n][16 i + 4 kq .. +3] and feeds 4 MFMAs with them; the k order inside a 16-block is permuted identically
// for both operands (k = 16 i + 4 kq + u at step u), which a sum over k does not care about.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// (32-row tiles -- half the per-row weight traffic, two MFMA row halves sharing each B load -- were measured SLOWER on the
// H-tree config 4: 0.585 against 0.474 ms/step; half as many blocks, twice the gather passes per block.)
// GS >= 32 (H-tree, hidden 128: ~1000 blocks of 16 rows): 4 waves per SIMD = 4 blocks per CU keeps the whole launch resident in
// one round (measured 0.152 -> 0.140 ms over the 4 layers of config 4 despite 148 bytes of spill; the same bound made the
// backward kernel slower and is not applied there)
// One 16-row tile [row0, row0 + 16) of destination entry D, rows below row_end, by ONE group of 256 threads (tid = thread inside
// the group) with its own LDS image Hs [256][17].  Every group of the workgroup must call this the same number of times: the
// barrier between the aggregation and the projection is workgroup-wide (`valid` = false: a group without a tile only joins it).
template <int GS>
__device__ __forceinline__ void agg_proj_tile(const AggArgs& a, const AggDst& D, int row0, int row_end, bool valid, float* Hs) {
  constexpr int TM = 16, LDH = 17;
  constexpr int RPP = 256 / GS;            // rows aggregated per pass
  constexpr int NP = TM / RPP;             // passes (GS = 16: 1, 32: 2, 64: 4)
  const int tid = threadIdx.x & 255;
  const int c0 = (tid % GS) * 4;
  const int lane = threadIdx.x & 63, w = tid >> 6;
  const int n = lane & 15, kq = lane >> 4;
  KT(0);
  // The weights of the projection do not depend on the aggregation: request the first PT column tiles of this wave (first
  // 64-deep k trip each) BEFORE the gather, so their round trip (every block pulls the whole Wp through its CU, ~25 GB/s)
  // overlaps the three round trips of the gather instead of following them.
  constexpr int PT = 3;
  float4 pre[PT][4];
  const int n_ct = (valid && D.pw) ? (D.pncols + 15) >> 4 : 0;
  if (valid && D.pw) {  // group-uniform
#pragma unroll
    for (int it = 0; it < PT; ++it) {
      const int col = min((w + 4 * it) * 16 + n, D.pncols - 1);  // clamped: tiles past n_ct are never used
      const float* wrow = D.pw + (int64_t)col * D.pldw;
#pragma unroll
      for (int u = 0; u < 4; ++u) pre[it][u] = *reinterpret_cast<const float4*>(wrow + min(16 * u, D.pK - 16) + 4 * kq);
    }
  }
  if (valid) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int m = p * RPP + tid / GS;
      const int row = row0 + m;
      Acc<4> tot[1];
      tot[0].zero();
      if (row < row_end) agg_row<GS, 1>(D, a.mean, row, c0, tot);
      if (c0 < D.pK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) Hs[(c0 + i) * LDH + m] = (c0 < D.F && row < row_end) ? tot[0].at(i) : 0.f;
      }
    }
  }
  __syncthreads();
  if (!valid || D.pw == nullptr) return;  // group-uniform: no tile / this node type is not read by the next layer
  KT(1);
  // one 16-column tile: bv0 = prefetched weights of the first k trip (null: load them here)
  auto tile = [&](int ct, const float4* bv0) {
    const int col = ct * 16 + n;
    const float* wrow = D.pw + (int64_t)min(col, D.pncols - 1) * D.pldw;  // clamped: padded columns are never stored
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // 4 k-blocks of 16 per trip: the 4 (clamped) 16-byte weight loads are in flight together, then 16 MFMAs
    for (int kb = 0; kb < D.pK; kb += 64) {
      float4 bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        bv[u] = (bv0 && kb == 0) ? bv0[u] : *reinterpret_cast<const float4*>(wrow + min(kb + 16 * u, D.pK - 16) + 4 * kq);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (kb + 16 * u >= D.pK) break;
        const float* hp = Hs + (kb + 16 * u + 4 * kq) * LDH + n;  // A operand: row m = lane & 15 of the tile
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hp[0 * LDH], bv[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hp[1 * LDH], bv[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hp[2 * LDH], bv[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hp[3 * LDH], bv[u].w, acc, 0, 0, 0);
      }
    }
    // D layout: column = lane & 15, row = (lane >> 4) * 4 + reg
    if (col < D.pncols) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + kq * 4 + r;
        if (row < row_end) D.pz[(int64_t)row * D.pldz + col] = acc[r];
      }
    }
  };
#pragma unroll
  for (int it = 0; it < PT; ++it)
    if (w + 4 * it < n_ct) tile(w + 4 * it, pre[it]);
  for (int ct = w + 4 * PT; ct < n_ct; ct += 4) tile(ct, nullptr);
  KT(2);
}

template <int GS>
__global__ __launch_bounds__(256, GS == 32 ? 3 : (GS == 64 ? 2 : 1)) void agg_proj_fwd_kernel(const AggArgs a) {
  __shared__ float Hs[256 * 17];
  KT_BLOCK_BEGIN();
  KT_SPAN_BEGIN(40);
  int ti = 0;
  while (ti + 1 < a.n && (int)blockIdx.x >= a.bstart[ti + 1]) ++ti;
  karg_warm<9>((int)offsetof(AggArgs, d) + ti * (int)sizeof(AggDst), (int)sizeof(AggDst));
  const AggDst& D = a.d[ti];
  // (an entry of heavy rows is cut into tiles of 8: see AggDst::tile_rows)
  const int row0 = ((int)blockIdx.x - D.block_start) * D.tile_rows;
  agg_proj_tile<GS>(a, D, row0, min(row0 + D.tile_rows, D.n_rows), true, Hs);
  KT_SPAN_END(40, ti);
  KT_BLOCK_END();
}

// fixed-order sum of the per-row {loss, valid} pairs -> {loss_sum, count}; run by ONE block (256 threads)
__device__ __forceinline__ void finalize_loss(const float* __restrict__ row_lv, int n_rows, float* __restrict__ out2, NetState* state) {
  __shared__ float sl[256], sv[256];
  float l = 0.f, v = 0.f;
  for (int r = threadIdx.x; r < n_rows; r += 256) { l += row_lv[2 * r]; v += row_lv[2 * r + 1]; }
  sl[threadIdx.x] = l;
  sv[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sv[threadIdx.x] += sv[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out2[0] = sl[0];
    out2[1] = sv[0];
    if (state) { state->loss_sum = sl[0]; state->count = sv[0]; }
  }
}

// dz[s][j, seg_e] = sum_{k in out_e(j)} g'[dst_k] / deg(dst_k);   dz[s][j, root] = g'[s][j]
// GB: the gradient rows (TAggOut::g, TAggSrc::groot) hold bf16 elements (bf16 compute mode at 10^6 rows, see load_z);
// DZB: so does the output dz (read back by the bf16 GEMMs as their A operand)
// one row of source entry S by its row group of GS lanes (c0 = first column of the lane); `row` must be < S.n_rows
template <int GS, int NV, bool GB = false, bool DZB = false>
__device__ __forceinline__ void agg_bwd_row(const TAggArgs& a, const TAggSrc& S, int row, int c0) {
  constexpr int VEC = 4;
  int rb[AGG_MAX_IN], re[AGG_MAX_IN];
#pragma unroll
  for (int oi = 0; oi < AGG_MAX_IN; ++oi) {
    rb[oi] = re[oi] = 0;
    if (oi < S.n_out) { rb[oi] = S.out[oi].t_rowptr[row]; re[oi] = S.out[oi].t_rowptr[row + 1]; }
  }
  // ELL ids of the first pair in the round trip of the extents (see agg_row)
  EllRow<GS <= 16> hA, hB;
  bool ell0 = false;
  if constexpr (NV == 1) {
    const TAggOut& A0 = S.out[0];
    const TAggOut& A1 = S.out[S.n_out > 1 ? 1 : 0];
    ell0 = S.n_out > 0 && A0.t_ell && A1.t_ell;  // block-uniform
    if (ell0) {
      hA.fetch(A0.t_ell, row);
      hB.fetch(A1.t_ell, row);
    }
  }
  // (the 16-wide scalar-id form of agg_row was measured here too: 6.94 -> 7.02 ms at config 5, not kept)
  if constexpr (NV == 1) {
    // outgoing edge types in PAIRS (see agg_row): ids of both, then 1/deg + gradient rows of both
    constexpr int UB = 8;
#pragma unroll
    for (int oi = 0; oi < AGG_MAX_IN; oi += 2) {
      if (oi >= S.n_out) break;
      const bool has2 = oi + 1 < S.n_out;
      const TAggOut& O0 = S.out[oi];
      const TAggOut& O1 = S.out[has2 ? oi + 1 : oi];
      const int b0 = rb[oi], e0 = re[oi];
      const int b1 = has2 ? rb[oi + 1] : b0, e1 = has2 ? re[oi + 1] : b0;
      // GS <= 16: the ids of out-edges 8..15 travel with those of 0..7 (see agg_row): one round trip less for rows of 9..16 edges
      constexpr bool PRE = GS <= 16;
      int i0[UB], i1[UB], i0t[PRE ? UB : 1], i1t[PRE ? UB : 1];
      if (oi == 0 ? ell0 : (O0.t_ell && O1.t_ell)) {  // block-uniform: ids by row address (see agg_row)
        if (oi != 0) {
          hA.fetch(O0.t_ell, row);
          hB.fetch(O1.t_ell, row);
        }
        hA.ids(e0 - b0, i0, i0t);
        hB.ids(e1 - b1, i1, i1t);
      } else {
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        i0[u] = O0.t_col[e0 > b0 ? min(b0 + u, e0 - 1) : 0];
        i1[u] = O1.t_col[e1 > b1 ? min(b1 + u, e1 - 1) : 0];
        if constexpr (PRE) {
          i0t[u] = O0.t_col[e0 > b0 ? min(b0 + UB + u, e0 - 1) : 0];
          i1t[u] = O1.t_col[e1 > b1 ? min(b1 + UB + u, e1 - 1) : 0];
        }
      }
      }
      float d0[UB], d1[UB];
      Acc<VEC> v0[UB], v1[UB];
      const int cc0 = c0 < O0.F ? c0 : 0, cc1 = c0 < O1.F ? c0 : 0;
      const bool dg = a.mean && O0.degf && O1.degf;  // block-uniform
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        d0[u] = dg ? O0.degf[i0[u]] : 1.f;
        d1[u] = dg ? O1.degf[i1[u]] : 1.f;
        load_z<GB>(v0[u], O0.g, (int64_t)i0[u] * O0.ldg + cc0);
        load_z<GB>(v1[u], O1.g, (int64_t)i1[u] * O1.ldg + cc1);
      }
      if (a.mean && !dg) {
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int g0 = O0.rowptr[i0[u] + 1] - O0.rowptr[i0[u]], g1 = O1.rowptr[i1[u] + 1] - O1.rowptr[i1[u]];
          d0[u] = 1.f / (float)(g0 > 1 ? g0 : 1);
          d1[u] = 1.f / (float)(g1 > 1 ? g1 : 1);
        }
      }
      Acc<VEC> a0[1], a1[1];
      a0[0].zero();
      a1[0].zero();
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (b0 + u < e0) a0[0].add_mul(v0[u], d0[u]);
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (b1 + u < e1) a1[0].add_mul(v1[u], d1[u]);
      int done = UB;
      if constexpr (PRE) {
        if (e0 - b0 > UB || e1 - b1 > UB) {  // second batch through the same registers (ids already here)
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            d0[u] = dg ? O0.degf[i0t[u]] : 1.f;
            d1[u] = dg ? O1.degf[i1t[u]] : 1.f;
            load_z<GB>(v0[u], O0.g, (int64_t)i0t[u] * O0.ldg + cc0);
            load_z<GB>(v1[u], O1.g, (int64_t)i1t[u] * O1.ldg + cc1);
          }
          if (a.mean && !dg) {
#pragma unroll
            for (int u = 0; u < UB; ++u) {
              const int g0 = O0.rowptr[i0t[u] + 1] - O0.rowptr[i0t[u]], g1 = O1.rowptr[i1t[u] + 1] - O1.rowptr[i1t[u]];
              d0[u] = 1.f / (float)(g0 > 1 ? g0 : 1);
              d1[u] = 1.f / (float)(g1 > 1 ? g1 : 1);
            }
          }
#pragma unroll
          for (int u = 0; u < UB; ++u)
            if (b0 + UB + u < e0) a0[0].add_mul(v0[u], d0[u]);
#pragma unroll
          for (int u = 0; u < UB; ++u)
            if (b1 + UB + u < e1) a1[0].add_mul(v1[u], d1[u]);
        }
        done = 2 * UB;
      }
      if (e0 - b0 > done) gather_sum_w<GS, 1, VEC, GB>(a0, O0.g, O0.ldg, O0.t_col, O0.rowptr, O0.degf, a.mean, b0 + done, e0, c0, O0.F);
      if (e1 - b1 > done) gather_sum_w<GS, 1, VEC, GB>(a1, O1.g, O1.ldg, O1.t_col, O1.rowptr, O1.degf, a.mean, b1 + done, e1, c0, O1.F);
      if (c0 < O0.F) store_z<DZB>(a0[0], S.dz, (int64_t)row * S.lddz + O0.coff + c0);
      if (has2 && c0 < O1.F) store_z<DZB>(a1[0], S.dz, (int64_t)row * S.lddz + O1.coff + c0);
    }
  } else {
#pragma unroll
  for (int oi = 0; oi < AGG_MAX_IN; ++oi) {
    if (oi >= S.n_out) break;
    const TAggOut& O = S.out[oi];
    Acc<VEC> acc[NV];
#pragma unroll
    for (int q = 0; q < NV; ++q) acc[q].zero();
    gather_sum_w<GS, NV, VEC, GB>(acc, O.g, O.ldg, O.t_col, O.rowptr, O.degf, a.mean, rb[oi], re[oi], c0, O.F);
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int c = c0 + q * GS * VEC;
      if (c < O.F) store_z<DZB>(acc[q], S.dz, (int64_t)row * S.lddz + O.coff + c);
    }
  }
  }
  if (S.groot) {
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int c = c0 + q * GS * VEC;
      if (c < S.Froot) {
        Acc<VEC> v;
        load_z<GB>(v, S.groot, (int64_t)row * S.ldgr + c);
        store_z<DZB>(v, S.dz, (int64_t)row * S.lddz + S.roff + c);
      }
    }
  }
}

template <int GS, int NV, bool GB = false, bool DZB = false>
__global__ __launch_bounds__(256) void agg_bwd_kernel(const TAggArgs a) {
  if ((int)blockIdx.x == a.total_blocks) {  // the extra block (only launched when fin_row_lv is set)
    finalize_loss(a.fin_row_lv, a.fin_rows, a.fin_out2, a.fin_state);
    return;
  }
  int si = 0;
  while (si + 1 < a.n && (int)blockIdx.x >= a.bstart[si + 1]) ++si;
  karg_warm<10>((int)offsetof(TAggArgs, s) + si * (int)sizeof(TAggSrc), (int)sizeof(TAggSrc));
  const TAggSrc& S = a.s[si];
  const int rpb = (S.tile_rows == 8 && 256 / GS > 8) ? 8 : 256 / GS;  // see agg_fwd_kernel
  int local = blockIdx.x - S.block_start;
  if (a.xcd) local = (local & 7) * ((cdiv_dev(S.n_rows, rpb) + 7) >> 3) + (local >> 3);  // see agg_fwd_kernel
  if ((int)threadIdx.x / GS >= rpb) return;
  int row = local * rpb + threadIdx.x / GS;
  if (GS == 64) row = __builtin_amdgcn_readfirstlane(row);  // wave-uniform, see agg_fwd_kernel
  if (row >= S.n_rows) return;
  agg_bwd_row<GS, NV, GB, DZB>(a, S, row, (int)(threadIdx.x % GS) * 4);
}

// ----- transposed aggregation of layer l FUSED with its input-gradient GEMM ------------------------------------------
// dH[l][s] = (dZ[l][s] * Wp[l][s]) . act'(H[l][s]) is row-local, so the block that has just gathered 16 rows of dZ keeps
// them in LDS ([k][row] image, LD 17) and multiplies them on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32): one
// kernel and one hand-off of dZ through L2 less per layer.  dZ is still written to HBM (the weight-gradient GEMM reads it).
// Wp [ncols][ldw] is read from L2: lane (n, kq) loads Wp[kb + kq][n0 + n] (16 lanes = one 64-byte segment).
// one 16-row tile of source entry S by one group of 256 threads, LDS image Hs [ncols][17]; see agg_proj_tile for `valid`
template <int GS>
__device__ __forceinline__ void agg_bwd_dx_tile(const TAggArgs& a, const TAggSrc& S, int row0, int row_end, bool valid, float* Hs) {
  constexpr int TM = 16, LDH = 17, VEC = 4;
  constexpr int RPP = 256 / GS;  // rows gathered per pass
  constexpr int NP = TM / RPP;   // passes (GS = 16: 1, 32: 2, 64: 4)
  const int tid = threadIdx.x & 255;
  const int c0 = (tid % GS) * VEC;
  const int lane = threadIdx.x & 63, w = tid >> 6;
  const int nn = lane & 15, kq = lane >> 4;
  const int K = S.ncols;
  KT(8);
  constexpr int WB = 48;
  if (valid) {
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int m = p * RPP + tid / GS;
    const int row = row0 + m;
    const bool live = row < row_end;
    int rb[AGG_MAX_IN], re[AGG_MAX_IN];
#pragma unroll
    for (int oi = 0; oi < AGG_MAX_IN; ++oi) {
      rb[oi] = re[oi] = 0;
      if (live && oi < S.n_out) { rb[oi] = S.out[oi].t_rowptr[row]; re[oi] = S.out[oi].t_rowptr[row + 1]; }
    }
    // ELL ids of the first pair in the round trip of the extents (see agg_row); a row past the end reads the tile's first row
    const int erow = live ? row : row0;
    EllRow<GS <= 16> hA, hB;
    bool ell0 = false;
    {
      const TAggOut& A0 = S.out[0];
      const TAggOut& A1 = S.out[S.n_out > 1 ? 1 : 0];
      ell0 = S.n_out > 0 && A0.t_ell && A1.t_ell;  // block-uniform
      if (ell0) {
        hA.fetch(A0.t_ell, erow);
        hB.fetch(A1.t_ell, erow);
      }
    }
    // outgoing edge types in PAIRS (see agg_row): ids of both, then 1/deg + gradient rows of both
    constexpr int UB = 8;
#pragma unroll
    for (int oi = 0; oi < AGG_MAX_IN; oi += 2) {
      if (oi >= S.n_out) break;
      const bool has2 = oi + 1 < S.n_out;
      const TAggOut& O0 = S.out[oi];
      const TAggOut& O1 = S.out[has2 ? oi + 1 : oi];
      const int b0 = rb[oi], e0 = re[oi];
      const int b1 = has2 ? rb[oi + 1] : b0, e1 = has2 ? re[oi + 1] : b0;
      // GS <= 16: the ids of out-edges 8..15 travel with those of 0..7 (see agg_row): one round trip less for rows of 9..16 edges
      constexpr bool PRE = GS <= 16;
      int i0[UB], i1[UB], i0t[PRE ? UB : 1], i1t[PRE ? UB : 1];
      if (oi == 0 ? ell0 : (O0.t_ell && O1.t_ell)) {  // block-uniform: ids by row address (see agg_row)
        if (oi != 0) {
          hA.fetch(O0.t_ell, erow);
          hB.fetch(O1.t_ell, erow);
        }
        hA.ids(e0 - b0, i0, i0t);
        hB.ids(e1 - b1, i1, i1t);
      } else {
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        i0[u] = O0.t_col[e0 > b0 ? min(b0 + u, e0 - 1) : 0];
        i1[u] = O1.t_col[e1 > b1 ? min(b1 + u, e1 - 1) : 0];
        if constexpr (PRE) {
          i0t[u] = O0.t_col[e0 > b0 ? min(b0 + UB + u, e0 - 1) : 0];
          i1t[u] = O1.t_col[e1 > b1 ? min(b1 + UB + u, e1 - 1) : 0];
        }
      }
      }
      float d0[UB], d1[UB];
      Acc<VEC> v0[UB], v1[UB];
      const int cc0 = c0 < O0.F ? c0 : 0, cc1 = c0 < O1.F ? c0 : 0;
      const bool dg = a.mean && O0.degf && O1.degf;  // block-uniform
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        d0[u] = dg ? O0.degf[i0[u]] : 1.f;
        d1[u] = dg ? O1.degf[i1[u]] : 1.f;
        v0[u].load(O0.g + (int64_t)i0[u] * O0.ldg + cc0);
        v1[u].load(O1.g + (int64_t)i1[u] * O1.ldg + cc1);
      }
      if (a.mean && !dg) {  // unit-test path without the plan's degree table
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          const int g0 = O0.rowptr[i0[u] + 1] - O0.rowptr[i0[u]], g1 = O1.rowptr[i1[u] + 1] - O1.rowptr[i1[u]];
          d0[u] = 1.f / (float)(g0 > 1 ? g0 : 1);
          d1[u] = 1.f / (float)(g1 > 1 ? g1 : 1);
        }
      }
      Acc<VEC> a0[1], a1[1];
      a0[0].zero();
      a1[0].zero();
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (b0 + u < e0) a0[0].add_mul(v0[u], d0[u]);
#pragma unroll
      for (int u = 0; u < UB; ++u)
        if (b1 + u < e1) a1[0].add_mul(v1[u], d1[u]);
      int done = UB;
      if constexpr (PRE) {
        if (e0 - b0 > UB || e1 - b1 > UB) {  // second batch through the same registers (ids already here)
#pragma unroll
          for (int u = 0; u < UB; ++u) {
            d0[u] = dg ? O0.degf[i0t[u]] : 1.f;
            d1[u] = dg ? O1.degf[i1t[u]] : 1.f;
            v0[u].load(O0.g + (int64_t)i0t[u] * O0.ldg + cc0);
            v1[u].load(O1.g + (int64_t)i1t[u] * O1.ldg + cc1);
          }
          if (a.mean && !dg) {
#pragma unroll
            for (int u = 0; u < UB; ++u) {
              const int g0 = O0.rowptr[i0t[u] + 1] - O0.rowptr[i0t[u]], g1 = O1.rowptr[i1t[u] + 1] - O1.rowptr[i1t[u]];
              d0[u] = 1.f / (float)(g0 > 1 ? g0 : 1);
              d1[u] = 1.f / (float)(g1 > 1 ? g1 : 1);
            }
          }
#pragma unroll
          for (int u = 0; u < UB; ++u)
            if (b0 + UB + u < e0) a0[0].add_mul(v0[u], d0[u]);
#pragma unroll
          for (int u = 0; u < UB; ++u)
            if (b1 + UB + u < e1) a1[0].add_mul(v1[u], d1[u]);
        }
        done = 2 * UB;
      }
      if (e0 - b0 > done) gather_sum_w<GS, 1, VEC>(a0, O0.g, O0.ldg, O0.t_col, O0.rowptr, O0.degf, a.mean, b0 + done, e0, c0, O0.F);
      if (e1 - b1 > done) gather_sum_w<GS, 1, VEC>(a1, O1.g, O1.ldg, O1.t_col, O1.rowptr, O1.degf, a.mean, b1 + done, e1, c0, O1.F);
      if (c0 < O0.F) {
        if (live) a0[0].store(S.dz + (int64_t)row * S.lddz + O0.coff + c0);
#pragma unroll
        for (int i = 0; i < 4; ++i) Hs[(O0.coff + c0 + i) * LDH + m] = a0[0].at(i);
      }
      if (has2 && c0 < O1.F) {
        if (live) a1[0].store(S.dz + (int64_t)row * S.lddz + O1.coff + c0);
#pragma unroll
        for (int i = 0; i < 4; ++i) Hs[(O1.coff + c0 + i) * LDH + m] = a1[0].at(i);
      }
    }
    if (S.groot && c0 < S.Froot) {
      Acc<VEC> v;
      v.zero();
      if (live) {
        v.load(S.groot + (int64_t)row * S.ldgr + c0);
        v.store(S.dz + (int64_t)row * S.lddz + S.roff + c0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) Hs[(S.roff + c0 + i) * LDH + m] = v.at(i);
    }
  }
  }
  __syncthreads();
  if (!valid || S.xw == nullptr) return;  // group-uniform
  KT(9);
  const int n_ct = (S.xN + 15) >> 4;
  for (int ct = w; ct < n_ct; ct += 4) {
    const int col = ct * 16 + nn;
    const float* wp = S.xw + min(col, S.xN - 1) + (int64_t)kq * S.xldw;  // clamped: padded columns are never stored
    const float* hp = Hs + kq * LDH + nn;                                  // A operand: row m = lane & 15 of the tile
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // activations whose derivative masks this tile's output: requested before the weight loads / MFMA chain instead of after
    float hv[4] = {0.f, 0.f, 0.f, 0.f};
    if (S.xh) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        hv[r] = S.xh[(int64_t)min(row0 + kq * 4 + r, row_end - 1) * S.xldh + min(col, S.xN - 1)];
    }
    // 48 k-steps (192 stacked columns) per trip: the (clamped) weight loads are all in flight together, then the MFMA
    // chain runs -- one L2 round trip for the typical stacked width.  (Requesting them before the gather was measured
    // SLOWER, +1 us: 48 scalar loads per lane queue in front of the gather's own loads.)
    for (int kb = 0; kb < K; kb += 4 * WB) {
      float bv[WB];
#pragma unroll
      for (int u = 0; u < WB; ++u) bv[u] = wp[(int64_t)min(kb + 4 * u, K - 4) * S.xldw];
#pragma unroll
      for (int u = 0; u < WB; ++u)
        if (kb + 4 * u < K) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hp[(kb + 4 * u) * LDH], bv[u], acc, 0, 0, 0);
    }
    // D layout: column = lane & 15, row = (lane >> 4) * 4 + reg
    if (col < S.xN) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = row0 + kq * 4 + r;
        if (row >= row_end) continue;
        float v = acc[r];
        if (S.xh) {
          const float h = hv[r];
          // dropped elements were stored as -0.0f by the forward: the keep bit is the sign of a zero
          const bool keep = !S.xdrop_on || (__float_as_uint(h) != 0x80000000u);
          float f = S.xscale;
          if (!keep) f = 0.f;
          else if (S.xact == HMP_ACT_RELU) f = h > 0.f ? S.xscale : 0.f;
          else if (S.xact == HMP_ACT_ELU) f = h > 0.f ? S.xscale : (h + S.xscale);
          v *= f;
        }
        S.xg[(int64_t)row * S.xldg + col] = v;
      }
    }
  }
  KT(10);
}

template <int GS>
__global__ __launch_bounds__(256) void agg_bwd_dx_kernel(const TAggArgs a) {
  extern __shared__ float Hs[];  // [ncols][17]
  if ((int)blockIdx.x == a.total_blocks) {
    finalize_loss(a.fin_row_lv, a.fin_rows, a.fin_out2, a.fin_state);
    return;
  }
  KT_SPAN_BEGIN(48);
  int si = 0;
  while (si + 1 < a.n && (int)blockIdx.x >= a.bstart[si + 1]) ++si;
  karg_warm<10>((int)offsetof(TAggArgs, s) + si * (int)sizeof(TAggSrc), (int)sizeof(TAggSrc));
  const TAggSrc& S = a.s[si];
  const int row0 = ((int)blockIdx.x - S.block_start) * S.tile_rows;  // (tiles of 8 for entries of heavy rows: AggDst::tile_rows)
  agg_bwd_dx_tile<GS>(a, S, row0, min(row0 + S.tile_rows, S.n_rows), true, Hs);
  KT_SPAN_END(48, si);
}

#ifdef HMP_EXPERIMENTS  // measured 5x slower than the multi-launch sequence (profiles/r02_c_graph_local_chain.md): kept out of the product library
// ----- graph-local chain: every launch between the front kernel and the weight-gradient GEMM, in ONE launch ------------------
// A batch is a disjoint union of scene graphs: no edge crosses graphs, so from the first aggregation to the last transposed
// aggregation a graph depends on nothing but itself.  One 1024-thread workgroup owns one graph for ALL of those phases
// (aggregation + next projection per hidden layer, last aggregation + masked cross entropy, transposed aggregation +
// input-gradient GEMM per layer): what were 2L launches with a cold L2 behind every boundary become phases separated by a
// workgroup barrier, their operands written moments earlier by the same CU.  The phases ARE the tile / row routines of the
// kernels above (four groups of 256 threads take the graph's 16-row tiles in turn), so results are bit-identical to the
// multi-launch sequence (HMP_CHAIN=0).  The last workgroup to finish (an atomic ticket, no spinning) sums the per-row losses.
// locate tile k of a phase: entries in order, ceil(rows / 16) tiles each
template <class ARGS, class ENTRY>
__device__ __forceinline__ bool chain_tile(const ChainArgs& A, const ARGS& a, const ENTRY* ents, const int* types, int g, int k, int& ei,
                                           int& row0, int& row_end) {
  for (int i = 0; i < a.n; ++i) {
    const int64_t* pt = A.ptr[types[i]];
    const int r0 = (int)pt[g], r1 = (int)pt[g + 1];
    const int nt = (r1 - r0 + 15) >> 4;
    if (k < nt) { ei = i; row0 = r0 + 16 * k; row_end = r1; return true; }
    k -= nt;
  }
  ei = 0; row0 = 0; row_end = 0;
  return false;
}
template <class ARGS>
__device__ __forceinline__ int chain_tiles(const ChainArgs& A, const ARGS& a, const int* types, int g) {
  int n = 0;
  for (int i = 0; i < a.n; ++i) {
    const int64_t* pt = A.ptr[types[i]];
    n += ((int)pt[g + 1] - (int)pt[g] + 15) >> 4;
  }
  return n;
}

template <int GS>
__device__ __forceinline__ void chain_ce_rows(const ChainArgs& A, const AggArgs& a, const int* types, int g) {
  for (int i = 0; i < a.n; ++i) {
    const AggDst& D = a.d[i];
    const int64_t* pt = A.ptr[types[i]];
    const int r0 = (int)pt[g], r1 = (int)pt[g + 1];
    const int c0 = (threadIdx.x % GS) * 4;
    for (int row = r0 + (int)threadIdx.x / GS; row < r1; row += WIN_THREADS_CHAIN / GS) {
      Acc<4> tot[1];
      int64_t y = 0;
      if (D.ce_labels) y = D.ce_labels[row];
      agg_row<GS, 1>(D, a.mean, row, c0, tot);
      if (D.ce_labels) ce_rowgroup<GS>(D, a.state, row, c0, tot[0], y);
    }
  }
}
template <int GS>
__device__ __forceinline__ void chain_bwd_rows(const ChainArgs& A, const TAggArgs& a, const int* types, int g) {
  for (int i = 0; i < a.n; ++i) {
    const TAggSrc& S = a.s[i];
    const int64_t* pt = A.ptr[types[i]];
    const int r0 = (int)pt[g], r1 = (int)pt[g + 1];
    const int c0 = (threadIdx.x % GS) * 4;
    for (int row = r0 + (int)threadIdx.x / GS; row < r1; row += WIN_THREADS_CHAIN / GS) agg_bwd_row<GS, 1>(a, S, row, c0);
  }
}

template <int GS>
__global__ __launch_bounds__(WIN_THREADS_CHAIN) void chain_kernel(const ChainArgs* __restrict__ Ap, int n_graphs, int fin_rows) {
  extern __shared__ __attribute__((aligned(16))) float clds[];
  __shared__ int s_last;
  // the argument block (28 KB: the descriptors of every phase) is copied into LDS once -- read through the global pointer, every
  // descriptor field an aggregation routine touches would be a dependent memory round trip of its own (measured: 0.51 ms
  // for what five launches did in 0.05 ms); in the multi-launch kernels these fields sit in scalar registers
  constexpr int ARG_F = (int)((sizeof(ChainArgs) + 15) / 16) * 4;  // floats
  {
    const uint4* src = reinterpret_cast<const uint4*>(Ap);
    uint4* dst = reinterpret_cast<uint4*>(clds);
    for (int i = threadIdx.x; i < ARG_F / 4; i += WIN_THREADS_CHAIN) dst[i] = src[i];
  }
  __syncthreads();
  const ChainArgs& A = *reinterpret_cast<const ChainArgs*>(clds);
  const int g = blockIdx.x;
  const int grp = threadIdx.x >> 8;
  float* Hs = clds + ARG_F + (size_t)grp * A.lds_stride;
  const int L = A.L;
  // ---- forward
  for (int l = 0; l < L; ++l) {
    const AggArgs& a = A.fwd[l];
    if (l < L - 1) {
      const int T = chain_tiles(A, a, A.fwd_type[l], g);
      for (int k0 = 0; k0 < T; k0 += CHAIN_GROUPS) {
        int ei, row0, row_end;
        const bool valid = chain_tile(A, a, a.d, A.fwd_type[l], g, k0 + grp, ei, row0, row_end);
        agg_proj_tile<GS>(a, a.d[ei], row0, row_end, valid, Hs);
        __syncthreads();  // the group's LDS image is rewritten by its next tile
      }
    } else {
      if (A.gs_last == 8) chain_ce_rows<8>(A, a, A.fwd_type[l], g);
      else if (A.gs_last == 16) chain_ce_rows<16>(A, a, A.fwd_type[l], g);
      else chain_ce_rows<32>(A, a, A.fwd_type[l], g);
    }
    __syncthreads();  // phase boundary: this graph's rows of the layer are complete and visible to the whole workgroup
  }
  // ---- backward
  for (int l = L - 1; l >= 0; --l) {
    const TAggArgs& a = A.bwd[l];
    if (l > 0) {
      const int T = chain_tiles(A, a, A.bwd_type[l], g);
      for (int k0 = 0; k0 < T; k0 += CHAIN_GROUPS) {
        int ei, row0, row_end;
        const bool valid = chain_tile(A, a, a.s, A.bwd_type[l], g, k0 + grp, ei, row0, row_end);
        agg_bwd_dx_tile<GS>(a, a.s[ei], row0, row_end, valid, Hs);
        __syncthreads();
      }
    } else {
      if (A.gs_first == 8) chain_bwd_rows<8>(A, a, A.bwd_type[l], g);
      else if (A.gs_first == 16) chain_bwd_rows<16>(A, a, A.bwd_type[l], g);
      else chain_bwd_rows<32>(A, a, A.bwd_type[l], g);
    }
    __syncthreads();
  }
  // ---- the last workgroup to arrive sums the per-row {loss, valid} pairs of ALL graphs (fixed order: run-to-run identical)
  if (A.fin_row_lv) {
    if (threadIdx.x == 0) {
      __threadfence();  // this workgroup's row_lv stores are out before the ticket
      const unsigned t = atomicAdd(A.ticket, 1u);
      s_last = (t == (unsigned)n_graphs - 1u) ? 1 : 0;
      if (s_last) { *A.ticket = 0u; __threadfence(); }  // re-armed for the next step; acquire side of the hand-off
    }
    __syncthreads();
    if (s_last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      // the summation order of finalize_loss (256 strided partial sums, then a tree): the same bits as the multi-launch path
      float l = 0.f, v = 0.f;
      const float* lv = A.fin_row_lv;
      if (threadIdx.x < 256) {
        for (int r = threadIdx.x; r < fin_rows; r += 256) {
          l += __hip_atomic_load(lv + 2 * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          v += __hip_atomic_load(lv + 2 * r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      float* sl = clds + ARG_F;
      float* sv = clds + ARG_F + 256;
      if (threadIdx.x < 256) { sl[threadIdx.x] = l; sv[threadIdx.x] = v; }
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sv[threadIdx.x] += sv[threadIdx.x + o]; }
        __syncthreads();
      }
      if (threadIdx.x == 0) {
        A.fin_out2[0] = sl[0];
        A.fin_out2[1] = sv[0];
        if (A.fin_state) { A.fin_state->loss_sum = sl[0]; A.fin_state->count = sv[0]; }
      }
    }
  }
}
#endif  // HMP_EXPERIMENTS


// ----- LDS sliding-window aggregation for the 10^6-row regime (bf16 rows of 256 elements = 512 bytes) ------------------------
// Scene graphs are local: an object's neighbours are objects of the same room, and a graph builder numbers the objects of a
// room consecutively, so the source rows that a run of consecutive destination rows gathers lie in a narrow index window
// around the run (config 5: 90 % of a row's 16 neighbours inside its room's ~100 rows).  The one-wave-per-row kernels above
// fetch every neighbour row through L2 -> L1 -> registers once per EDGE (17 x 512 B per row, 9 GB per launch at 10^6 rows) and
// -- measured -- are bound by LATENCY x rows in flight, not by bytes: a row is a chain of 4 dependent round trips (extent ->
// neighbour ids -> rows -> more rows) of ~2 us each with ~24 rows in flight per CU.
//
// Here ONE persistent 1024-thread workgroup per CU walks a contiguous range of destination rows in chunks of WR = 64 and keeps
// a RING of WRING = 256 source rows in LDS (128 KiB): the chunk's own rows plus WM = 64 rows of margin either side.  Moving on
// one chunk brings exactly WR new rows in -- into the ring slots the new window no longer covers -- so every source row enters
// the CU ONCE per launch, with coalesced 16-byte loads issued one chunk ahead (they land while the current chunk computes).
// The chunk's CSR slice (row extents and neighbour ids of every incoming edge type: one contiguous range per type) is staged
// in LDS the same way, two / one chunks ahead.  A row then costs LDS reads only, except for the neighbours outside the window
// (the 10 % global edges; the other edge types, e.g. the room row of an object).  Those are requested FIRST, four rows at a time
// (buffer loads at a scalar offset; a descriptor of zero records where there is no such edge), the in-window edges follow in
// batches of eight LDS reads at scalar-computed slots (an out-of-window edge reads the all-zero row there), and the requested rows
// are added LAST (round 3; rounds 1-2 mixed both kinds inside a batch and waited for HBM per batch).  One accumulator per edge type;
// in-window rows in edge order, then the others in edge order.  Only the edge type whose source index space IS the destination's
// (objects -> objects) is windowed -- for bipartite types every source row is used once and staging buys nothing.
constexpr int WIN_THREADS = 1024;
constexpr int WIN_WAVES = WIN_THREADS / 64;
constexpr int WIN_ROW_BYTES = 512;  // 256 bf16
constexpr int WR = 64;              // destination rows per chunk
constexpr int WM = 64;              // margin rows either side
constexpr int WRING = 256;          // ring rows = WR + 2 WM + WR (power of two: slot = row & 255)
constexpr int WG = WR / WIN_WAVES;  // rows per wave and chunk (4)
constexpr int WIDCAP = 3072;        // neighbour ids per chunk staged in LDS (all incoming edge types; avg 64 x 17 = 1088)
constexpr int WRP = WR + 1;
constexpr int WIN_LDS_RING = (WRING + 1) * WIN_ROW_BYTES;  // + one all-zero row (slot WRING): what an out-of-window edge reads from LDS
constexpr int WIN_LDS_IDS = 2 * WIDCAP * 4;
constexpr int WIN_LDS_RP = 3 * AGG_MAX_IN * WRP * 4;
constexpr int WIN_LDS_DEG = WRING * 4;  // backward: reciprocal in-degrees of the ring rows
constexpr int WIN_LDS_FWD = WIN_LDS_RING + WIN_LDS_IDS + WIN_LDS_RP;
constexpr int WIN_LDS_BWD = WIN_LDS_FWD + WIN_LDS_DEG;
static_assert(WIN_LDS_BWD <= 160 * 1024, "LDS budget");
constexpr unsigned WIN_SKIP_OFF = 0xFFFFF000u;  // buffer offset beyond any record: the load returns 0 without touching memory
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// raw buffer over [base, base + bytes): an out-of-range offset reads as zero (no memory access) -- lets a batch of edges issue BOTH
// an LDS read and a global read per edge without a branch: the one that does not apply is aimed at the zero row / out of range
__device__ __forceinline__ __amdgpu_buffer_rsrc_t win_rsrc(const void* base, unsigned bytes) { return buf_rsrc(base, bytes); }

static_assert(AGG_MAX_IN == 6 && WG == 4, "sel_q / sel_root are written for 6 / 4 entries");
struct WinFwd {
  AggDst d;  // ONE destination entry, win_in >= 0
  int mean;
  int n_chunks, chunks_per_block;
  int dbg;   // HMP_WIN_DBG=1 (measurement only, wrong results): no edge reads global memory
};
struct WinBwd {
  TAggSrc s;  // ONE source entry, win_out >= 0
  int mean, dzb16;
  int n_chunks, chunks_per_block;
};

__device__ __forceinline__ void widen_bf16x4(Acc<4>& a, const uint2 b) {
  a.v = make_float4(__uint_as_float(b.x << 16), __uint_as_float(b.x & 0xffff0000u), __uint_as_float(b.y << 16),
                    __uint_as_float(b.y & 0xffff0000u));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint2 sel_root(const uint2 (&s)[WG], int g) {  // g wave-uniform: a scalar branch tree
  switch (g) {
    case 0: return s[0];
    case 1: return s[1];
    case 2: return s[2];
    default: return s[3];
  }
}
__device__ __forceinline__ int sel_st(const int (&s)[AGG_MAX_IN + 1], int q) {  // q wave-uniform
  switch (q) {
    case 0: return s[0];
    case 1: return s[1];
    case 2: return s[2];
    case 3: return s[3];
    case 4: return s[4];
    case 5: return s[5];
    default: return s[6];
  }
}
__device__ __forceinline__ int sel_q(const int (&s)[AGG_MAX_IN], int q) {  // q wave-uniform
  switch (q) {
    case 0: return s[0];
    case 1: return s[1];
    case 2: return s[2];
    case 3: return s[3];
    case 4: return s[4];
    default: return s[5];
  }
}
// Shared chunk pipeline of the forward and the transposed kernel.  `Side` supplies, per edge type q of the entry: the extent
// array (rowptr), the id array (col), the gathered matrix (base pointer as bf16 elements, column offset, pitch) and, for the
// transposed side, the per-row weights.
struct WinStage {  // registers holding what phase A requested for the chunks ahead
  uint4 rows[(WR * 32) / WIN_THREADS];          // 2: the WR new ring rows of chunk c + 1
  int ids[(WIDCAP + WIN_THREADS - 1) / WIN_THREADS];  // 3: ids of chunk c + 1
  int rp;                                        // extents of chunk c + 2
  float deg;                                     // backward: reciprocal degrees of the new ring rows
};

template <class E>  // E: AggDst (forward: in[].rowptr / col) or TAggSrc (transposed: out[].t_rowptr / t_col)
struct WinTopo;
template <>
struct WinTopo<AggDst> {
  static __device__ __forceinline__ int n(const AggDst& D) { return D.n_in; }
  static __device__ __forceinline__ const int* rowptr(const AggDst& D, int q) { return D.in[q].rowptr; }
  static __device__ __forceinline__ const int* col(const AggDst& D, int q) { return D.in[q].col; }
  static __device__ __forceinline__ int rows(const AggDst& D) { return D.n_rows; }
};
template <>
struct WinTopo<TAggSrc> {
  static __device__ __forceinline__ int n(const TAggSrc& S) { return S.n_out; }
  static __device__ __forceinline__ const int* rowptr(const TAggSrc& S, int q) { return S.out[q].t_rowptr; }
  static __device__ __forceinline__ const int* col(const TAggSrc& S, int q) { return S.out[q].t_col; }
  static __device__ __forceinline__ int rows(const TAggSrc& S) { return S.n_rows; }
};

// extents of chunk `ch` -> register (thread t < n * WRP owns element t)
template <class E>
__device__ __forceinline__ int win_load_rp(const E& X, int ch, int n_chunks) {
  const int t = threadIdx.x, nq = WinTopo<E>::n(X);
  if (ch >= n_chunks || t >= nq * WRP) return 0;
  const int q = t / WRP, i = t - q * WRP;
  const int r = min(ch * WR + i, WinTopo<E>::rows(X));
  return WinTopo<E>::rowptr(X, q)[r];
}
// ids of chunk `ch` (extents already in LDS buffer rp) -> registers; returns the id total of the chunk through *total
template <class E>
__device__ __forceinline__ void win_load_ids(const E& X, const int* rp, bool live, int (&ids)[(WIDCAP + WIN_THREADS - 1) / WIN_THREADS]) {
  const int nq = WinTopo<E>::n(X);
  int start[AGG_MAX_IN + 1], eb[AGG_MAX_IN];
  start[0] = 0;
#pragma unroll
  for (int q = 0; q < AGG_MAX_IN; ++q) {
    eb[q] = 0;
    int len = 0;
    if (live && q < nq) { eb[q] = rp[q * WRP]; len = rp[q * WRP + WR] - eb[q]; }
    start[q + 1] = start[q] + len;
  }
  const bool fits = start[AGG_MAX_IN] <= WIDCAP;
#pragma unroll
  for (int it = 0; it < (WIDCAP + WIN_THREADS - 1) / WIN_THREADS; ++it) {
    const int p = (int)threadIdx.x + it * WIN_THREADS;
    ids[it] = 0;
    if (fits && p < start[AGG_MAX_IN]) {
      const int* cq = WinTopo<E>::col(X, 0);
      int ebq = eb[0], stq = 0;
#pragma unroll
      for (int t = 1; t < AGG_MAX_IN; ++t)
        if (t < nq && p >= start[t]) { cq = WinTopo<E>::col(X, t); ebq = eb[t]; stq = start[t]; }
      ids[it] = cq[ebq + (p - stq)];
    }
  }
}
// per-chunk view of the staged CSR slice: id offset of every edge type inside the id buffer; false: the chunk did not fit
__device__ __forceinline__ bool win_offsets(const int* rp, int nq, int (&off)[AGG_MAX_IN], int (&eb)[AGG_MAX_IN]) {
  int acc = 0;
#pragma unroll
  for (int q = 0; q < AGG_MAX_IN; ++q) {
    off[q] = acc;
    eb[q] = 0;
    if (q < nq) { eb[q] = rp[q * WRP]; acc += rp[q * WRP + WR] - eb[q]; }
  }
  return acc <= WIDCAP;
}

template <bool HB>
__global__ __launch_bounds__(WIN_THREADS) void agg_fwd_win_kernel(const WinFwd a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wlds[];
  unsigned char* ring = wlds;
  int* idbuf = reinterpret_cast<int*>(wlds + WIN_LDS_RING);                 // [2][WIDCAP]
  int* rpbuf = reinterpret_cast<int*>(wlds + WIN_LDS_RING + WIN_LDS_IDS);   // [3][AGG_MAX_IN][WRP]
  const AggDst& D = a.d;
  const int n_rows = D.n_rows, nq = D.n_in;
  const int c_begin = (int)blockIdx.x * a.chunks_per_block;
  const int c_end = min(c_begin + a.chunks_per_block, a.n_chunks);
  if (c_begin >= c_end) return;  // block-uniform
  const int wave = uni((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
  const AggIn& IW = D.in[D.win_in];
  const uint16_t* zwin = reinterpret_cast<const uint16_t*>(IW.z) + IW.coff;
  const int c0 = lane * 4;
  DropCfg dcfg = D.drop;
  if (D.drop_on) dcfg = drop_resolve(D.drop);
  Acc<4> biasv;  // the lane's 4 bias columns: the same for every row
  biasv.zero();
  if (D.bias) biasv.load(D.bias + c0);

  // ---- prologue: extents of chunks c_begin, c_begin + 1; ids of c_begin; the first window -------------------------------
  {
    const int r0 = win_load_rp(D, c_begin, a.n_chunks), r1 = win_load_rp(D, c_begin + 1, a.n_chunks);
    if ((int)threadIdx.x < nq * WRP) {
      rpbuf[(c_begin % 3) * AGG_MAX_IN * WRP + threadIdx.x] = r0;
      rpbuf[((c_begin + 1) % 3) * AGG_MAX_IN * WRP + threadIdx.x] = r1;
    }
    if ((int)threadIdx.x < 32) *reinterpret_cast<uint4*>(ring + (size_t)WRING * WIN_ROW_BYTES + threadIdx.x * 16) = make_uint4(0u, 0u, 0u, 0u);
    // rows [lo, hi) of the first window (everything the ring will hold for chunk c_begin)
    const int lo = max(c_begin * WR - WM, 0), hi = min(c_begin * WR + WR + WM, n_rows);
    for (int p = threadIdx.x; p < (hi - lo) * 32; p += WIN_THREADS) {
      const int r = lo + (p >> 5), piece = p & 31;
      const uint4 v = *reinterpret_cast<const uint4*>(zwin + (int64_t)r * IW.ldz + piece * 8);
      *reinterpret_cast<uint4*>(ring + (size_t)(r & (WRING - 1)) * WIN_ROW_BYTES + piece * 16) = v;
    }
    __syncthreads();
    int ids[(WIDCAP + WIN_THREADS - 1) / WIN_THREADS];
    win_load_ids(D, rpbuf + (c_begin % 3) * AGG_MAX_IN * WRP, true, ids);
#pragma unroll
    for (int it = 0; it < (WIDCAP + WIN_THREADS - 1) / WIN_THREADS; ++it) {
      const int p = (int)threadIdx.x + it * WIN_THREADS;
      if (p < WIDCAP) idbuf[(c_begin & 1) * WIDCAP + p] = ids[it];
    }
    __syncthreads();
  }

  for (int i = 20; i < 31; ++i) KT_ZERO(i);
  for (int ch = c_begin; ch < c_end; ++ch) {
    const int r_c = ch * WR;
    const int* rp = rpbuf + (ch % 3) * AGG_MAX_IN * WRP;
    const int* idc = idbuf + (ch & 1) * WIDCAP;
    [[maybe_unused]] const unsigned long long kt_a = KT_NOW();
    // ---- phase A: request what the NEXT chunks need (lands while this chunk computes) ------------------------------------
    WinStage stg;
    const bool next = ch + 1 < c_end;
    {
      const int nlo = r_c + WR + WM;  // new ring rows of chunk ch + 1: [nlo, nlo + WR)
#pragma unroll
      for (int it = 0; it < (WR * 32) / WIN_THREADS; ++it) {
        const int p = (int)threadIdx.x + it * WIN_THREADS;
        const int r = nlo + (p >> 5), piece = p & 31;
        stg.rows[it] = make_uint4(0u, 0u, 0u, 0u);
        if (next && r < n_rows) stg.rows[it] = *reinterpret_cast<const uint4*>(zwin + (int64_t)r * IW.ldz + piece * 8);
      }
      win_load_ids(D, rpbuf + ((ch + 1) % 3) * AGG_MAX_IN * WRP, next, stg.ids);
      stg.rp = (ch + 2 < c_end) ? win_load_rp(D, ch + 2, a.n_chunks) : 0;
    }
    KT_ADD(20, kt_a);
    [[maybe_unused]] const unsigned long long kt_c = KT_NOW();
    // ---- this chunk ----------------------------------------------------------------------------------------------------------
    int off[AGG_MAX_IN], eb[AGG_MAX_IN];
    const bool fits = win_offsets(rp, nq, off, eb);
    const int wlo = max(r_c - WM, 0), whi = min(r_c + WR + WM, n_rows);  // rows the ring holds now
    {  // (!fits, block-uniform: a chunk with more ids than the LDS slice holds reads extents / ids from global memory, no window)
      // Round 3: the counters had the waves of this kernel waiting 57 % of their cycles (SQ_WAIT_ANY / SQ_WAVE_CYCLES) and the vector
      // unit 56 % busy -- a row was a chain of ~7 dependent waits (extents, ids and row batches per edge type).  Now (a) the extents of
      // all of the wave's rows are ONE lane-parallel LDS read per chunk and reach the scalar unit by readlane; (b) the neighbour ids
      // of the NEXT row's window list and of its deferred list are requested while this row computes; (c) the edge type after the
      // window one (rooms -> objects: one or two global rows) has its rows requested BEFORE the window batches and is added after
      // them -- same accumulators, same order of additions (the out-of-window rows of the window list are added behind the in-window
      // ones since the far-first change below: fp32, mostly the same bits).
      const int WQ = D.win_in, DQ = (fits && WQ + 1 < nq) ? WQ + 1 : -1;  // wave-uniform
      int rpv = 0;  // lane 8 q + 2 g + h: extent h of the wave's g-th row, edge type q
      if (fits && (lane >> 3) < nq) rpv = rp[(lane >> 3) * WRP + ((lane & 7) >> 1) * WIN_WAVES + wave + (lane & 1)];
      const int offW = sel_q(off, WQ) - sel_q(eb, WQ), offD = DQ >= 0 ? sel_q(off, DQ) - sel_q(eb, DQ) : 0;  // + extent = id slot
      auto ids_of = [&](int g, int q, int offq) {  // ids 0..63 of row g's list of edge type q (staged chunk only)
        const int b = __builtin_amdgcn_readlane(rpv, q * 8 + g * 2), e = __builtin_amdgcn_readlane(rpv, q * 8 + g * 2 + 1);
        return lane < e - b ? idc[offq + b + lane] : 0;
      };
      int idw = 0, idd = 0;
      if (fits) {
        idw = ids_of(0, WQ, offW);
        if (DQ >= 0) idd = ids_of(0, DQ, offD);
      }
      // phase B: the root row travels one row ahead as well
      auto root_of = [&](int g) {
        const int row = r_c + g * WIN_WAVES + wave;
        uint2 r = make_uint2(0u, 0u);
        if (g < WG && row < n_rows && D.zroot)
          r = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(D.zroot) + (int64_t)row * D.ldzr + D.roff + c0);
        return r;
      };
      uint2 root = root_of(0);
      // phase C: sums in edge order, one accumulator per edge type.  The neighbour id of an edge is wave-uniform (readlane), so
      // its row comes either from the LDS ring (one ds_read_b64 at a scalar-computed slot) or from global memory (scalar base +
      // lane offset): a scalar branch per edge, no per-lane address arithmetic, 16 loads in flight per batch.
#pragma unroll 1
      for (int g = 0; g < WG; ++g) {
        const int rl = g * WIN_WAVES + wave, row = r_c + rl;
        if (row >= n_rows) break;  // rows grow with g
        [[maybe_unused]] const unsigned long long kt_r = KT_NOW();
        const uint2 root_n = root_of(g + 1);
        int idw_n = 0, idd_n = 0;  // (b) the next row's ids: in flight while this row's batches run
        if (fits && g + 1 < WG) {
          idw_n = ids_of(g + 1, WQ, offW);
          if (DQ >= 0) idd_n = ids_of(g + 1, DQ, offD);
        }
        // (c) deferred edge type: a list of one or two rows is requested here and consumed after the window list
        u32x2 vd[2] = {{0u, 0u}, {0u, 0u}};
        int cntD = 0;
        if (DQ >= 0) {
          const int bD = __builtin_amdgcn_readlane(rpv, DQ * 8 + g * 2), eD = __builtin_amdgcn_readlane(rpv, DQ * 8 + g * 2 + 1);
          if (eD - bD >= 1 && eD - bD <= 2) {
            cntD = eD - bD;
            const AggIn& I = D.in[DQ];
            const uint16_t* zq = reinterpret_cast<const uint16_t*>(I.z) + I.coff;
            const __amdgpu_buffer_rsrc_t rs = win_rsrc(zq, (unsigned)I.n_src * (unsigned)(I.ldz * 2) - (unsigned)(I.coff * 2));
            const __amdgpu_buffer_rsrc_t rs_null = win_rsrc(zq, 0u);
            const int gov = lane < cntD ? idd * (I.ldz * 2) : 0;
            vd[0] = __builtin_amdgcn_raw_buffer_load_b64(rs, lane * 8, __builtin_amdgcn_readlane(gov, 0), 0);
            vd[1] = __builtin_amdgcn_raw_buffer_load_b64(cntD > 1 ? rs : rs_null, lane * 8, __builtin_amdgcn_readlane(gov, 1), 0);
          }
        }
        Acc<4> tot;
        tot.zero();
        if (D.zroot) widen_bf16x4(tot, root);
        tot.add(biasv);
#pragma unroll 1
        for (int q = 0; q < nq; ++q) {
          const AggIn& I = D.in[q];
          if (q == DQ && cntD) {  // the deferred list: its rows have landed behind the window batches
            Acc<4> acc, w;
            acc.zero();
            widen_bf16x4(w, make_uint2(vd[0][0], vd[0][1]));
            acc.add(w);
            widen_bf16x4(w, make_uint2(vd[1][0], vd[1][1]));
            acc.add(w);
            tot.add_div(acc, a.mean ? (float)cntD : 1.f);
            continue;
          }
          const int b = fits ? __builtin_amdgcn_readlane(rpv, q * 8 + g * 2) : I.rowptr[row];
          const int e = fits ? __builtin_amdgcn_readlane(rpv, q * 8 + g * 2 + 1) : I.rowptr[row + 1];
          if (e == b) continue;
          const uint16_t* zq = reinterpret_cast<const uint16_t*>(I.z) + I.coff;
          const int ldq = I.ldz;
          const __amdgpu_buffer_rsrc_t rs = win_rsrc(zq, (unsigned)I.n_src * (unsigned)(ldq * 2) - (unsigned)(I.coff * 2));
          const __amdgpu_buffer_rsrc_t rs_null = win_rsrc(zq, 0u);
          const int offq = fits ? sel_q(off, q) + (b - sel_q(eb, q)) : 0;
          const bool wq = fits && q == WQ;
          Acc<4> acc;
          acc.zero();
          for (int base = 0; base < e - b; base += 64) {
            const int cnt = min(e - b - base, 64);
            int idv = 0;
            if (wq && base == 0) idv = idw;  // requested one row ago
            else if (lane < cnt) idv = fits ? idc[offq + base + lane] : I.col[b + base + lane];
            // Round 3: a row's OUT-OF-WINDOW rows (and the rows of an edge type without a window) are requested first, four at a time,
            // and added last; every batch in between reads LDS only (an out-of-window edge reads the all-zero row there).  Before, an
            // out-of-window edge sat inside its batch of eight -- LDS read OR buffer read per edge, branch-free -- and the batch waited
            // for a random 512-byte row from HBM before the next batch was even issued (profiles/r03_n_matrix_pipe_window_sum.md).
            // fp32 sums: in-window rows in edge order, then the others in edge order -- another association than the plain kernels'.
            const bool inw = lane < cnt && wq && idv >= wlo && idv < whi;
            const int lov = inw ? (idv & (WRING - 1)) * WIN_ROW_BYTES : WRING * WIN_ROW_BYTES;
            const int gov = idv * (ldq * 2);
            unsigned long long fm = __ballot(lane < cnt && !inw) & (a.dbg ? 0ull : ~0ull);  // (dbg: measurement only -- no global reads)
            u32x2 fv[4];
            auto far_issue = [&]() {  // the next (up to) four set bits of fm
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                const int u = fm ? __builtin_ctzll(fm) : 0;
                fv[t] = __builtin_amdgcn_raw_buffer_load_b64(fm ? rs : rs_null, lane * 8, __builtin_amdgcn_readlane(gov, u), 0);
                fm &= fm - 1ull;  // (0 stays 0)
              }
            };
            auto far_add = [&]() {
#pragma unroll
              for (int t = 0; t < 4; ++t) {
                Acc<4> w;
                widen_bf16x4(w, make_uint2(fv[t][0], fv[t][1]));
                acc.add(w);
              }
            };
            const bool any_far = fm != 0ull;
            if (any_far) far_issue();
            if (wq) {
              auto batch = [&](int u0, auto nb) {
                constexpr int NB = decltype(nb)::value;
                uint2 va[NB];
#pragma unroll
                for (int t = 0; t < NB; ++t) va[t] = *reinterpret_cast<const uint2*>(ring + __builtin_amdgcn_readlane(lov, u0 + t) + lane * 8);
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                  Acc<4> w;
                  widen_bf16x4(w, va[t]);
                  acc.add(w);
                }
              };
              int u0 = 0;
              for (; cnt - u0 > 2; u0 += 8) batch(u0, std::integral_constant<int, 8>());
              if (u0 < cnt) batch(u0, std::integral_constant<int, 2>());
            }
            if (any_far) {
              far_add();
              while (fm) {  // more than four: further rounds, each waited for
                far_issue();
                far_add();
              }
            }
          }
          tot.add_div(acc, a.mean ? (float)(e - b) : 1.f);
        }
        KT_ADD(29, kt_r);
        bool keep[4] = {true, true, true, true};
        if (D.drop_on) drop_keep4(dcfg, (uint32_t)row * (uint32_t)(D.ldo >> 2) + (uint32_t)(c0 >> 2), keep);
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // as agg_row's epilogue
          float v = tot.at(i);
          if (D.act == HMP_ACT_RELU) v = v > 0.f ? v : 0.f;
          else if (D.act == HMP_ACT_ELU) v = v > 0.f ? v : expm1f(v);
          if (D.drop_on) v = keep[i] ? (v * D.drop.scale + 0.0f) : -0.0f;
          tot.at(i) = v;
        }
        store_z<HB>(tot, D.out, (int64_t)row * D.ldo + c0);
        KT_ADD(30, kt_r);
        idw = idw_n;
        idd = idd_n;
        root = root_n;
      }
    }
    KT_ADD(21, kt_c);
    [[maybe_unused]] const unsigned long long kt_d = KT_NOW();
    // ---- phase D: what phase A requested goes to LDS (ring slots / buffers this chunk did not read) -------------------------
    if (next) {
      const int nlo = r_c + WR + WM;
#pragma unroll
      for (int it = 0; it < (WR * 32) / WIN_THREADS; ++it) {
        const int p = (int)threadIdx.x + it * WIN_THREADS;
        const int r = nlo + (p >> 5), piece = p & 31;
        if (r < n_rows) *reinterpret_cast<uint4*>(ring + (size_t)(r & (WRING - 1)) * WIN_ROW_BYTES + piece * 16) = stg.rows[it];
      }
#pragma unroll
      for (int it = 0; it < (WIDCAP + WIN_THREADS - 1) / WIN_THREADS; ++it) {
        const int p = (int)threadIdx.x + it * WIN_THREADS;
        if (p < WIDCAP) idbuf[((ch + 1) & 1) * WIDCAP + p] = stg.ids[it];
      }
      if (ch + 2 < c_end && (int)threadIdx.x < nq * WRP) rpbuf[((ch + 2) % 3) * AGG_MAX_IN * WRP + threadIdx.x] = stg.rp;
    }
    KT_ADD(22, kt_d);
    [[maybe_unused]] const unsigned long long kt_b = KT_NOW();
    __syncthreads();
    KT_ADD(23, kt_b);
    KT_ADD(24, kt_a);
  }
}

#ifdef HMP_EXPERIMENTS
// ---------------------------------------------------------------------------------------------------------------------------
// EXPERIMENT (make EXPERIMENTS=1, HMP_AGG_W4=1; correct -- tests/test_gpu_fusion.py, test_gpu_config5.py pass with it -- and SLOWER than
// the per-edge kernel: 1.8 against 1.12 ms per launch; stamps and the two earlier forms: profiles/r03_n_matrix_pipe_window_sum.md).
// Round 3, third form of the matrix-pipe window sum (agg_fwd_w4_kernel): the per-edge kernel's ROW-PER-WAVE layout for everything per
// row, the product only for the in-window sum -- without any hand-over between waves.  v_mfma_f32_4x4x4_16b_bf16 multiplies 16
// independent 4 x 4 x 4 blocks per wave; with the SAME A block in all of them (the count-matrix rows of the wave's 4 destination
// rows: lane 4 b + i supplies row i) and B block b = 4 ring slots x features 4 b .. 4 b + 3 (one ds_read_b64_tr_b16 per lane from the
// ring's natural [slot][feature] image), one instruction adds 4 slots into 4 rows x 64 features: lane l = feature 64 c + l of call c,
// register i = row i (tools/microbench/mfma4_layout.hip checks this layout on the hardware).  48 k steps x 4 calls per chunk and
// wave cover the window; only k steps in which one of the wave's rows has an edge are executed (a 48-bit mask built with the counts).
// A lane then owns features l, 64 + l, 128 + l, 192 + l of each row: every row access is a 128-byte run per 64-feature slice.
// Out-of-window edges (a per-row far table built with the counts) and the other edge types are requested per slice for all 4 rows.
// LDS: ring 96 KB (192 slots = the window; the next chunk's rows wait in registers) + counts 25 KB + ids / extents / far table.
// Sums: slot order + far edges in edge order (fp32; another association).
constexpr int W4S = 192, W4_SUB = W4S * 256, W4_RING = 2 * W4_SUB, W4_CP = 400, W4_CBYTES = WR * W4_CP, W4_FARCAP = 8;
constexpr int W4_LDS_FAR = WR * W4_FARCAP * 4 + WR * 4 + WIN_WAVES * 8;  // far ids, far counts, k-step masks
constexpr int W4_LDS = W4_RING + W4_CBYTES + WIN_LDS_IDS + W4_LDS_FAR + WIN_LDS_RP;
static_assert(W4_LDS <= 160 * 1024 && WM + WR + WM == W4S && WG == 4 && (W4_RING + W4_CBYTES + WIN_LDS_IDS) % 16 == 0, "LDS budget / window = ring / 4 rows per wave");
typedef short w4_s16x4 __attribute__((ext_vector_type(4)));
typedef float w4_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned char* w4_lds_ptr_t;
__device__ __forceinline__ uint32_t w4_lds_addr(const void* p) { return (uint32_t)(uintptr_t)(w4_lds_ptr_t)p; }
__device__ __forceinline__ int w4_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }
__device__ __forceinline__ w4_s16x4 w4_read_tr(uint32_t addr) {
  w4_s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}
__device__ __forceinline__ void w4_barrier() {  // LDS operations done, then the barrier; vector-memory requests stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ int w4_slot(int r) {
  int s = r % W4S;
  return s < 0 ? s + W4S : s;
}
__device__ __forceinline__ unsigned char* w4_ring_at(unsigned char* ring, int slot, int piece) {  // 16-byte piece (0..31) of a ring row
  return ring + (piece >> 4) * W4_SUB + w4_off(slot, piece & 15);
}
template <bool HB>
__global__ __launch_bounds__(WIN_THREADS) void agg_fwd_w4_kernel(const WinFwd a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wlds[];
  unsigned char* ring = wlds;
  unsigned char* cmat = wlds + W4_RING;
  int* idbuf = reinterpret_cast<int*>(wlds + W4_RING + W4_CBYTES);                 // [2][WIDCAP]
  int* farid = reinterpret_cast<int*>(wlds + W4_RING + W4_CBYTES + WIN_LDS_IDS);   // [WR][W4_FARCAP]
  int* farcnt = farid + WR * W4_FARCAP;                                            // [WR]
  unsigned long long* kmask = reinterpret_cast<unsigned long long*>(farcnt + WR);  // [WIN_WAVES]: k steps (of 4 slots) with an edge into the wave's rows
  int* rpbuf = reinterpret_cast<int*>(kmask + WIN_WAVES);                          // [3][AGG_MAX_IN][WRP]
  const AggDst& D = a.d;
  const int n_rows = D.n_rows, nq = D.n_in;
  const int c_begin = (int)blockIdx.x * a.chunks_per_block;
  const int c_end = min(c_begin + a.chunks_per_block, a.n_chunks);
  if (c_begin >= c_end) return;  // block-uniform
  const int wave = uni((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int WQ = D.win_in;
  const AggIn& IW = D.in[WQ];
  const uint16_t* zwin = reinterpret_cast<const uint16_t*>(IW.z) + IW.coff;
  DropCfg dcfg = D.drop;
  if (D.drop_on) dcfg = drop_resolve(D.drop);
  const uint32_t t16 = dcfg.thresh >> 16;
  float biasv[4] = {0.f, 0.f, 0.f, 0.f};
  if (D.bias) {
#pragma unroll
    for (int c = 0; c < 4; ++c) biasv[c] = D.bias[64 * c + lane];
  }
  auto zero_c = [&]() {
    for (int p = threadIdx.x; p < W4_CBYTES / 16; p += WIN_THREADS) *reinterpret_cast<uint4*>(cmat + p * 16) = make_uint4(0u, 0u, 0u, 0u);
    if ((int)threadIdx.x < WIN_WAVES) kmask[threadIdx.x] = 0ull;
  };
  // ---- prologue ------------------------------------------------------------------------------------------------------------------------
  {
    const int r0 = win_load_rp(D, c_begin, a.n_chunks), r1 = win_load_rp(D, c_begin + 1, a.n_chunks);
    if ((int)threadIdx.x < nq * WRP) {
      rpbuf[(c_begin % 3) * AGG_MAX_IN * WRP + threadIdx.x] = r0;
      rpbuf[((c_begin + 1) % 3) * AGG_MAX_IN * WRP + threadIdx.x] = r1;
    }
    const int lo = c_begin * WR - WM;
    for (int p = threadIdx.x; p < W4S * 32; p += WIN_THREADS) {  // the whole first window; rows outside the matrix: zeros (every slot is multiplied)
      const int r = lo + (p >> 5), piece = p & 31;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (r >= 0 && r < n_rows) v = *reinterpret_cast<const uint4*>(zwin + (int64_t)r * IW.ldz + piece * 8);
      *reinterpret_cast<uint4*>(w4_ring_at(ring, w4_slot(r), piece)) = v;
    }
    zero_c();
    __syncthreads();
    int ids[(WIDCAP + WIN_THREADS - 1) / WIN_THREADS];
    win_load_ids(D, rpbuf + (c_begin % 3) * AGG_MAX_IN * WRP, true, ids);
#pragma unroll
    for (int it = 0; it < (WIDCAP + WIN_THREADS - 1) / WIN_THREADS; ++it) {
      const int p = (int)threadIdx.x + it * WIN_THREADS;
      if (p < WIDCAP) idbuf[(c_begin & 1) * WIDCAP + p] = ids[it];
    }
    __syncthreads();
  }

  for (int i = 20; i < 31; ++i) KT_ZERO(i);
  for (int ch = c_begin; ch < c_end; ++ch) {
    [[maybe_unused]] const unsigned long long kt_0 = KT_NOW();
    const int r_c = ch * WR;
    const int* rp = rpbuf + (ch % 3) * AGG_MAX_IN * WRP;
    const int* idc = idbuf + (ch & 1) * WIDCAP;
    const bool next = ch + 1 < c_end;
    // ---- requests for the chunks ahead --------------------------------------------------------------------------------------------------
    WinStage stg;
    {
      const int nlo = r_c + WR + WM;  // new ring rows of chunk ch + 1: [nlo, nlo + WR)
#pragma unroll
      for (int it = 0; it < (WR * 32) / WIN_THREADS; ++it) {
        const int p = (int)threadIdx.x + it * WIN_THREADS;
        const int r = nlo + (p >> 5), piece = p & 31;
        stg.rows[it] = make_uint4(0u, 0u, 0u, 0u);
        if (next && r < n_rows) stg.rows[it] = *reinterpret_cast<const uint4*>(zwin + (int64_t)r * IW.ldz + piece * 8);
      }
      win_load_ids(D, rpbuf + ((ch + 1) % 3) * AGG_MAX_IN * WRP, next, stg.ids);
      stg.rp = (ch + 2 < c_end) ? win_load_rp(D, ch + 2, a.n_chunks) : 0;
    }
    int off[AGG_MAX_IN], eb[AGG_MAX_IN];
    const bool fits = win_offsets(rp, nq, off, eb);
    const int wlo = max(r_c - WM, 0), whi = min(r_c + WR + WM, n_rows);  // rows the ring holds now
    const int offW = sel_q(off, WQ) - sel_q(eb, WQ);
    // ---- counts + far table + k-step masks: the 16 threads (i, k) of row i take its edges k, k + 16, .. -------------------------------
    {
      const int i = (int)threadIdx.x >> 4, k0 = threadIdx.x & 15;
      const int b = rp[WQ * WRP + i], e = rp[WQ * WRP + i + 1];
      int nfar = 0;
      unsigned long long km = 0ull;
      for (int base = b;; base += 16) {
        if (!__any(base < e)) break;  // wave-uniform (the four rows of a wave may differ in length)
        const int p = base + k0;
        const bool act = p < e;
        const int id = act ? (fits ? idc[offW + p] : IW.col[p]) : 0;
        const bool inw = act && id >= wlo && id < whi;
        const bool far = act && !inw;
        if (inw) {
          const int sl = id % W4S;
          asm volatile("ds_pk_add_bf16 %0, %1" ::"v"(w4_lds_addr(cmat + i * W4_CP + (sl >> 1) * 4)), "v"((sl & 1) ? 0x3f800000u : 0x00003f80u) : "memory");
          km |= 1ull << (sl >> 2);
        }
        const unsigned long long fb = __ballot(far);
        const unsigned grp = (unsigned)(fb >> (lane & 48)) & 0xffffu;  // the far flags of this row's 16 threads
        const int pos = nfar + __popc(grp & ((1u << k0) - 1u));
        if (far && pos < W4_FARCAP) farid[i * W4_FARCAP + pos] = id;
        nfar += __popc(grp);
      }
      if (k0 == 0) farcnt[i] = nfar;
      if (km) atomicOr(&kmask[i >> 2], km);  // rows 4 w .. 4 w + 3 belong to wave w (an OR of flags: order-independent)
    }
    w4_barrier();  // counts, far table and masks complete
    KT_ADD(20, kt_0);
    [[maybe_unused]] const unsigned long long kt_4 = KT_NOW();
    // ---- the product for the wave's rows 4 w .. 4 w + 3 ---------------------------------------------------------------------------------------
    w4_f32x4 accm[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) accm[c] = w4_f32x4{0.f, 0.f, 0.f, 0.f};
    {
      unsigned long long km = kmask[wave];
      km = ((unsigned long long)(uint32_t)uni((int)(km >> 32)) << 32) | (uint32_t)uni((int)km);
      const unsigned char* Ai = cmat + (4 * wave + (lane & 3)) * W4_CP;
      const int q = (lane & 15) >> 2;
      uint32_t boff[4];  // byte offset of this lane's transposed-read address inside a ring row group, per call c (without the row term)
      int bch[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int fo = 64 * (c & 1) + 16 * (lane >> 4) + 4 * (lane & 3);  // feature offset inside the sub-image
        bch[c] = fo >> 3;
        boff[c] = (c >> 1) * W4_SUB + 8 * ((fo >> 2) & 1);
      }
      while (km) {  // wave-uniform
        const int k4 = __builtin_ctzll(km);
        km &= km - 1ull;
        const int row = 4 * k4 + q;
        const uint2 araw = *reinterpret_cast<const uint2*>(Ai + 8 * k4);
        w4_s16x4 bv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) bv[c] = w4_read_tr(w4_lds_addr(ring + boff[c] + w4_off(row, bch[c])));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        union { uint2 u; w4_s16x4 v; } ua;
        ua.u = araw;
#pragma unroll
        for (int c = 0; c < 4; ++c) accm[c] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ua.v, bv[c], accm[c], 0, 0, 0);
      }
    }
    w4_barrier();  // every wave has left the ring and C
    KT_ADD(24, kt_4);
    [[maybe_unused]] const unsigned long long kt_7 = KT_NOW();
    // ---- the wave's 4 rows x 256 features, one 64-feature slice (MFMA call c) at a time: per slice ALL the rows' requests -- root, the
    // first three far edges, the first edge of the other edge type -- are issued together (20 loads of 128 contiguous bytes), then
    // summed, finished and stored.  (Row by row -- the per-edge kernel's order -- every row waited for its own far row and then for its
    // rooms -> objects row with nothing else to do: 13 us of a 22 us chunk.) -----------------------------------------------------------------
    const int DQ = (nq > 1) ? (WQ == 0 ? 1 : 0) : -1;  // first other edge type; further types (and further edges of this one) are walked
    const AggIn& ID = D.in[DQ >= 0 ? DQ : WQ];
    const uint16_t* zoth = reinterpret_cast<const uint16_t*>(ID.z) + ID.coff;
    const __amdgpu_buffer_rsrc_t rsW = win_rsrc(zwin, (unsigned)IW.n_src * (unsigned)(IW.ldz * 2) - (unsigned)(IW.coff * 2));
    const __amdgpu_buffer_rsrc_t rsD = win_rsrc(zoth, (unsigned)ID.n_src * (unsigned)(ID.ldz * 2) - (unsigned)(ID.coff * 2));
    const int offD = DQ >= 0 ? sel_q(off, DQ) - sel_q(eb, DQ) : 0;
    int rowg[4], nfg[4], dWg[4], bDg[4], dDg[4], f0g[4], f1g[4], f2g[4], d0g[4];  // wave-uniform per row
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int il = 4 * wave + g;
      rowg[g] = r_c + il;
      const bool live = rowg[g] < n_rows;
      dWg[g] = live ? uni(rp[WQ * WRP + il + 1]) - uni(rp[WQ * WRP + il]) : 0;
      nfg[g] = live ? uni(farcnt[il]) : 0;
      f0g[g] = uni(farid[il * W4_FARCAP + 0]);
      f1g[g] = uni(farid[il * W4_FARCAP + 1]);
      f2g[g] = uni(farid[il * W4_FARCAP + 2]);
      bDg[g] = DQ >= 0 ? uni(rp[DQ * WRP + il]) : 0;
      dDg[g] = (live && DQ >= 0) ? uni(rp[DQ * WRP + il + 1]) - bDg[g] : 0;
      d0g[g] = dDg[g] > 0 ? uni(fits ? idc[offD + bDg[g]] : ID.col[bDg[g]]) : 0;
    }
    const int ldW2 = IW.ldz * 2, ldD2 = ID.ldz * 2;
    auto bf = [](uint32_t v) { return __uint_as_float(v << 16); };
    auto slice = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      const unsigned vo = (unsigned)(lane * 2 + 128 * c);
      uint32_t rootv[4], farv[4][3], othv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int rowc = min(rowg[g], n_rows - 1);
        rootv[g] = reinterpret_cast<const uint16_t*>(D.zroot)[(int64_t)rowc * D.ldzr + D.roff + 64 * c + lane];
        farv[g][0] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsW, nfg[g] > 0 ? vo : WIN_SKIP_OFF, f0g[g] * ldW2, 0);
        farv[g][1] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsW, nfg[g] > 1 ? vo : WIN_SKIP_OFF, f1g[g] * ldW2, 0);
        farv[g][2] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsW, nfg[g] > 2 ? vo : WIN_SKIP_OFF, f2g[g] * ldW2, 0);
        othv[g] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsD, dDg[g] > 0 ? vo : WIN_SKIP_OFF, d0g[g] * ldD2, 0);
      }
      // dropout decisions of quad 16 c + (lane >> 2) of the four rows: quad lane j computes row j's pair, the four lanes exchange them
      uint32_t ha = 0u, hb = 0u;
      if (D.drop_on) {
        const int rsel = (lane & 3) == 0 ? rowg[0] : ((lane & 3) == 1 ? rowg[1] : ((lane & 3) == 2 ? rowg[2] : rowg[3]));
        drop_pair(dcfg, (uint32_t)min(rsel, n_rows - 1) * (uint32_t)(D.ldo >> 2) + (uint32_t)(16 * c + (lane >> 2)), ha, hb);
      }
      uint32_t pa[4], pb[4];
      pa[0] = __builtin_amdgcn_mov_dpp((int)ha, 0x00, 0xf, 0xf, true); pb[0] = __builtin_amdgcn_mov_dpp((int)hb, 0x00, 0xf, 0xf, true);
      pa[1] = __builtin_amdgcn_mov_dpp((int)ha, 0x55, 0xf, 0xf, true); pb[1] = __builtin_amdgcn_mov_dpp((int)hb, 0x55, 0xf, 0xf, true);
      pa[2] = __builtin_amdgcn_mov_dpp((int)ha, 0xaa, 0xf, 0xf, true); pb[2] = __builtin_amdgcn_mov_dpp((int)hb, 0xaa, 0xf, 0xf, true);
      pa[3] = __builtin_amdgcn_mov_dpp((int)ha, 0xff, 0xf, 0xf, true); pb[3] = __builtin_amdgcn_mov_dpp((int)hb, 0xff, 0xf, 0xf, true);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (rowg[g] >= n_rows) continue;  // wave-uniform
        const int il = 4 * wave + g;
        float sw = accm[c][g] + bf(farv[g][0]);
        sw += bf(farv[g][1]);
        sw += bf(farv[g][2]);
        if (nfg[g] > 3) {  // further far edges: table entries 3 .. 7, then (a graph without locality) the rest of the row's list
          for (int jf = 3; jf < min(nfg[g], W4_FARCAP); ++jf)
            sw += bf((uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsW, vo, uni(farid[il * W4_FARCAP + jf]) * ldW2, 0));
          if (nfg[g] > W4_FARCAP) {
            const int b = uni(rp[WQ * WRP + il]);
            int seen = 0;
            for (int p = 0; p < dWg[g]; ++p) {
              const int id = uni(fits ? idc[offW + b + p] : IW.col[b + p]);
              if (id >= wlo && id < whi) continue;
              if (seen++ < W4_FARCAP) continue;
              sw += bf((uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsW, vo, id * ldW2, 0));
            }
          }
        }
        float v = biasv[c] + bf(rootv[g]);
        if (dWg[g] > 0) v = fmaf(sw, a.mean ? 1.0f / (float)dWg[g] : 1.f, v);
        if (dDg[g] > 0) {
          float sd = bf(othv[g]);
          for (int p = 1; p < dDg[g]; ++p)  // (further edges of the other type: one waited-for row at a time)
            sd += bf((uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rsD, vo, uni(fits ? idc[offD + bDg[g] + p] : ID.col[bDg[g] + p]) * ldD2, 0));
          v = fmaf(sd, a.mean ? 1.0f / (float)dDg[g] : 1.f, v);
        }
        for (int q = 0; q < nq; ++q) {  // a third, fourth .. edge type: walked
          if (q == WQ || q == DQ) continue;
          const AggIn& I = D.in[q];
          const int b = uni(rp[q * WRP + il]), e = uni(rp[q * WRP + il + 1]);
          if (e == b) continue;
          const uint16_t* zq = reinterpret_cast<const uint16_t*>(I.z) + I.coff;
          const int offq = sel_q(off, q) - sel_q(eb, q);
          float sq = 0.f;
          for (int p = b; p < e; ++p) sq += bf((uint32_t)zq[(int64_t)uni(fits ? idc[offq + p] : I.col[p]) * I.ldz + 64 * c + lane]);
          v = fmaf(sq, a.mean ? 1.0f / (float)(e - b) : 1.f, v);
        }
        if (D.act == HMP_ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (D.act == HMP_ACT_ELU) v = v > 0.f ? v : expm1f(v);
        if (D.drop_on) {
          const uint32_t word = (lane & 2) ? pb[g] : pa[g];
          const uint32_t draw = (lane & 1) ? (word >> 16) : (word & 0xffffu);
          v = draw >= t16 ? (v * D.drop.scale + 0.0f) : -0.0f;
        }
        if constexpr (HB) reinterpret_cast<__bf16*>(D.out)[(int64_t)rowg[g] * D.ldo + 64 * c + lane] = (__bf16)v;
        else D.out[(int64_t)rowg[g] * D.ldo + 64 * c + lane] = v;
      }
    };
    slice(std::integral_constant<int, 0>());
    slice(std::integral_constant<int, 1>());
    slice(std::integral_constant<int, 2>());
    slice(std::integral_constant<int, 3>());
    KT_ADD(27, kt_7);
    [[maybe_unused]] const unsigned long long kt_5 = KT_NOW();
    // ---- what the requests brought goes to LDS; the counts are cleared for the next chunk -----------------------------------------------
    if (next) {
      const int nlo = r_c + WR + WM;
#pragma unroll
      for (int it = 0; it < (WR * 32) / WIN_THREADS; ++it) {
        const int p = (int)threadIdx.x + it * WIN_THREADS;
        const int r = nlo + (p >> 5), piece = p & 31;
        *reinterpret_cast<uint4*>(w4_ring_at(ring, w4_slot(r), piece)) = stg.rows[it];  // (rows past the matrix: zeros)
      }
#pragma unroll
      for (int it = 0; it < (WIDCAP + WIN_THREADS - 1) / WIN_THREADS; ++it) {
        const int p = (int)threadIdx.x + it * WIN_THREADS;
        if (p < WIDCAP) idbuf[((ch + 1) & 1) * WIDCAP + p] = stg.ids[it];
      }
      if (ch + 2 < c_end && (int)threadIdx.x < nq * WRP) rpbuf[((ch + 2) % 3) * AGG_MAX_IN * WRP + threadIdx.x] = stg.rp;
      zero_c();
    }
    KT_ADD(25, kt_5);
    [[maybe_unused]] const unsigned long long kt_8 = KT_NOW();
    w4_barrier();
    KT_ADD(28, kt_8);
    KT_ADD(29, kt_0);
  }
}
#endif  // HMP_EXPERIMENTS

// transposed counterpart: source rows in chunks, ring over the gradient rows G of the destination type (same index space) and
// their reciprocal in-degrees; one accumulator and one output segment per outgoing edge type
template <bool DZB>
__global__ __launch_bounds__(WIN_THREADS) void agg_bwd_win_kernel(const WinBwd a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wlds[];
  unsigned char* ring = wlds;
  int* idbuf = reinterpret_cast<int*>(wlds + WIN_LDS_RING);
  int* rpbuf = reinterpret_cast<int*>(wlds + WIN_LDS_RING + WIN_LDS_IDS);
  float* wdeg = reinterpret_cast<float*>(wlds + WIN_LDS_FWD);
  const TAggSrc& S = a.s;
  const int n_rows = S.n_rows, nq = S.n_out;
  const int c_begin = (int)blockIdx.x * a.chunks_per_block;
  const int c_end = min(c_begin + a.chunks_per_block, a.n_chunks);
  if (c_begin >= c_end) return;
  const int wave = uni((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
  const TAggOut& OW = S.out[S.win_out];
  const uint16_t* gwin = reinterpret_cast<const uint16_t*>(OW.g);
  const bool wdg = a.mean && OW.degf != nullptr;
  const int c0 = lane * 4;
  {
    const int r0 = win_load_rp(S, c_begin, a.n_chunks), r1 = win_load_rp(S, c_begin + 1, a.n_chunks);
    if ((int)threadIdx.x < nq * WRP) {
      rpbuf[(c_begin % 3) * AGG_MAX_IN * WRP + threadIdx.x] = r0;
      rpbuf[((c_begin + 1) % 3) * AGG_MAX_IN * WRP + threadIdx.x] = r1;
    }
    if ((int)threadIdx.x < 32) *reinterpret_cast<uint4*>(ring + (size_t)WRING * WIN_ROW_BYTES + threadIdx.x * 16) = make_uint4(0u, 0u, 0u, 0u);
    const int lo = max(c_begin * WR - WM, 0), hi = min(c_begin * WR + WR + WM, n_rows);
    for (int p = threadIdx.x; p < (hi - lo) * 32; p += WIN_THREADS) {
      const int r = lo + (p >> 5), piece = p & 31;
      const uint4 v = *reinterpret_cast<const uint4*>(gwin + (int64_t)r * OW.ldg + piece * 8);
      *reinterpret_cast<uint4*>(ring + (size_t)(r & (WRING - 1)) * WIN_ROW_BYTES + piece * 16) = v;
    }
    if (wdg && (int)threadIdx.x < hi - lo) wdeg[(lo + threadIdx.x) & (WRING - 1)] = OW.degf[lo + threadIdx.x];
    __syncthreads();
    int ids[(WIDCAP + WIN_THREADS - 1) / WIN_THREADS];
    win_load_ids(S, rpbuf + (c_begin % 3) * AGG_MAX_IN * WRP, true, ids);
#pragma unroll
    for (int it = 0; it < (WIDCAP + WIN_THREADS - 1) / WIN_THREADS; ++it) {
      const int p = (int)threadIdx.x + it * WIN_THREADS;
      if (p < WIDCAP) idbuf[(c_begin & 1) * WIDCAP + p] = ids[it];
    }
    __syncthreads();
  }
  for (int ch = c_begin; ch < c_end; ++ch) {
    const int r_c = ch * WR;
    const int* rp = rpbuf + (ch % 3) * AGG_MAX_IN * WRP;
    const int* idc = idbuf + (ch & 1) * WIDCAP;
    WinStage stg;
    const bool next = ch + 1 < c_end;
    {
      const int nlo = r_c + WR + WM;
#pragma unroll
      for (int it = 0; it < (WR * 32) / WIN_THREADS; ++it) {
        const int p = (int)threadIdx.x + it * WIN_THREADS;
        const int r = nlo + (p >> 5), piece = p & 31;
        stg.rows[it] = make_uint4(0u, 0u, 0u, 0u);
        if (next && r < n_rows) stg.rows[it] = *reinterpret_cast<const uint4*>(gwin + (int64_t)r * OW.ldg + piece * 8);
      }
      stg.deg = 0.f;
      if (next && wdg && (int)threadIdx.x < WR && nlo + (int)threadIdx.x < n_rows) stg.deg = OW.degf[nlo + threadIdx.x];
      win_load_ids(S, rpbuf + ((ch + 1) % 3) * AGG_MAX_IN * WRP, next, stg.ids);
      stg.rp = (ch + 2 < c_end) ? win_load_rp(S, ch + 2, a.n_chunks) : 0;
    }
    int off[AGG_MAX_IN], eb[AGG_MAX_IN];
    const bool fits = win_offsets(rp, nq, off, eb);
    const int wlo = max(r_c - WM, 0), whi = min(r_c + WR + WM, n_rows);
    // phase C (see agg_fwd_win_kernel): one accumulator and one output segment per outgoing edge type
#pragma unroll 1
    for (int g = 0; g < WG; ++g) {
      const int rl = g * WIN_WAVES + wave, row = r_c + rl;
      if (row >= n_rows) continue;
#pragma unroll 1
      for (int q = 0; q < nq; ++q) {
        const TAggOut& O = S.out[q];
        const int b = fits ? uni(rp[q * WRP + rl]) : O.t_rowptr[row], e = fits ? uni(rp[q * WRP + rl + 1]) : O.t_rowptr[row + 1];
        const bool cin = c0 < O.F;
        const int cc = cin ? c0 : 0;
        const bool dg = a.mean && O.degf != nullptr;
        const uint16_t* gq = reinterpret_cast<const uint16_t*>(O.g);
        const int ldq = O.ldg;
        const __amdgpu_buffer_rsrc_t rs = win_rsrc(gq, (unsigned)O.n_dst * (unsigned)(ldq * 2));
        const __amdgpu_buffer_rsrc_t rs_null = win_rsrc(gq, 0u);
        const int offq = fits ? sel_q(off, q) + (b - sel_q(eb, q)) : 0;
        const bool wq = fits && q == S.win_out;
        Acc<4> acc;
        acc.zero();
        for (int base = 0; base < e - b; base += 64) {
          const int cnt = min(e - b - base, 64);
          const int idv = lane < cnt ? (fits ? idc[offq + base + lane] : O.t_col[b + base + lane]) : 0;
          const bool inw = lane < cnt && wq && idv >= wlo && idv < whi;
          const bool farl = lane < cnt && !inw;
          // the edge's weight 1 / deg(dst), lane u = edge u: in-window edges from the ring's table (LDS), the others from global memory --
          // in TWO registers, so that the LDS batches never wait for the global ones (see agg_fwd_win_kernel: out-of-window rows are
          // requested first and added last)
          float dvl = 1.f, dvg = 1.f;
          if (a.mean) {
            if (dg) {
              if (inw) dvl = wdeg[idv & (WRING - 1)];
              if (farl) dvg = O.degf[idv];
            } else if (lane < cnt) {
              const int g0 = O.rowptr[idv + 1] - O.rowptr[idv];
              dvl = dvg = 1.f / (float)(g0 > 1 ? g0 : 1);
            }
          }
          const int dvli = __float_as_int(dvl), dvgi = __float_as_int(dvg);
          const int lov = inw ? (idv & (WRING - 1)) * WIN_ROW_BYTES : WRING * WIN_ROW_BYTES;  // see agg_fwd_win_kernel
          const int gov = idv * (ldq * 2);
          unsigned long long fm = __ballot(farl);
          u32x2 fv[4];
          int fu[4];
          auto far_issue = [&]() {  // the next (up to) four set bits of fm
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              fu[t] = fm ? __builtin_ctzll(fm) : 0;
              fv[t] = __builtin_amdgcn_raw_buffer_load_b64(fm ? rs : rs_null, cc * 2, __builtin_amdgcn_readlane(gov, fu[t]), 0);
              fm &= fm - 1ull;  // (0 stays 0)
            }
          };
          auto far_add = [&]() {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              Acc<4> w;
              widen_bf16x4(w, make_uint2(fv[t][0], fv[t][1]));
              if (a.mean) acc.add_mul(w, __int_as_float(__builtin_amdgcn_readlane(dvgi, fu[t]))); else acc.add(w);  // (a null read: 0 x a finite weight)
            }
          };
          const bool any_far = fm != 0ull;
          if (any_far) far_issue();
          if (wq) {
            auto batch = [&](int u0, auto nb) {
              constexpr int NB = decltype(nb)::value;
              uint2 va[NB];
#pragma unroll
              for (int t = 0; t < NB; ++t) va[t] = *reinterpret_cast<const uint2*>(ring + __builtin_amdgcn_readlane(lov, u0 + t) + lane * 8);
#pragma unroll
              for (int t = 0; t < NB; ++t) {
                Acc<4> w;
                widen_bf16x4(w, va[t]);
                if (a.mean) acc.add_mul(w, __int_as_float(__builtin_amdgcn_readlane(dvli, u0 + t))); else acc.add(w);
              }
            };
            int u0 = 0;
            for (; cnt - u0 > 2; u0 += 8) batch(u0, std::integral_constant<int, 8>());
            if (u0 < cnt) batch(u0, std::integral_constant<int, 2>());
          }
          if (any_far) {
            far_add();
            while (fm) {  // more than four: further rounds, each waited for
              far_issue();
              far_add();
            }
          }
        }
        if (cin) store_z<DZB>(acc, S.dz, (int64_t)row * S.lddz + O.coff + c0);
      }
      if (S.groot && c0 < S.Froot) {
        Acc<4> v;
        load_z<true>(v, S.groot, (int64_t)row * S.ldgr + c0);
        store_z<DZB>(v, S.dz, (int64_t)row * S.lddz + S.roff + c0);
      }
    }
    if (next) {
      const int nlo = r_c + WR + WM;
#pragma unroll
      for (int it = 0; it < (WR * 32) / WIN_THREADS; ++it) {
        const int p = (int)threadIdx.x + it * WIN_THREADS;
        const int r = nlo + (p >> 5), piece = p & 31;
        if (r < n_rows) *reinterpret_cast<uint4*>(ring + (size_t)(r & (WRING - 1)) * WIN_ROW_BYTES + piece * 16) = stg.rows[it];
      }
      if (wdg && (int)threadIdx.x < WR && nlo + (int)threadIdx.x < n_rows) wdeg[(nlo + threadIdx.x) & (WRING - 1)] = stg.deg;
#pragma unroll
      for (int it = 0; it < (WIDCAP + WIN_THREADS - 1) / WIN_THREADS; ++it) {
        const int p = (int)threadIdx.x + it * WIN_THREADS;
        if (p < WIDCAP) idbuf[((ch + 1) & 1) * WIDCAP + p] = stg.ids[it];
      }
      if (ch + 2 < c_end && (int)threadIdx.x < nq * WRP) rpbuf[((ch + 2) % 3) * AGG_MAX_IN * WRP + threadIdx.x] = stg.rp;
    }
    __syncthreads();
  }
}

#ifdef HMP_EXPERIMENTS
int chain_launch(const ChainArgs* d_args, int n_graphs, int fin_rows, int gs, size_t lds_bytes, hipStream_t st) {
  lds_bytes += ((sizeof(ChainArgs) + 15) / 16) * 16;  // + the LDS copy of the argument block
  HMP_CHECK_ARG(d_args && n_graphs > 0 && (gs == 16 || gs == 32) && lds_bytes <= 150 * 1024, "chain: bad launch (%d graphs, gs %d, %zu LDS bytes)", n_graphs, gs, lds_bytes);
  static bool attr_done = false;
  if (!attr_done) {
    HMP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    HMP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&chain_kernel<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    attr_done = true;
  }
  if (gs == 16) hipLaunchKernelGGL((chain_kernel<16>), dim3(n_graphs), dim3(WIN_THREADS_CHAIN), lds_bytes, st, d_args, n_graphs, fin_rows);
  else hipLaunchKernelGGL((chain_kernel<32>), dim3(n_graphs), dim3(WIN_THREADS_CHAIN), lds_bytes, st, d_args, n_graphs, fin_rows);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}
#else
int chain_launch(const ChainArgs*, int, int, int, size_t, hipStream_t) { HMP_FAIL(HMP_E_UNSUPPORTED, "chain_launch: experiment build only (make EXPERIMENTS=1)"); }
#endif

// ----- dispatch ---------------------------------------------------------------------------------------
// lanes needed = ceil(F / VEC); GS = next pow2 in [8, 64]; NV = ceil(lanes / GS) <= 4
// Measured at config 5 (10^6 rows of 256 floats, rocprofv3 --pmc): 152 M L2 requests per forward launch (= the 128-byte lines
// of the algorithm: 17 neighbour rows + root + output per row), 70 % hits, 0.1 % memory-unit stalls, VALU 40 % busy, 2.33 ms.
// The microarchitecture guide's gather rates (~18 TB/s from L2, ~6 TB/s beyond it) put the same traffic at ~1.7 ms, so the
// one-row-per-wave kernels run at ~75 % of the practical ceiling.  Tried on top and measured neutral (+-3 %), hence not kept:
// bf16 rows; 6 instead of 4 resident waves per SIMD (80-register shape without pair batching); a streaming form in which a
// wave walks 8 consecutive rows with the index chain of the next two rows in flight (bit-identical, forward 4.62 -> 4.96 ms,
// backward 7.54 -> 7.38 ms).  Kept: the XCD-contiguous row ranges below (neutral here, but it is the mapping that does not
// depend on the Infinity Cache absorbing 8 copies of every shared row).

static inline void pick_shape(int F, int vec, int& gs, int& nv) {
  const int lanes = cdiv(F, vec);
  gs = 8;
  while (gs < 64 && gs < lanes) gs <<= 1;
  nv = cdiv(lanes, gs);
}

#define HMP_DISPATCH_GS_NV(GSV, NVV, MACRO)                                       \
  switch ((GSV) * 8 + (NVV)) {                                                    \
    case 8 * 8 + 1: MACRO(8, 1); break;                                           \
    case 16 * 8 + 1: MACRO(16, 1); break;                                         \
    case 32 * 8 + 1: MACRO(32, 1); break;                                         \
    case 64 * 8 + 1: MACRO(64, 1); break;                                         \
    case 64 * 8 + 2: MACRO(64, 2); break;                                         \
    case 64 * 8 + 3: MACRO(64, 3); break;                                         \
    case 64 * 8 + 4: MACRO(64, 4); break;                                         \
    default: HMP_FAIL(HMP_E_ARG, "row width %d not supported by the aggregation kernels (max %d)", Fmax, 64 * 4 * vec); \
  }

constexpr int64_t AGG_XCD_ROWS = 65536;  // from here on the launch is several waves of blocks deep and L2 locality pays
static bool agg_xcd_enabled() {  // HMP_AGG_XCD=0: plain block order (tests compare both orders bit for bit)
  const char* v = getenv("HMP_AGG_XCD");
  return !(v && v[0] == '0');
}

// HMP_AGG_WIN=0 turns the LDS sliding-window kernels off (tests compare both forms bit for bit)
static bool agg_win_enabled() {
  const char* v = getenv("HMP_AGG_WIN");
  return !(v && v[0] == '0');
}
#ifdef HMP_EXPERIMENTS
// experiment builds, HMP_AGG_W4=1: the forward in-window sum by 4x4x4 MFMAs over a count matrix (agg_fwd_w4_kernel) instead of edge by edge
static bool agg_w4_enabled() {
  const char* v = getenv("HMP_AGG_W4");
  return v && v[0] == '1';
}
#endif
constexpr int AGG_WIN_MIN_ROWS = 16384;  // below: a persistent grid would leave CUs idle, the plain kernels do as well
static int agg_win_grid(int n_chunks, int& per_block) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cus = v;
  }
  per_block = cdiv(n_chunks, cus);
  return cdiv(n_chunks, per_block);
}

static int agg_win_in(const AggDst& D) {  // in-conv to serve from the LDS ring, -1: none
  if (D.F != 256 || D.n_rows < AGG_WIN_MIN_ROWS || D.ce_labels || (D.ldo & 3) || !D.zroot) return -1;
  for (int ii = 0; ii < D.n_in; ++ii)  // every source matrix addressable by a 32-bit buffer offset below WIN_SKIP_OFF
    if ((uint64_t)D.in[ii].n_src * (uint64_t)D.in[ii].ldz * 2 > (uint64_t)WIN_SKIP_OFF) return -1;
  for (int ii = 0; ii < D.n_in; ++ii) {
    const AggIn& I = D.in[ii];
    if (I.same_type && !(I.coff & 7) && !(I.ldz & 7) && !(reinterpret_cast<uintptr_t>(I.z) & 15)) return ii;
  }
  return -1;
}
static int agg_win_out(const TAggSrc& S) {
  if (S.n_rows < AGG_WIN_MIN_ROWS) return -1;
  for (int oi = 0; oi < S.n_out; ++oi)
    if ((uint64_t)S.out[oi].n_dst * (uint64_t)S.out[oi].ldg * 2 > (uint64_t)WIN_SKIP_OFF) return -1;
  for (int oi = 0; oi < S.n_out; ++oi) {
    const TAggOut& O = S.out[oi];
    if (O.same_type && O.F == 256 && !(O.ldg & 7) && !(reinterpret_cast<uintptr_t>(O.g) & 15)) return oi;
  }
  return -1;
}

constexpr int AGG_SMALL_TILES = 224;  // 16-row tiles of a launch up to which heavy entries are cut into tiles of 8 (AggDst::tile_rows)
// the compact copy of the entries' first blocks that the kernels search (AggArgs::bstart)
static inline void sync_bstart(AggArgs& a) {
  for (int i = 0; i < a.n; ++i) a.bstart[i] = a.d[i].block_start;
}
static inline void sync_bstart(TAggArgs& a) {
  for (int i = 0; i < a.n; ++i) a.bstart[i] = a.s[i].block_start;
}
int agg_fwd_launch(AggArgs& a, hipStream_t st) {
  int Fmax = 0, blocks = 0;
  const int vec = 4;
  for (int i = 0; i < a.n; ++i) Fmax = a.d[i].F > Fmax ? a.d[i].F : Fmax;
  if (a.n == 0 || Fmax == 0) return HMP_OK;
  int gs, nv;
  pick_shape(Fmax, vec, gs, nv);
  int64_t rows_total = 0;
  for (int i = 0; i < a.n; ++i) rows_total += a.d[i].n_rows;
  a.xcd = (rows_total >= AGG_XCD_ROWS && agg_xcd_enabled()) ? 1 : 0;
  for (int i = 0; i < a.n; ++i) {
    AggDst& D = a.d[i];
    HMP_CHECK_ARG((D.ldo & 3) == 0 && (D.F & 3) == 0, "agg_fwd: widths must be padded to 4");
    D.block_start = blocks;
    if (a.xcd || a.zb16 || rows_total > 16 * AGG_SMALL_TILES) D.tile_rows = 0;  // (tiles of 8 rows are a small-launch device)
    const int nb = cdiv(D.n_rows, (D.tile_rows == 8 && 256 / gs > 8) ? 8 : 256 / gs);
    blocks += a.xcd ? ((nb + 7) & ~7) : nb;  // xcd: every entry starts at a multiple of 8 and owns whole groups of 8 blocks
  }
  a.total_blocks = blocks;
  if (blocks == 0) return HMP_OK;
  if (a.zb16) {  // bf16 projected rows: only the one-wavefront-per-row shape reads them
    HMP_CHECK_ARG(gs == 64 && nv == 1, "agg_fwd: bf16 projected rows need row widths in (128, 256], got %d", Fmax);
    if (agg_win_enabled()) {
      // entries with a same-type edge type over >= 16384 rows go to the sliding-window kernel (one launch each, persistent
      // grid of one workgroup per CU); the others stay in the plain launch below
      AggArgs rest = a;
      rest.n = 0;
      bool any = false;
      for (int i = 0; i < a.n; ++i) {
        const int wi = agg_win_in(a.d[i]);
        if (wi < 0) { rest.d[rest.n++] = a.d[i]; continue; }
        static bool attr_done = false;
        if (!attr_done) {
          HMP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&agg_fwd_win_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, WIN_LDS_FWD));
          HMP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&agg_fwd_win_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, WIN_LDS_FWD));
          attr_done = true;
        }
        WinFwd w;
        memset(&w, 0, sizeof(w));
        w.d = a.d[i];
        w.d.win_in = wi;
        w.mean = a.mean;
        w.n_chunks = cdiv(w.d.n_rows, WR);
        w.dbg = 0;
#ifdef HMP_KTIME  // measurement-only switch (wrong results): profiling build only
        {
          const char* dv = getenv("HMP_WIN_DBG");
          w.dbg = dv ? atoi(dv) : 0;
        }
#endif
        const int grid = agg_win_grid(w.n_chunks, w.chunks_per_block);
#ifdef HMP_EXPERIMENTS
        if (agg_w4_enabled()) {  // the in-window sum on the matrix pipe, row-per-wave layout
          static bool w4_attr_done = false;
          if (!w4_attr_done) {
            HMP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&agg_fwd_w4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS));
            HMP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&agg_fwd_w4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS));
            w4_attr_done = true;
          }
          if (a.hb16) hipLaunchKernelGGL((agg_fwd_w4_kernel<true>), dim3(grid), dim3(WIN_THREADS), W4_LDS, st, w);
          else hipLaunchKernelGGL((agg_fwd_w4_kernel<false>), dim3(grid), dim3(WIN_THREADS), W4_LDS, st, w);
        } else
#endif
        if (a.hb16) hipLaunchKernelGGL((agg_fwd_win_kernel<true>), dim3(grid), dim3(WIN_THREADS), WIN_LDS_FWD, st, w);
        else hipLaunchKernelGGL((agg_fwd_win_kernel<false>), dim3(grid), dim3(WIN_THREADS), WIN_LDS_FWD, st, w);
        HMP_LAUNCH_CHECK();
        any = true;
      }
      if (any) {
        if (rest.n == 0) return HMP_OK;
        a = rest;
        blocks = 0;
        for (int i = 0; i < a.n; ++i) {
          a.d[i].block_start = blocks;
          const int nb = cdiv(a.d[i].n_rows, 256 / gs);
          blocks += a.xcd ? ((nb + 7) & ~7) : nb;
        }
        a.total_blocks = blocks;
      }
    }
    sync_bstart(a);
    if (a.hb16) hipLaunchKernelGGL((agg_fwd_kernel<64, 1, true, true>), dim3(blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((agg_fwd_kernel<64, 1, true>), dim3(blocks), dim3(256), 0, st, a);
    HMP_LAUNCH_CHECK();
    return HMP_OK;
  }
  HMP_CHECK_ARG(!a.hb16, "agg_fwd: bf16 outputs need bf16 projected rows (the one-wavefront-per-row kernel)");
  sync_bstart(a);
#define LAUNCH_FWD(GS_, NV_) hipLaunchKernelGGL((agg_fwd_kernel<GS_, NV_>), dim3(blocks), dim3(256), 0, st, a)
  HMP_DISPATCH_GS_NV(gs, nv, LAUNCH_FWD)
#undef LAUNCH_FWD
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

int agg_proj_fwd_launch(AggArgs& a, hipStream_t st) {
  int Fmax = 0, blocks = 0;
  for (int i = 0; i < a.n; ++i) Fmax = a.d[i].F > Fmax ? a.d[i].F : Fmax;
  // tiles of 8 rows for heavy entries only while the launch leaves CUs idle (more workgroups in a launch of several rounds only add rounds)
  int tiles16 = 0;
  for (int i = 0; i < a.n; ++i) tiles16 += cdiv(a.d[i].n_rows, 16);
  const bool small_launch = tiles16 <= AGG_SMALL_TILES;
  if (a.n == 0 || Fmax == 0) return HMP_OK;
  HMP_CHECK_ARG(Fmax <= 256, "agg_proj_fwd: row width %d > 256", Fmax);
  int gs = 16;
  while (gs < 64 && gs * 4 < Fmax) gs <<= 1;
  for (int i = 0; i < a.n; ++i) {
    AggDst& D = a.d[i];
    HMP_CHECK_ARG((D.ldo & 3) == 0 && (D.F & 3) == 0, "agg_proj_fwd: widths must be padded to 4");
    if (D.pw)
      HMP_CHECK_ARG((D.pK & 15) == 0 && D.pK <= 256 && D.pK <= D.F && (D.pldw & 3) == 0 && D.pncols > 0,
                    "agg_proj_fwd: projection K %d / ld %d not supported", D.pK, D.pldw);
    if (D.tile_rows != 8 || !small_launch) D.tile_rows = 16;
    D.block_start = blocks;
    blocks += cdiv(D.n_rows, D.tile_rows);
  }
  a.total_blocks = blocks;
  if (blocks == 0) return HMP_OK;
  sync_bstart(a);
  switch (gs) {
    case 16: hipLaunchKernelGGL((agg_proj_fwd_kernel<16>), dim3(blocks), dim3(256), 0, st, a); break;
    case 32: hipLaunchKernelGGL((agg_proj_fwd_kernel<32>), dim3(blocks), dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL((agg_proj_fwd_kernel<64>), dim3(blocks), dim3(256), 0, st, a); break;
  }
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

int agg_bwd_launch(TAggArgs& a, hipStream_t st) {
  int Fmax = 0, blocks = 0;
  const int vec = 4;
  for (int i = 0; i < a.n; ++i) {
    for (int o = 0; o < a.s[i].n_out; ++o) Fmax = a.s[i].out[o].F > Fmax ? a.s[i].out[o].F : Fmax;
    if (a.s[i].groot) Fmax = a.s[i].Froot > Fmax ? a.s[i].Froot : Fmax;
  }
  if (a.n == 0 || Fmax == 0) {
    if (!a.fin_row_lv) return HMP_OK;
    Fmax = 4;  // nothing to gather, but the loss still has to be finalised
  }
  int gs, nv;
  pick_shape(Fmax, vec, gs, nv);
  int64_t rows_total = 0;
  for (int i = 0; i < a.n; ++i) rows_total += a.s[i].n_rows;
  a.xcd = (rows_total >= AGG_XCD_ROWS && agg_xcd_enabled()) ? 1 : 0;
  for (int i = 0; i < a.n; ++i) {
    a.s[i].block_start = blocks;
    if (a.xcd || a.gb16 || rows_total > 16 * AGG_SMALL_TILES) a.s[i].tile_rows = 0;  // (tiles of 8 rows are a small-launch device)
    const int nb = cdiv(a.s[i].n_rows, (a.s[i].tile_rows == 8 && 256 / gs > 8) ? 8 : 256 / gs);
    blocks += a.xcd ? ((nb + 7) & ~7) : nb;
  }
  a.total_blocks = blocks;
  if (blocks == 0 && !a.fin_row_lv) return HMP_OK;
  int grid = blocks + (a.fin_row_lv ? 1 : 0);
  if (a.gb16) {  // bf16 gradient rows: only the one-wavefront-per-row shape reads them
    HMP_CHECK_ARG(gs == 64 && nv == 1, "agg_bwd: bf16 gradient rows need row widths in (128, 256], got %d", Fmax);
    if (agg_win_enabled()) {
      TAggArgs rest = a;
      rest.n = 0;
      bool any = false;
      for (int i = 0; i < a.n; ++i) {
        const int wo = agg_win_out(a.s[i]);
        if (wo < 0) { rest.s[rest.n++] = a.s[i]; continue; }
        static bool attr_done = false;
        if (!attr_done) {
          HMP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&agg_bwd_win_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, WIN_LDS_BWD));
          HMP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&agg_bwd_win_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, WIN_LDS_BWD));
          attr_done = true;
        }
        WinBwd w;
        memset(&w, 0, sizeof(w));
        w.s = a.s[i];
        w.s.win_out = wo;
        w.mean = a.mean;
        w.n_chunks = cdiv(w.s.n_rows, WR);
        const int wgrid = agg_win_grid(w.n_chunks, w.chunks_per_block);
        if (a.dzb16) hipLaunchKernelGGL((agg_bwd_win_kernel<true>), dim3(wgrid), dim3(WIN_THREADS), WIN_LDS_BWD, st, w);
        else hipLaunchKernelGGL((agg_bwd_win_kernel<false>), dim3(wgrid), dim3(WIN_THREADS), WIN_LDS_BWD, st, w);
        HMP_LAUNCH_CHECK();
        any = true;
      }
      if (any) {
        a = rest;
        blocks = 0;
        for (int i = 0; i < a.n; ++i) {
          a.s[i].block_start = blocks;
          const int nb = cdiv(a.s[i].n_rows, 256 / gs);
          blocks += a.xcd ? ((nb + 7) & ~7) : nb;
        }
        a.total_blocks = blocks;
        if (blocks == 0 && !a.fin_row_lv) return HMP_OK;
        grid = blocks + (a.fin_row_lv ? 1 : 0);
      }
    }
    sync_bstart(a);
    if (a.dzb16) hipLaunchKernelGGL((agg_bwd_kernel<64, 1, true, true>), dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((agg_bwd_kernel<64, 1, true, false>), dim3(grid), dim3(256), 0, st, a);
    HMP_LAUNCH_CHECK();
    return HMP_OK;
  }
  HMP_CHECK_ARG(!a.dzb16, "agg_bwd: a bf16 dz needs bf16 gradient rows (the one-wavefront-per-row kernel)");
  sync_bstart(a);
#define LAUNCH_BWD(GS_, NV_) hipLaunchKernelGGL((agg_bwd_kernel<GS_, NV_>), dim3(grid), dim3(256), 0, st, a)
  HMP_DISPATCH_GS_NV(gs, nv, LAUNCH_BWD)
#undef LAUNCH_BWD
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

int agg_bwd_dx_launch(TAggArgs& a, hipStream_t st) {
  int Fmax = 0, blocks = 0, kmax = 0;
  for (int i = 0; i < a.n; ++i) {
    for (int o = 0; o < a.s[i].n_out; ++o) Fmax = a.s[i].out[o].F > Fmax ? a.s[i].out[o].F : Fmax;
    if (a.s[i].groot) Fmax = a.s[i].Froot > Fmax ? a.s[i].Froot : Fmax;
    kmax = a.s[i].ncols > kmax ? a.s[i].ncols : kmax;
    HMP_CHECK_ARG((a.s[i].ncols & 3) == 0, "agg_bwd_dx: ncols must be padded to 4");
  }
  if (a.n == 0 || Fmax == 0) {
    if (!a.fin_row_lv) return HMP_OK;
    Fmax = 4;
  }
  HMP_CHECK_ARG(Fmax <= 256 && kmax <= 896, "agg_bwd_dx: segment width %d / stacked width %d not supported", Fmax, kmax);
  int gs = 16;
  while (gs < 64 && gs * 4 < Fmax) gs <<= 1;
  int tiles16 = 0;
  for (int i = 0; i < a.n; ++i) tiles16 += cdiv(a.s[i].n_rows, 16);
  const bool small_launch = tiles16 <= AGG_SMALL_TILES;  // see agg_proj_fwd_launch
  for (int i = 0; i < a.n; ++i) {
    if (a.s[i].tile_rows != 8 || !small_launch) a.s[i].tile_rows = 16;
    a.s[i].block_start = blocks;
    blocks += cdiv(a.s[i].n_rows, a.s[i].tile_rows);
  }
  a.total_blocks = blocks;
  if (blocks == 0 && !a.fin_row_lv) return HMP_OK;
  const int grid = blocks + (a.fin_row_lv ? 1 : 0);
  const size_t smem = (size_t)kmax * 17 * sizeof(float);
  sync_bstart(a);
  switch (gs) {
    case 16: hipLaunchKernelGGL((agg_bwd_dx_kernel<16>), dim3(grid), dim3(256), smem, st, a); break;
    case 32: hipLaunchKernelGGL((agg_bwd_dx_kernel<32>), dim3(grid), dim3(256), smem, st, a); break;
    default: hipLaunchKernelGGL((agg_bwd_dx_kernel<64>), dim3(grid), dim3(256), smem, st, a); break;
  }
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

template <bool FWD>
static int segment_mean_dispatch(const float* in, int ldi, int F, const hmp_plan& plan, float* out, int ldo, hipStream_t st) {
  const bool vec_ok = ((ldi & 3) == 0) && ((ldo & 3) == 0) && ((reinterpret_cast<uintptr_t>(in) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && ((F & 3) == 0);
  const int vec = vec_ok ? 4 : 1;
  const int Fmax = F;
  const int n_rows = FWD ? plan.n_dst : plan.n_src;
  if (n_rows == 0 || F == 0) return HMP_OK;
  int gs, nv;
  pick_shape(F, vec, gs, nv);
  const int blocks = cdiv(n_rows, 256 / gs);
#define LAUNCH_SM(GS_, NV_)                                                                                                   \
  do {                                                                                                                        \
    if (FWD) {                                                                                                                \
      if (vec_ok) hipLaunchKernelGGL((segment_mean_fwd_kernel<GS_, NV_, 4>), dim3(blocks), dim3(256), 0, st, in, ldi, F, plan.d_rowptr, plan.d_col, n_rows, out, ldo); \
      else hipLaunchKernelGGL((segment_mean_fwd_kernel<GS_, NV_, 1>), dim3(blocks), dim3(256), 0, st, in, ldi, F, plan.d_rowptr, plan.d_col, n_rows, out, ldo); \
    } else {                                                                                                                  \
      if (vec_ok) hipLaunchKernelGGL((segment_mean_bwd_kernel<GS_, NV_, 4>), dim3(blocks), dim3(256), 0, st, in, ldi, F, plan.d_t_rowptr, plan.d_t_col, plan.d_rowptr, n_rows, out, ldo); \
      else hipLaunchKernelGGL((segment_mean_bwd_kernel<GS_, NV_, 1>), dim3(blocks), dim3(256), 0, st, in, ldi, F, plan.d_t_rowptr, plan.d_t_col, plan.d_rowptr, n_rows, out, ldo); \
    }                                                                                                                         \
  } while (0)
  HMP_DISPATCH_GS_NV(gs, nv, LAUNCH_SM)
#undef LAUNCH_SM
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

}  // namespace hmp

extern "C" int hmp_segment_mean_fwd(const float* d_x, int32_t ldx, int32_t F, hmp_plan plan, float* d_out, int32_t ldo, void* stream) {
  using namespace hmp;
  HMP_CHECK_ARG(d_x && d_out && plan.d_rowptr, "hmp_segment_mean_fwd: null pointer");
  HMP_CHECK_ARG(F >= 0 && ldx >= F && ldo >= F, "hmp_segment_mean_fwd: bad widths");
  HMP_CHECK_ARG(plan.n_edges == 0 || plan.d_col, "hmp_segment_mean_fwd: null col");
  // widths beyond one pass are processed in column panels
  const int panel = ((ldx & 3) == 0 && (ldo & 3) == 0 && (F & 3) == 0) ? 1024 : 256;
  for (int c = 0; c < F; c += panel) {
    const int w = (F - c) < panel ? (F - c) : panel;
    HMP_TRY((segment_mean_dispatch<true>(d_x + c, ldx, w, plan, d_out + c, ldo, (hipStream_t)stream)));
  }
  return HMP_OK;
}

extern "C" int hmp_segment_mean_bwd(const float* d_gout, int32_t ldg, int32_t F, hmp_plan plan, float* d_gx, int32_t ldgx, void* stream) {
  using namespace hmp;
  HMP_CHECK_ARG(d_gout && d_gx && plan.d_rowptr && plan.d_t_rowptr, "hmp_segment_mean_bwd: null pointer");
  HMP_CHECK_ARG(F >= 0 && ldg >= F && ldgx >= F, "hmp_segment_mean_bwd: bad widths");
  HMP_CHECK_ARG(plan.n_edges == 0 || plan.d_t_col, "hmp_segment_mean_bwd: null t_col");
  const int panel = ((ldg & 3) == 0 && (ldgx & 3) == 0 && (F & 3) == 0) ? 1024 : 256;
  for (int c = 0; c < F; c += panel) {
    const int w = (F - c) < panel ? (F - c) : panel;
    HMP_TRY((segment_mean_dispatch<false>(d_gout + c, ldg, w, plan, d_gx + c, ldgx, (hipStream_t)stream)));
  }
  return HMP_OK;
}
