// Aggregate-first SAGE convs ("mean aggregation and projection commute ... aggregate first when it is cheaper").
// A conv whose destination type is much smaller than its source type (objects -> rooms: 10^6 -> 10^4 rows at BASELINE config 5)
// is evaluated in the reference's own order -- [PyG] SAGEConv: out = lin_l(mean_j x_j) -- instead of projecting every source row:
//   forward   M[i, :] = (1 / max(deg_i, 1)) * sum_{k in CSR row i} X[col[k], :]        (seg_mean_rows_kernel: fp32 or bf16 rows in,
//             fp32 out), then a 10^4-row GEMM M * W_l^T into the destination type's block of projected rows;
//   backward  dM = dZ_block * W_l (10^4 rows), and the source rows' gradient  sum_{k in CSC row j} dM[t_col[k], :] / deg(t_col[k])
//             is ADDED in the epilogue of the source type's input-gradient GEMM (GemmProblem::g_*), or -- when the source type has no
//             other conv in the layer, i.e. no such GEMM -- written by seg_mean_rows_t_kernel together with the activation mask.
// What this saves at config 5, per hidden layer: a third of the objects' projection (768 -> 512 stacked columns), the 512 MB block
// of projected rows and the same block of dZ (written once, read by both backward GEMMs).
#include "kernels.h"

namespace hmp {

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const uint16_t* p) {
  const uint2 b = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(b.x << 16), __uint_as_float(b.x & 0xffff0000u), __uint_as_float(b.y << 16), __uint_as_float(b.y & 0xffff0000u));
}

// element-wise form for rows that are not whole aligned 4-vectors (the caller's input features: 306 floats at pitch 306)
__device__ __forceinline__ float4 ld4_edge(const float* p, int n) {
  return make_float4(n > 0 ? p[0] : 0.f, n > 1 ? p[1] : 0.f, n > 2 ? p[2] : 0.f, n > 3 ? p[3] : 0.f);
}
__device__ __forceinline__ float4 ld4_edge(const uint16_t* p, int n) {
  float v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = q < n ? __uint_as_float((uint32_t)p[q] << 16) : 0.f;
  return make_float4(v[0], v[1], v[2], v[3]);
}

// one wavefront per destination row, 4 elements per lane (rows wider than 256 take more passes), 8 neighbour rows in flight
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void seg_mean_rows_kernel(const T* __restrict__ x, int ldx, int F, const int* __restrict__ rowptr,
                                                            const int* __restrict__ col, int n_rows, float* __restrict__ m, int ldm) {
  const int row = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int lane = threadIdx.x & 63;
  const int b = rowptr[row], e = rowptr[row + 1];
  const float scale = 1.f / (float)((e - b) > 1 ? (e - b) : 1);
  for (int c0 = lane * 4; c0 < F; c0 += 256) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = b; k < e; k += 8) {
      int j[8];
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) j[u] = col[min(k + u, e - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if constexpr (VEC) v[u] = ld4(x + (int64_t)j[u] * ldx + c0);
        else v[u] = ld4_edge(x + (int64_t)j[u] * ldx + c0, F - c0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k + u < e) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    *reinterpret_cast<float4*>(m + (int64_t)row * ldm + c0) = make_float4(acc.x * scale, acc.y * scale, acc.z * scale, acc.w * scale);
  }
}

__device__ __forceinline__ float t_mask(float h, int act, bool keep, float scale) {
  if (!keep) return 0.f;
  if (act == HMP_ACT_RELU) return h > 0.f ? scale : 0.f;
  if (act == HMP_ACT_ELU) return h > 0.f ? scale : (h + scale);
  return scale;
}

// G[j, :] = mask(H[j, :]) . sum_{k in CSC row j} dM[t_col[k], :] * degf[t_col[k]].  One wavefront walks TR consecutive source rows at a
// time with every row's chain (extent -> destination id -> reciprocal degree + dM row) and its H row requested together: a source row
// has one or two such edges (the conv qualified by E <= 2 N_src), so a wave that owns a single row is a chain of four dependent round
// trips with one 512-byte row in flight (measured 2.5 TB/s at config 5; TR rows per wave keep TR of them in flight).
constexpr int TR = 4;
template <bool HB, bool GB>
__global__ __launch_bounds__(256) void seg_mean_rows_t_kernel(const float* __restrict__ dm, int lddm, int F, const int* __restrict__ t_rowptr,
                                                              const int* __restrict__ t_col, const float* __restrict__ degf, int n_rows,
                                                              const void* __restrict__ h, int ldh, int act, int drop_on, float dscale,
                                                              void* __restrict__ g, int ldg) {
  const int row0 = (blockIdx.x * 4 + (int)(threadIdx.x >> 6)) * TR;
  if (row0 >= n_rows) return;
  const int lane = threadIdx.x & 63;
  int b[TR], e[TR];
#pragma unroll
  for (int u = 0; u < TR; ++u) {
    const int r = min(row0 + u, n_rows - 1);
    b[u] = t_rowptr[r];
    e[u] = row0 + u < n_rows ? t_rowptr[r + 1] : b[u];
  }
  for (int c0 = lane * 4; c0 < F; c0 += 256) {
    float4 hv[TR];
    if (h) {
#pragma unroll
      for (int u = 0; u < TR; ++u) {
        const int r = min(row0 + u, n_rows - 1);
        if constexpr (HB) hv[u] = ld4(reinterpret_cast<const uint16_t*>(h) + (int64_t)r * ldh + c0);
        else hv[u] = ld4(reinterpret_cast<const float*>(h) + (int64_t)r * ldh + c0);
      }
    }
    float4 acc[TR];
    int i0[TR];
#pragma unroll
    for (int u = 0; u < TR; ++u) i0[u] = e[u] > b[u] ? t_col[b[u]] : -1;  // first edge of every row: the TR chains run side by side
    float wd0[TR];
    float4 v0[TR];
#pragma unroll
    for (int u = 0; u < TR; ++u) {
      wd0[u] = 0.f;
      v0[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i0[u] >= 0) { wd0[u] = degf[i0[u]]; v0[u] = ld4(dm + (int64_t)i0[u] * lddm + c0); }
    }
#pragma unroll
    for (int u = 0; u < TR; ++u) {
      acc[u] = make_float4(wd0[u] * v0[u].x, wd0[u] * v0[u].y, wd0[u] * v0[u].z, wd0[u] * v0[u].w);
      for (int k = b[u] + 1; k < e[u]; ++k) {  // further edges of the row, in edge order
        const int i = t_col[k];
        const float wd = degf[i];
        const float4 v = ld4(dm + (int64_t)i * lddm + c0);
        acc[u].x += wd * v.x; acc[u].y += wd * v.y; acc[u].z += wd * v.z; acc[u].w += wd * v.w;
      }
    }
#pragma unroll
    for (int u = 0; u < TR; ++u) {
      if (row0 + u >= n_rows) break;
      float4 a4 = acc[u];
      if (h) {
        const float hh[4] = {hv[u].x, hv[u].y, hv[u].z, hv[u].w};
        float f[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) f[q] = t_mask(hh[q], act, !drop_on || __float_as_uint(hh[q]) != 0x80000000u, dscale);
        a4.x *= f[0]; a4.y *= f[1]; a4.z *= f[2]; a4.w *= f[3];
      }
      if constexpr (GB) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 o;
        o[0] = (__bf16)a4.x; o[1] = (__bf16)a4.y; o[2] = (__bf16)a4.z; o[3] = (__bf16)a4.w;
        *reinterpret_cast<bf16x4*>(reinterpret_cast<uint16_t*>(g) + (int64_t)(row0 + u) * ldg + c0) = o;
      } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(g) + (int64_t)(row0 + u) * ldg + c0) = a4;
      }
    }
  }
}

}  // namespace

int seg_mean_rows_launch(const void* x, int ldx, int x_bf16, int F, const int* rowptr, const int* col, int n_rows, float* m, int ldm,
                         hipStream_t st) {
  if (n_rows <= 0) return HMP_OK;
  HMP_CHECK_ARG((ldm & 3) == 0 && ldm >= align4(F) && (reinterpret_cast<uintptr_t>(m) & 15) == 0, "seg_mean_rows: output pitch %d for %d columns", ldm, F);
  // whole aligned 4-vectors (every buffer the engine owns); else element by element
  const bool vec = (ldx & 3) == 0 && ldx >= align4(F) && (reinterpret_cast<uintptr_t>(x) & (x_bf16 ? 7 : 15)) == 0;
  const dim3 grid(cdiv(n_rows, 4)), block(256);
#define HMP_SM_LAUNCH(T_, V_) \
  hipLaunchKernelGGL((seg_mean_rows_kernel<T_, V_>), grid, block, 0, st, reinterpret_cast<const T_*>(x), ldx, F, rowptr, col, n_rows, m, ldm)
  if (x_bf16 && vec) HMP_SM_LAUNCH(uint16_t, true);
  else if (x_bf16) HMP_SM_LAUNCH(uint16_t, false);
  else if (vec) HMP_SM_LAUNCH(float, true);
  else HMP_SM_LAUNCH(float, false);
#undef HMP_SM_LAUNCH
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

int seg_mean_rows_t_launch(const float* dm, int lddm, int F, const int* t_rowptr, const int* t_col, const float* degf, int n_rows,
                           const void* h, int ldh, int h_bf16, int act, int drop_on, float dscale, void* g, int ldg, int g_bf16, hipStream_t st) {
  if (n_rows <= 0) return HMP_OK;
  HMP_CHECK_ARG((F & 3) == 0 && (lddm & 3) == 0 && (ldg & 3) == 0 && (!h || (ldh & 3) == 0), "seg_mean_rows_t: rows must be whole 4-element vectors");
  const dim3 grid(cdiv(n_rows, 4 * TR)), block(256);
#define HMP_T_LAUNCH(HB_, GB_) \
  hipLaunchKernelGGL((seg_mean_rows_t_kernel<HB_, GB_>), grid, block, 0, st, dm, lddm, F, t_rowptr, t_col, degf, n_rows, h, ldh, act, drop_on, dscale, g, ldg)
  if (h_bf16 && g_bf16) HMP_T_LAUNCH(true, true);
  else if (h_bf16) HMP_T_LAUNCH(true, false);
  else if (g_bf16) HMP_T_LAUNCH(false, true);
  else HMP_T_LAUNCH(false, false);
#undef HMP_T_LAUNCH
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

}  // namespace hmp
