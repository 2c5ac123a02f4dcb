// Device-side batch collation: the step in front of the path.
//
// The reference collates on the host with PyG's DataLoader (base_training_job.py:164-178 -> Batch.from_data_list) and
// copies the batch to the device every step; at MP3D sizes that costs as much as the GPU step itself.  Here the DATASET is
// resident in HBM as packed per-type arrays (all graphs' node rows back to back, all edge lists back to back with
// graph-local indices, plus [G+1] offset vectors), and a batch is assembled by two gather kernels:
//   rows   out[dst_off[b] + i]        = src[src_ptr[sel[b]] + i]                                  (x, y, pos, edge_attr)
//   edges  out[r][dst_off[b] + j]     = src[r][edge_ptr[sel[b]] + j] + node_off_{r}[b]            (edge_index, r = 0 / 1)
// which is exactly PyG's collation rule: node stores concatenated in batch order, edge indices
// shifted by the cumulative node counts of their endpoint types.  Byte work, HBM-bound, no GEMM.
#include "kernels.h"

namespace hmp {

// segment of flat output element g: last b with dst_off[b] <= g  (dst_off has B + 1 entries, B <= 10^4: ~14 steps)
__device__ __forceinline__ int find_seg(const int64_t* __restrict__ off, int B, int64_t g) {
  int lo = 0, hi = B - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (off[mid] <= g) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// unit = 4 bytes; rows are `row_units` units long
__global__ __launch_bounds__(256) void collate_rows_kernel(const uint32_t* __restrict__ src, int row_units, const int64_t* __restrict__ src_ptr,
                                                           const int32_t* __restrict__ sel, const int64_t* __restrict__ dst_off, int B,
                                                           uint32_t* __restrict__ dst) {
  const int64_t total = dst_off[B] * row_units;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = g / row_units;
    const int b = find_seg(dst_off, B, row);
    const int64_t srow = src_ptr[sel[b]] + (row - dst_off[b]);
    dst[g] = src[srow * row_units + (g - row * row_units)];
  }
}

__global__ __launch_bounds__(256) void collate_edges_kernel(const int64_t* __restrict__ src, int64_t e_total, const int64_t* __restrict__ edge_ptr,
                                                            const int32_t* __restrict__ sel, const int64_t* __restrict__ dst_off,
                                                            const int64_t* __restrict__ off_src, const int64_t* __restrict__ off_dst, int B,
                                                            int64_t* __restrict__ dst) {
  const int64_t e_out = dst_off[B];
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < e_out; g += (int64_t)gridDim.x * blockDim.x) {
    const int b = find_seg(dst_off, B, g);
    const int64_t se = edge_ptr[sel[b]] + (g - dst_off[b]);
    dst[g] = src[se] + off_src[b];
    dst[e_out + g] = src[e_total + se] + off_dst[b];
  }
}

}  // namespace hmp

extern "C" int hmp_collate_rows(const void* d_src, int64_t row_bytes, const int64_t* d_src_ptr, const int32_t* d_sel,
                                const int64_t* d_dst_off, int32_t B, int64_t n_out_rows, void* d_dst, void* stream) {
  using namespace hmp;
  HMP_CHECK_ARG(B >= 0 && n_out_rows >= 0 && row_bytes > 0 && (row_bytes & 3) == 0, "hmp_collate_rows: row_bytes must be a positive multiple of 4");
  if (B == 0 || n_out_rows == 0) return HMP_OK;
  HMP_CHECK_ARG(d_src && d_src_ptr && d_sel && d_dst_off && d_dst, "hmp_collate_rows: null pointer");
  const int64_t units = n_out_rows * (row_bytes / 4);
  const int64_t want = cdiv(units, 256);
  hipLaunchKernelGGL(collate_rows_kernel, dim3((int)(want > 4096 ? 4096 : want)), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)d_src,
                     (int)(row_bytes / 4), d_src_ptr, d_sel, d_dst_off, B, (uint32_t*)d_dst);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

extern "C" int hmp_collate_edges(const int64_t* d_src, int64_t e_total, const int64_t* d_edge_ptr, const int32_t* d_sel,
                                 const int64_t* d_dst_off, const int64_t* d_off_src, const int64_t* d_off_dst, int32_t B,
                                 int64_t e_out, int64_t* d_dst, void* stream) {
  using namespace hmp;
  HMP_CHECK_ARG(B >= 0 && e_out >= 0 && e_total >= 0, "hmp_collate_edges: negative size");
  if (B == 0 || e_out == 0) return HMP_OK;
  HMP_CHECK_ARG(d_src && d_edge_ptr && d_sel && d_dst_off && d_off_src && d_off_dst && d_dst, "hmp_collate_edges: null pointer");
  const int64_t want = cdiv(e_out, 256);
  hipLaunchKernelGGL(collate_edges_kernel, dim3((int)(want > 4096 ? 4096 : want)), dim3(256), 0, (hipStream_t)stream, d_src, e_total,
                     d_edge_ptr, d_sel, d_dst_off, d_off_src, d_off_dst, B, d_dst);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}
