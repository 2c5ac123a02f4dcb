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
#include <vector>

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

// =============================================================================================================================
// One-launch batch collation (round 2): everything a DataLoader step does between two training steps -- choose B graphs, lay
// their node rows / edge lists out back to back, shift the edge indices -- as ONE host call that issues ONE kernel.  The
// dataset description (which packed arrays exist, their per-graph offset vectors) is fixed at creation; per batch the host
// sums B counts per offset vector (a few hundred integer adds), writes them + the selection into the kernel's argument block
// (small batches) or a pinned ring slot copied asynchronously (large ones), and launches.  No allocation, no synchronisation.
// =============================================================================================================================
namespace hmp {

constexpr int CB_MAX_ITEMS = 20;
constexpr int CB_MAX_SLOTS = 12;
constexpr int CB_INLINE_WORDS = 356;  // int64 words of per-batch tables carried by value in the kernel argument block (whole block < 4 KB)

struct CbItem {
  const uint32_t* src;    // rows: packed rows (4-byte units); edges: int64 edge_index [2][src_total] viewed as units
  uint32_t* dst;
  const int64_t* ptr;     // [G + 1] per-graph offsets into src (device)
  int64_t src_total;      // edges: E_total
  int row_units;          // rows: 4-byte units per row; edges: 0
  int slot, slot_src, slot_dst;
  int block_start;
};
struct CbArgs {
  int n_items, B, total_blocks, n_slots;
  const int64_t* tables;  // device tables (large batches), null: inline
  int64_t* off_out;       // optional: the offset vectors are also written here, [n_slots][off_stride] (Batch.ptr of every slot)
  int off_stride;
  CbItem item[CB_MAX_ITEMS];
  int64_t inl[CB_INLINE_WORDS];  // [n_slots][B + 1] offsets, then sel packed as int64
};

static_assert(sizeof(CbArgs) <= 4096, "kernel argument block");

__global__ __launch_bounds__(256) void collate_batch_kernel(const CbArgs a) {
  int ii = 0;
  while (ii + 1 < a.n_items && (int)blockIdx.x >= a.item[ii + 1].block_start) ++ii;
  const CbItem& I = a.item[ii];
  const int B = a.B;
  const int64_t* tab = a.tables ? a.tables : a.inl;
  const int64_t* off = tab + (int64_t)I.slot * (B + 1);
  const int64_t* sel = tab + (int64_t)a.n_slots * (B + 1);
  const int nb = (ii + 1 < a.n_items ? a.item[ii + 1].block_start : a.total_blocks) - I.block_start;
  const int64_t n_out = off[B];
  if (a.off_out && blockIdx.x == 0)
    for (int q = threadIdx.x; q < a.n_slots * (B + 1); q += 256) a.off_out[(int64_t)(q / (B + 1)) * a.off_stride + q % (B + 1)] = tab[q];
  if (I.row_units >= 32) {
    // wide rows (features): one wavefront per row -- the batch position of the row is found ONCE (wave-uniform search in the
    // offset table), then 64 lanes copy the row with coalesced 8-byte (even widths) or 4-byte accesses
    const int lane = threadIdx.x & 63;
    const int64_t w0 = (int64_t)((int)blockIdx.x - I.block_start) * 4 + (threadIdx.x >> 6);
    for (int64_t row = w0; row < n_out; row += (int64_t)nb * 4) {
      const int b = __builtin_amdgcn_readfirstlane(find_seg(off, B, row));
      const int64_t srow = I.ptr[sel[b]] + (row - off[b]);
      const uint32_t* sp = I.src + srow * I.row_units;
      uint32_t* dp = I.dst + row * I.row_units;
      if ((I.row_units & 1) == 0 && ((reinterpret_cast<uintptr_t>(sp) | reinterpret_cast<uintptr_t>(dp)) & 7) == 0) {
        for (int u = lane; u < (I.row_units >> 1); u += 64) reinterpret_cast<uint2*>(dp)[u] = reinterpret_cast<const uint2*>(sp)[u];
      } else {
        for (int u = lane; u < I.row_units; u += 64) dp[u] = sp[u];
      }
    }
  } else if (I.row_units > 0) {
    // narrow rows (labels, positions, edge attributes): one thread per row
    for (int64_t row = (int64_t)((int)blockIdx.x - I.block_start) * 256 + threadIdx.x; row < n_out; row += (int64_t)nb * 256) {
      const int b = find_seg(off, B, row);
      const int64_t srow = I.ptr[sel[b]] + (row - off[b]);
      for (int u = 0; u < I.row_units; ++u) I.dst[row * I.row_units + u] = I.src[srow * I.row_units + u];
    }
  } else {
    const int64_t* src = reinterpret_cast<const int64_t*>(I.src);
    int64_t* dst = reinterpret_cast<int64_t*>(I.dst);
    const int64_t* os = tab + (int64_t)I.slot_src * (B + 1);
    const int64_t* od = tab + (int64_t)I.slot_dst * (B + 1);
    for (int64_t g = (int64_t)((int)blockIdx.x - I.block_start) * 256 + threadIdx.x; g < n_out; g += (int64_t)nb * 256) {
      const int b = find_seg(off, B, g);
      const int64_t se = I.ptr[sel[b]] + (g - off[b]);
      dst[g] = src[se] + os[b];
      dst[n_out + g] = src[I.src_total + se] + od[b];
    }
  }
}

}  // namespace hmp

struct hmp_collator {
  int n_slots = 0, n_items = 0;
  std::vector<std::vector<int64_t>> slot_ptr;  // host copies of the [G + 1] offset vectors
  hmp::CbItem item[hmp::CB_MAX_ITEMS];
  int64_t n_graphs = 0;
  // large batches: pinned ring + device tables
  static constexpr int RING = 8;
  int64_t* pinned[RING] = {nullptr};
  int64_t* dev[RING] = {nullptr};
  hipEvent_t done[RING] = {nullptr};
  size_t cap_words = 0;
  int cur = 0;
};

extern "C" int hmp_collator_create(int32_t n_slots, const int64_t* const* h_slot_ptr, int64_t n_graphs, int32_t n_items,
                                   const hmp_collate_item* items, hmp_collator** out) {
  using namespace hmp;
  HMP_CHECK_ARG(out && h_slot_ptr && items && n_graphs > 0, "hmp_collator_create: null / empty argument");
  HMP_CHECK_ARG(n_slots >= 1 && n_slots <= CB_MAX_SLOTS && n_items >= 1 && n_items <= CB_MAX_ITEMS, "hmp_collator_create: %d slots / %d items (max %d / %d)",
                n_slots, n_items, CB_MAX_SLOTS, CB_MAX_ITEMS);
  hmp_collator* c = new hmp_collator;
  c->n_slots = n_slots; c->n_items = n_items; c->n_graphs = n_graphs;
  for (int s = 0; s < n_slots; ++s) c->slot_ptr.emplace_back(h_slot_ptr[s], h_slot_ptr[s] + n_graphs + 1);
  for (int i = 0; i < n_items; ++i) {
    const hmp_collate_item& it = items[i];
    if (!(it.d_src && it.d_ptr && it.slot >= 0 && it.slot < n_slots && it.row_bytes >= 0 && (it.row_bytes & 3) == 0 &&
          (it.row_bytes > 0 || (it.slot_src >= 0 && it.slot_src < n_slots && it.slot_dst >= 0 && it.slot_dst < n_slots)))) {
      delete c;
      HMP_FAIL(HMP_E_ARG, "hmp_collator_create: item %d is malformed", i);
    }
    CbItem& I = c->item[i];
    I.src = (const uint32_t*)it.d_src; I.dst = nullptr; I.ptr = it.d_ptr; I.src_total = it.src_total;
    I.row_units = (int)(it.row_bytes / 4); I.slot = it.slot; I.slot_src = it.slot_src; I.slot_dst = it.slot_dst; I.block_start = 0;
  }
  *out = c;
  return HMP_OK;
}

extern "C" void hmp_collator_destroy(hmp_collator* c) {
  if (!c) return;
  for (int i = 0; i < hmp_collator::RING; ++i) {
    if (c->pinned[i]) (void)hipHostFree(c->pinned[i]);
    if (c->dev[i]) (void)hipFree(c->dev[i]);
    if (c->done[i]) (void)hipEventDestroy(c->done[i]);
  }
  delete c;
}

extern "C" int hmp_collator_run(hmp_collator* c, const int32_t* h_sel, int32_t B, void* const* d_dst, const int64_t* dst_capacity,
                                int64_t* h_totals, int64_t* d_offsets_out, int32_t offsets_stride, void* stream) {
  using namespace hmp;
  HMP_CHECK_ARG(c && h_sel && d_dst && dst_capacity && h_totals && B > 0, "hmp_collator_run: null / empty argument");
  hipStream_t st = (hipStream_t)stream;
  CbArgs a;
  a.n_items = c->n_items; a.B = B; a.n_slots = c->n_slots; a.tables = nullptr;
  HMP_CHECK_ARG(!d_offsets_out || offsets_stride >= B + 1, "hmp_collator_run: offsets_stride %d < B + 1", offsets_stride);
  a.off_out = d_offsets_out; a.off_stride = offsets_stride;
  const size_t words = (size_t)c->n_slots * (B + 1) + (size_t)B;
  int64_t* tab = a.inl;
  int slot_i = -1;
  if (words > (size_t)CB_INLINE_WORDS) {  // large batch: tables through a pinned ring slot
    slot_i = c->cur;
    c->cur = (c->cur + 1) % hmp_collator::RING;
    if (words > c->cap_words) {  // (re)allocate every slot at the new capacity; in-flight copies first
      HMP_HIP(hipStreamSynchronize(st));
      const size_t cap = words * 2;
      for (int i = 0; i < hmp_collator::RING; ++i) {
        if (c->pinned[i]) HMP_HIP(hipHostFree(c->pinned[i]));
        if (c->dev[i]) HMP_HIP(hipFree(c->dev[i]));
        HMP_HIP(hipHostMalloc((void**)&c->pinned[i], cap * 8, hipHostMallocDefault));
        HMP_HIP(hipMalloc((void**)&c->dev[i], cap * 8));
        if (!c->done[i]) HMP_HIP(hipEventCreateWithFlags(&c->done[i], hipEventDisableTiming));
      }
      c->cap_words = cap;
    } else {
      HMP_HIP(hipEventSynchronize(c->done[slot_i]));  // the copy out of this pinned slot (8 batches ago) has finished
    }
    tab = c->pinned[slot_i];
  }
  for (int s = 0; s < c->n_slots; ++s) {
    int64_t* off = tab + (size_t)s * (B + 1);
    const std::vector<int64_t>& p = c->slot_ptr[s];
    int64_t acc = 0;
    for (int b = 0; b < B; ++b) {
      const int32_t g = h_sel[b];
      HMP_CHECK_ARG(g >= 0 && g < c->n_graphs, "hmp_collator_run: graph id %d outside [0, %lld)", g, (long long)c->n_graphs);
      off[b] = acc;
      acc += p[g + 1] - p[g];
    }
    off[B] = acc;
    h_totals[s] = acc;
  }
  int64_t* sel64 = tab + (size_t)c->n_slots * (B + 1);
  for (int b = 0; b < B; ++b) sel64[b] = h_sel[b];
  int blocks = 0;
  for (int i = 0; i < c->n_items; ++i) {
    a.item[i] = c->item[i];
    a.item[i].dst = (uint32_t*)d_dst[i];
    const int64_t n_out = h_totals[a.item[i].slot];
    HMP_CHECK_ARG(n_out <= dst_capacity[i], "hmp_collator_run: item %d needs %lld rows, buffer holds %lld", i, (long long)n_out, (long long)dst_capacity[i]);
    HMP_CHECK_ARG(n_out == 0 || d_dst[i], "hmp_collator_run: item %d has no output buffer", i);
    // wide rows: a wavefront per row, 2 rows per wavefront and pass; narrow rows / edges: a thread per row / edge
    int64_t nb = a.item[i].row_units >= 32 ? (n_out + 7) / 8 : (n_out + 255) / 256;
    if (nb > 2048) nb = 2048;
    a.item[i].block_start = blocks;
    blocks += (int)nb;
  }
  a.total_blocks = blocks;
  if (slot_i >= 0) {
    HMP_HIP(hipMemcpyAsync(c->dev[slot_i], c->pinned[slot_i], words * 8, hipMemcpyHostToDevice, st));
    HMP_HIP(hipEventRecord(c->done[slot_i], st));
    a.tables = c->dev[slot_i];
  }
  if (blocks == 0) return HMP_OK;
  hipLaunchKernelGGL(collate_batch_kernel, dim3(blocks), dim3(256), 0, st, a);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}
