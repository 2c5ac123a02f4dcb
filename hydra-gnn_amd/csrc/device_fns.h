// Device-side bodies shared by kernels of different translation units (no relocatable device code: header-inline).
#pragma once
#include "kernels.h"

namespace hmp {

// One packed element per lane: item = (packed row, 64-column chunk) of segment S, one wavefront per item.
__device__ __forceinline__ void pack_item(const PackSeg& S, int item, const float* __restrict__ params, float* __restrict__ packed) {
  const int chunks = (S.ld_dst + 63) >> 6;
  const int r = item / chunks;
  if (r >= S.rows_pad) return;
  float* dst = packed + S.dst + (int64_t)r * S.ld_dst;
  const int c = (item % chunks) * 64 + (threadIdx.x & 63);
  if (c >= S.ld_dst) return;
  float v = 0.f;
  if (S.kind == PACK_SUM) {
    if (r < S.rows && c < S.cols)
      for (int q = 0; q < S.nsrc; ++q) v += params[S.src[q] + (int64_t)r * S.ld_src + c];
  } else if (S.kind == PACK_HEADS) {
    const int h = r / S.Cp, cc = r % S.Cp;
    if (h < S.H && cc < S.C && c < S.cols) v = params[S.src[0] + (int64_t)(h * S.C + cc) * S.ld_src + c];
  } else if (S.kind == PACK_ATTDOT) {  // row r = head
    if (r < S.H && c < S.cols) {
      // C-long dot product; 8 terms' loads in flight at a time (clamped), added in index order
      const float* av = params + S.att + r * S.C;
      const float* wv = params + S.src[0] + (int64_t)(r * S.C) * S.ld_src + c;
      for (int cc = 0; cc < S.C; cc += 8) {
        float a8[8], w8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int ci = min(cc + u, S.C - 1);
          a8[u] = av[ci];
          w8[u] = wv[(int64_t)ci * S.ld_src];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (cc + u < S.C) v += a8[u] * w8[u];
      }
    }
  } else {  // PACK_ATTDOT_T: row r = edge-attribute dimension d, column c = head
    if (r < S.rows && c < S.H) {
      const float* av = params + S.att + c * S.C;
      const float* wv = params + S.src[0] + (int64_t)(c * S.C) * S.ld_src + r;
      for (int cc = 0; cc < S.C; cc += 8) {
        float a8[8], w8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int ci = min(cc + u, S.C - 1);
          a8[u] = av[ci];
          w8[u] = wv[(int64_t)ci * S.ld_src];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (cc + u < S.C) v += a8[u] * w8[u];
      }
    }
  }
  dst[c] = v;
}

// One pack block (256 threads): one wavefront per item; `blk` indexes the SegBlocks table (4 items per block).
// Block 0 also starts the step: it bumps the device step counter that dropout and Adam read later in the same step.
__device__ __forceinline__ void pack_block(const PackSeg* __restrict__ segs, const SegBlocks& sb, const float* __restrict__ params,
                                           float* __restrict__ packed, int* step_ctr, int* step_mirror, int blk) {
  if (step_ctr && blk == 0 && threadIdx.x == 0) {
    const int t = *step_ctr + 1;
    *step_ctr = t;
    if (step_mirror) *step_mirror = t;
  }
  int si = 0;
  while (si + 1 < sb.n && blk >= sb.start[si + 1]) ++si;  // wave-uniform scan of the kernarg table
  const PackSeg S = segs[si];
  pack_item(S, (blk - sb.start[si]) * 4 + (int)(threadIdx.x >> 6), params, packed);
}

}  // namespace hmp
