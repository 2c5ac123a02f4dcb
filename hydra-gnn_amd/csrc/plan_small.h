// Single-launch plan build for small batches: the per-part body, shared by plan_small_kernel (plan.hip) and the front
// kernel of the fused step (front.hip).
#pragma once
#include "kernels.h"

namespace hmp {

// ---------------------------------------------------------------------------------------------------------------
// Small batches (all edge types together <= PS_MAX_EDGES edges): the whole build in ONE launch, no global atomics,
// no counters to keep clean.  A 1024-thread block owns one (edge type, direction, row range) part: it reads ALL edges
// of its type twice from L2 (a few 10^4 edges: a handful of iterations per thread) and keeps everything about its
// <= PS_ROWS rows in LDS --
//   pass 1  histogram of its rows (LDS atomics) + count of the edges that belong to earlier rows (-> its rowptr base)
//   scan    exclusive scan of the histogram -> rowptr (global) and the local row starts
//   pass 2  every edge of the part takes a slot in its row (LDS atomic cursor: arbitrary order inside the row)
//   rank    one thread per slot counts the smaller edge ids of its row -> stable position, writes col / eid
// The result is the same stable order as the multi-launch path (and bit-identical from run to run).
// ---------------------------------------------------------------------------------------------------------------
constexpr int PS_ROWS = 1024;        // rows per part
constexpr int PS_TMP = 12 * 1024;    // edge slots of a part kept in LDS (parts with more edges use the global scratch)
constexpr int PS_UB = 8;             // edges per thread in flight in the two passes over the edge list
constexpr int64_t PS_MAX_EDGES = 65536;

struct PlanSmallArgs {
  PlanBatch pb;
  int part_start[2 * HMP_MAX_EDGE_TYPES + 1];  // first block of (job, dir)
  int rows_per_part[2 * HMP_MAX_EDGE_TYPES];
};

constexpr int PS_ROWS_SLICED = 256;  // rows per part when the part reads only its graphs' edges (plan_small_part)
#ifndef PS_KT
#define PS_KT(i)  // phase stamps of one part (front.hip defines it in the profiling build)
#endif
constexpr int PS_RC = 24;             // register-cached variant: edges per thread (every edge type <= 24 * 1024 edges)

// 130 KB of LDS (a gfx950 workgroup may take up to 160 KB): one block per CU, ~10-30 blocks per launch.
// RC: the block's 1024 threads hold ALL edges of the type in registers (one global round trip; pass 2 re-reads nothing).
// `blk` = index of the part inside the plan's block range; `lds` = PS_LDS_BYTES of shared memory (16-byte aligned).
constexpr int PS_LDS_BYTES = (PS_ROWS + PS_ROWS + 1 + 2 * PS_TMP + 16 + 2 + 1 + 4) * 4 + PS_TMP * 2;

template <bool RC>
__device__ __forceinline__ void plan_small_part(const PlanJob& J, int dir, int part, int rpp, bool last_part, int need_tpos, int* status,
                                                char* lds) {
  int* cnt = reinterpret_cast<int*>(lds);
  int* lrow = cnt + PS_ROWS;
  int* ltmp = lrow + PS_ROWS + 1;   // edge id per slot
  int* lcol = ltmp + PS_TMP;        // the other endpoint of that edge
  int* wsum = lcol + PS_TMP;
  int* s_scal = wsum + 16;          // [0] base, [1] total, [3..6] the part's edge range and key range (graph-sorted edge lists)
  unsigned short* lkey = reinterpret_cast<unsigned short*>(s_scal + 7);  // row (relative to the part) of the slot
  int& s_base = s_scal[0];
  int& s_total = s_scal[1];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n_rows = dir ? J.n_src : J.n_dst;
  const int r0 = part * rpp;
  const int r1 = min(n_rows, r0 + rpp);
  const int nr = max(r1 - r0, 0);
  const int E = (int)J.E;
  const int64_t* __restrict__ es = J.ei;
  const int64_t* __restrict__ ed = J.ei + J.E;
  const int n_src = J.n_src, n_dst = J.n_dst;

  PS_KT(8);
  for (int i = tid; i < PS_ROWS; i += 1024) cnt[i] = 0;
  // Graph-sorted edge list (PlanJob::gp_*): the edges whose key row lies in [r0, r1) are those of the graphs that own these rows,
  // a contiguous slice [e_lo, e_hi) of the list -- every earlier edge has a smaller key, every later one a larger.  The part
  // reads the slice instead of the whole list (what made a part cost ~18 us at 17k edges: all of them through one CU).  One
  // round trip: thread g looks at graph g's row and edge offsets; the graph holding r0 gives e_lo, the one holding r1 - 1 e_hi.
  const bool sliced = J.gp_edge != nullptr && nr > 0;  // block-uniform
  int bad = 0;  // bit 0: an endpoint out of range; bit 1: offsets / edge order not as vouched for
  if (tid == 0) { s_scal[3] = 0; s_scal[4] = sliced ? 0 : E; s_scal[5] = 0; s_scal[6] = n_rows; }
  __syncthreads();
  if (sliced) {
    const int64_t* __restrict__ gp = dir ? J.gp_src : J.gp_dst;
    for (int g = tid; g < J.n_graphs; g += 1024) {
      const int64_t a = gp[g], b = gp[g + 1];
      const int64_t ea = J.gp_edge[g], eb = J.gp_edge[g + 1];
      if (a <= r0 && r0 < b) { s_scal[3] = (int)ea; s_scal[5] = (int)a; }
      if (a < r1 && r1 <= b) { s_scal[4] = (int)eb; s_scal[6] = (int)b; }
      if (g + 1 == J.n_graphs && (b != n_rows || eb != E)) bad = 2;  // offsets that do not describe this batch
    }
    __syncthreads();
  }
  // (block-uniform values read from LDS: kept in scalar registers -- the register-cached variant has none to spare)
  const int e_lo = __builtin_amdgcn_readfirstlane(min(max(s_scal[3], 0), E));
  const int e_hi = __builtin_amdgcn_readfirstlane(min(max(s_scal[4], e_lo), E));
  const int key_lo = __builtin_amdgcn_readfirstlane(s_scal[5]), key_hi = __builtin_amdgcn_readfirstlane(s_scal[6]);
  // pass 1: PS_UB (clamped) edges per thread in flight
  int below = tid == 0 ? e_lo : 0;  // (the edges in front of the slice all lie below r0)
  int cs[RC ? PS_RC : 1], cd[RC ? PS_RC : 1];  // RC: endpoints of edge e_lo + tid + i * 1024 (source -1: not an edge / dropped)
  if constexpr (RC) {
#pragma unroll
    for (int i = 0; i < PS_RC; ++i) {
      const int e = e_lo + tid + i * 1024;
      const int ec = max(min(e, e_hi - 1), 0);
      int64_t s = -1, d = -1;
      if (e_hi > e_lo) { s = es[ec]; d = ed[ec]; }  // block-uniform condition
      const bool in = e < e_hi;
      const bool ok = in && s >= 0 && s < n_src && d >= 0 && d < n_dst;
      if (in && !ok) bad |= 1;
      cs[i] = ok ? (int)s : -1;
      cd[i] = (int)d;
    }
#pragma unroll
    for (int i = 0; i < PS_RC; ++i) {
      if (cs[i] < 0) continue;
      const int key = dir ? cs[i] : cd[i];
      if (key < key_lo || key >= key_hi) bad |= 2;
      if (key < r0) ++below;
      else if (key < r1) atomicAdd(&cnt[key - r0], 1);
    }
  } else
  for (int e0 = e_lo + tid; e0 < e_hi; e0 += PS_UB * 1024) {
    int64_t sv[PS_UB], dv[PS_UB];
#pragma unroll
    for (int u = 0; u < PS_UB; ++u) {
      const int e = min(e0 + u * 1024, e_hi - 1);
      sv[u] = es[e];
      dv[u] = ed[e];
    }
#pragma unroll
    for (int u = 0; u < PS_UB; ++u) {
      if (e0 + u * 1024 >= e_hi) break;
      const int64_t s = sv[u], d = dv[u];
      if (s < 0 || s >= n_src || d < 0 || d >= n_dst) { bad |= 1; continue; }
      const int key = (int)(dir ? s : d);
      if (key < key_lo || key >= key_hi) bad |= 2;
      if (key < r0) ++below;
      else if (key < r1) atomicAdd(&cnt[key - r0], 1);
    }
  }
  PS_KT(9);
  if ((bad & 1) && status && (sliced || (part == 0 && dir == 0))) atomicOr(status, 1);
  if ((bad & 2) && status) atomicOr(status, 4);  // an edge of the slice outside its graphs' rows: the list is not graph-sorted
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) below += __shfl_xor(below, o);
  if (lane == 0) wsum[w] = below;
  __syncthreads();
  if (tid == 0) {
    int b = 0;
    for (int q = 0; q < 16; ++q) b += wsum[q];
    s_base = b;
  }
  __syncthreads();
  const int base = s_base;
  // scan (one row per thread)
  {
    const int v = tid < nr ? cnt[tid] : 0;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    __syncthreads();  // everybody has read wsum / s_base
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int woff = 0;
    for (int q = 0; q < w; ++q) woff += wsum[q];
    const int excl = woff + x - v;
    if (tid < nr) {
      lrow[tid] = excl;
      (dir ? J.t_rowptr : J.rowptr)[r0 + tid] = base + excl;
      if (!dir && J.degf) J.degf[r0 + tid] = 1.f / (float)(v > 1 ? v : 1);
      cnt[tid] = 0;  // becomes the row cursor of pass 2
    }
    if (tid == 1023) s_total = woff + x;
    __syncthreads();
    if (tid == 0) {
      lrow[nr] = s_total;
      if (last_part) (dir ? J.t_rowptr : J.rowptr)[n_rows] = base + s_total;
    }
  }
  __syncthreads();
  PS_KT(10);
  const int total = s_total;
  const bool in_lds = total <= PS_TMP;
  int* __restrict__ gtmp = (dir ? J.tmp_out : J.tmp_in) + base;
  int* __restrict__ ell = dir ? J.t_ell : J.ell;  // first ELL_W ids of every row, next to the CSR arrays (kernels.h)
  // pass 2: every edge of the part takes a slot in its row
  if constexpr (RC) {
#pragma unroll
    for (int i = 0; i < PS_RC; ++i) {
      if (cs[i] < 0) continue;
      const int key = dir ? cs[i] : cd[i];
      if (key < r0 || key >= r1) continue;
      const int e = e_lo + tid + i * 1024;
      const int slot = lrow[key - r0] + atomicAdd(&cnt[key - r0], 1);
      if (in_lds) {
        ltmp[slot] = e;
        lcol[slot] = dir ? cd[i] : cs[i];
        lkey[slot] = (unsigned short)(key - r0);
      } else {
        gtmp[slot] = e;
      }
    }
  } else
  for (int e0 = e_lo + tid; e0 < e_hi; e0 += PS_UB * 1024) {
    int64_t sv[PS_UB], dv[PS_UB];
#pragma unroll
    for (int u = 0; u < PS_UB; ++u) {
      const int e = min(e0 + u * 1024, e_hi - 1);
      sv[u] = es[e];
      dv[u] = ed[e];
    }
#pragma unroll
    for (int u = 0; u < PS_UB; ++u) {
      const int e = e0 + u * 1024;
      if (e >= e_hi) break;
      const int64_t s = sv[u], d = dv[u];
      if (s < 0 || s >= n_src || d < 0 || d >= n_dst) continue;
      const int key = (int)(dir ? s : d);
      if (key < r0 || key >= r1) continue;
      const int slot = lrow[key - r0] + atomicAdd(&cnt[key - r0], 1);
      if (in_lds) {
        ltmp[slot] = e;
        lcol[slot] = (int)(dir ? d : s);
        lkey[slot] = (unsigned short)(key - r0);
      } else {
        gtmp[slot] = e;
      }
    }
  }
  __syncthreads();
  PS_KT(11);
  // rank: one thread per slot counts the smaller edge ids of its row
  if (in_lds) {
    for (int q = tid; q < total; q += 1024) {
      const int r = lkey[q];
      const int b = lrow[r], en = lrow[r + 1];
      const int mine = ltmp[q];
      int rank = 0;
      for (int i = b; i < en; ++i) rank += (ltmp[i] < mine) ? 1 : 0;
      const int pos = base + b + rank;
      if (dir == 0) {
        J.eid[pos] = mine;
        J.col[pos] = lcol[q];
        if (need_tpos) J.pos_of_eid[mine] = pos;
      } else {
        if (need_tpos) J.t_eid[pos] = mine;
        J.t_col[pos] = lcol[q];
      }
      if (ell && rank < ELL_W) ell[(int64_t)(r0 + r) * ELL_W + rank] = lcol[q];
    }
  } else {
    for (int q = tid; q < total; q += 1024) {
      int lo = 0, hi = nr - 1;  // row of slot q: last r with lrow[r] <= q
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (lrow[mid] <= q) lo = mid; else hi = mid - 1;
      }
      const int b = lrow[lo], en = lrow[lo + 1];
      const int mine = gtmp[q];
      int rank = 0;
      for (int i = b; i < en; ++i) rank += (gtmp[i] < mine) ? 1 : 0;
      const int pos = base + b + rank;
      if (dir == 0) {
        J.eid[pos] = mine;
        J.col[pos] = (int)es[mine];
        if (need_tpos) J.pos_of_eid[mine] = pos;
      } else {
        if (need_tpos) J.t_eid[pos] = mine;
        J.t_col[pos] = (int)ed[mine];
      }
      if (ell && rank < ELL_W) ell[(int64_t)(r0 + lo) * ELL_W + rank] = (int)(dir ? ed[mine] : es[mine]);
    }
  }
  PS_KT(12);
}


template <bool RC>
__device__ __forceinline__ void plan_small_block(const PlanSmallArgs& a, int* status, int blk, char* lds) {
  int jd = 0;
  while (jd + 1 < 2 * a.pb.n && blk >= a.part_start[jd + 1]) ++jd;
  plan_small_part<RC>(a.pb.j[jd >> 1], jd & 1, blk - a.part_start[jd], a.rows_per_part[jd], blk + 1 == a.part_start[jd + 1],
                      a.pb.need_tpos, status, lds);
}

// host: block layout of the single-launch build; returns the number of blocks (0: nothing to do)
inline int plan_small_layout(const PlanBatch& pb, int* part_start /*[2n+1]*/, int* rows_per_part /*[2n]*/) {
  int blocks = 0;
  for (int jd = 0; jd < 2 * pb.n; ++jd) {
    const PlanJob& J = pb.j[jd >> 1];
    const int nrows = (jd & 1) ? J.n_src : J.n_dst;
    // graph-sorted edge lists: a part's cost follows ITS rows (it reads only their graphs' edges), so many small parts finish sooner
    const int rows_max = J.gp_edge ? PS_ROWS_SLICED : PS_ROWS;
    const int parts = nrows > 0 ? cdiv(nrows, rows_max) : 1;
    part_start[jd] = blocks;
    rows_per_part[jd] = nrows > 0 ? cdiv(nrows, parts) : 1;
    blocks += parts;
  }
  part_start[2 * pb.n] = blocks;
  return blocks;
}
inline bool plan_small_fits_rc(const PlanBatch& pb) {
  int64_t emax = 0;
  bool sliced = pb.n > 0;
  for (int j = 0; j < pb.n; ++j) {
    emax = pb.j[j].E > emax ? pb.j[j].E : emax;
    sliced = sliced && (pb.j[j].gp_edge != nullptr || pb.j[j].E == 0);
  }
  // every part reads a slice of a few thousand edges: the register-cached variant would still issue its PS_RC (clamped) loads
  // per thread -- 393 KB through the part's CU whatever the slice holds; the streaming variant issues what the slice needs
  if (sliced) return false;
  return emax <= (int64_t)PS_RC * 1024;
}

}  // namespace hmp
