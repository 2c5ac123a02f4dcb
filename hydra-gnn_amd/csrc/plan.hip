// Graph plan: CSR-by-destination and CSC-by-source neighbour lists for every edge type of a batch, built
// on the device from the PyG `edge_index` (int64 [2][E]) in five short launches that cover ALL edge
// types and both directions at once (scene-graph batches have 4..15 edge types of 10^2..10^7 edges).
//
// Stable order without a sort pass: (1) histogram, (2) exclusive scan, (3) atomic fill into a scratch
// list (unordered inside a row), (4) one wavefront per row ranks the row's edge ids by counting -- rows
// are short (scene graphs: mean in-degree ~6, rooms ~10^2) -- and writes them in ascending edge id, which
// IS the stable order torch.sort(dst, stable=True) gives.  The result is therefore bit-identical from run
// to run although step (3) uses atomics.  (5) links every CSC entry to its CSR position.
#include <atomic>
#include <cstdlib>

#include "plan_small.h"

namespace hmp {

KT_DEFINE(plan)

__device__ __forceinline__ int find_job(const int64_t* start, int n, int64_t g) {
  int j = 0;
  while (j + 1 < n && g >= start[j + 1]) ++j;
  return j;
}

// The value an edge's atomicAdd returns is its arrival slot inside the row: it is parked (coalesced) in pos_of_eid / t_eid --
// both are first written by the rank pass -- so that the fill pass needs no second round of 2 atomics per edge.
// Wave-aggregated: consecutive edges of one wavefront that hit the same counter (edge lists written row by row -- a scene
// graph's object->object edges come grouped by object -- put 16 equal keys in neighbouring lanes) form a run; the run's
// first lane adds the run length once and hands base + offset to the others.  Any distinct slots per row will do: the rank
// pass restores the edge order.  Unsorted lists degenerate to one atomic per lane as before (+ 3 shuffles / 2 ballots).
__device__ __forceinline__ int run_slot(int* counter, int key, int job, bool valid) {
  const int lane = threadIdx.x & 63;
  const int pk = __shfl_up(key, 1), pj = __shfl_up(job, 1);
  const bool pv = __shfl_up((int)valid, 1) != 0;
  const bool head = valid && (lane == 0 || !pv || pk != key || pj != job);
  const unsigned long long heads = __ballot(head), vmask = __ballot(valid);
  const unsigned long long le = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull);
  const int hpos = 63 - __clzll((long long)(heads & le));  // valid lanes only: their run's head is at or below them
  const unsigned long long stop = (heads | ~vmask) & ~le;  // next head or hole above this lane ends the run
  const int next = stop ? __ffsll((long long)stop) - 1 : 64;
  int base = 0;
  if (head) base = atomicAdd(counter, next - lane);
  base = __shfl(base, valid ? hpos : lane);
  return base + (lane - hpos);
}

__global__ __launch_bounds__(256) void plan_hist_kernel(const PlanBatch pb, int* status) {
  const int64_t total = pb.edge_start[pb.n];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned flagged0 = 0u, flagged1 = 0u;  // jobs this wavefront has already flagged as unordered (by destination / by source)
  // wave-uniform trip count: every lane takes part in the shuffles of run_slot
  for (int64_t g0 = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); g0 < total; g0 += stride) {
    const int64_t g = g0 + (threadIdx.x & 63);
    const bool live = g < total;
    const int j = find_job(pb.edge_start, pb.n, live ? g : total - 1);
    const PlanJob& J = pb.j[j];
    const int64_t e = (live ? g : total - 1) - pb.edge_start[j];
    const int64_t s = J.ei[e], d = J.ei[J.E + e];
    const bool ok = s >= 0 && s < J.n_src && d >= 0 && d < J.n_dst;
    if (live && !ok && status) atomicOr(status, 1);
    const bool valid = live && ok;
    {  // ordered by destination / by source?  (the previous edge's endpoints: the same cache lines this wave just read.)  At most
       // ONE store per wavefront and (job, direction) over the whole kernel: millions of stores to one address serialise in L2
       // (measured: plan 1.6 -> 3.6 ms with a store per out-of-order edge)
      const int64_t ps = (live && e > 0) ? J.ei[e - 1] : s, pd = (live && e > 0) ? J.ei[J.E + e - 1] : d;
      const bool b0 = live && (!ok || d < pd), b1 = live && (!ok || s < ps);
      // (lanes of one wavefront may sit in two jobs at a job boundary: the mask is per job)
      if (((flagged0 >> j) & 1u) == 0u && b0) { pb.flags_all[2 * j] = pb.build_id; }
      if (((flagged1 >> j) & 1u) == 0u && b1) { pb.flags_all[2 * j + 1] = pb.build_id; }
      if (__any((b0 && ((flagged0 >> j) & 1u) == 0u) || (b1 && ((flagged1 >> j) & 1u) == 0u))) {
        for (int jj = 0; jj < pb.n; ++jj) {  // wave-uniform update of the per-job "already flagged" masks
          if (__any(b0 && j == jj)) flagged0 |= 1u << jj;
          if (__any(b1 && j == jj)) flagged1 |= 1u << jj;
        }
      }
    }
    const int din = run_slot(&J.cnt_in[valid ? d : 0], (int)d, j, valid);
    const int dout = run_slot(&J.cnt_out[valid ? s : 0], (int)s, j, valid);
    if (valid) {
      J.pos_of_eid[e] = din;
      J.t_eid[e] = dout;
    }
  }
}

// exclusive scan of the counts of every (job, direction) in three short launches over 4096-row segments (a single block
// per (job, direction) took 1.06 ms for the 10^6-row types of config 5):
//   partial  each segment's total  -> parked in ptr[segment start]
//   tops     one block per (job, direction): exclusive scan of the parked totals (<= a few hundred), total -> ptr[n]
//   final    each segment: local exclusive scan + its base -> rowptr
constexpr int SCAN_SEG = 4096;
struct ScanSegs {
  int start[2 * HMP_MAX_EDGE_TYPES + 1];  // first block of (job, dir)
};

__device__ __forceinline__ int block_sum_1024(int v, int* wsum) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  if (lane == 0) wsum[w] = v;
  __syncthreads();
  int t = 0;
  for (int q = 0; q < 16; ++q) t += wsum[q];
  __syncthreads();
  return t;
}

__global__ __launch_bounds__(1024) void plan_scan_partial_kernel(const PlanBatch pb, const ScanSegs sg) {
  __shared__ int wsum[16];
  int jd = 0;
  while (jd + 1 < 2 * pb.n && (int)blockIdx.x >= sg.start[jd + 1]) ++jd;
  const PlanJob& J = pb.j[jd >> 1];
  const int dir = jd & 1;
  const int n = dir ? J.n_src : J.n_dst;
  const int* cnt = dir ? J.cnt_out : J.cnt_in;
  int* ptr = dir ? J.t_rowptr : J.rowptr;
  const int base = ((int)blockIdx.x - sg.start[jd]) * SCAN_SEG;
  int v = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = base + threadIdx.x + j * 1024;
    if (i < n) v += cnt[i];
  }
  const int t = block_sum_1024(v, wsum);
  if (threadIdx.x == 0) ptr[base] = t;
}

__global__ __launch_bounds__(1024) void plan_scan_tops_kernel(const PlanBatch pb) {
  const int j = blockIdx.x >> 1, dir = blockIdx.x & 1;
  const PlanJob& J = pb.j[j];
  const int n = dir ? J.n_src : J.n_dst;
  int* ptr = dir ? J.t_rowptr : J.rowptr;
  const int nseg = (n + SCAN_SEG - 1) / SCAN_SEG;
  __shared__ int wsum[16];
  __shared__ int carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int b0 = 0; b0 < nseg; b0 += 1024) {
    const int i = b0 + threadIdx.x;
    const int v = i < nseg ? ptr[(int64_t)i * SCAN_SEG] : 0;
    int x = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int y = __shfl_up(x, o);
      if (lane >= o) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    int woff = 0;
    for (int q = 0; q < w; ++q) woff += wsum[q];
    const int carry = carry_s;
    if (i < nseg) ptr[(int64_t)i * SCAN_SEG] = carry + woff + x - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + woff + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) ptr[n] = carry_s;
}

__global__ __launch_bounds__(1024) void plan_scan_final_kernel(const PlanBatch pb, const ScanSegs sg) {
  __shared__ int wsum[16];
  int jd = 0;
  while (jd + 1 < 2 * pb.n && (int)blockIdx.x >= sg.start[jd + 1]) ++jd;
  const PlanJob& J = pb.j[jd >> 1];
  const int dir = jd & 1;
  const int n = dir ? J.n_src : J.n_dst;
  const int* cnt = dir ? J.cnt_out : J.cnt_in;
  int* ptr = dir ? J.t_rowptr : J.rowptr;
  const int base = ((int)blockIdx.x - sg.start[jd]) * SCAN_SEG;
  const int segbase = ptr[base];  // parked by the tops kernel; row `base` itself gets exactly this value back
  const int r0 = base + 4 * (int)threadIdx.x;
  int c[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) c[j] = (r0 + j < n) ? cnt[r0 + j] : 0;
  const int s4 = c[0] + c[1] + c[2] + c[3];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int x = s4;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int y = __shfl_up(x, o);
    if (lane >= o) x += y;
  }
  if (lane == 63) wsum[w] = x;
  __syncthreads();  // also orders every thread's read of ptr[base] before thread 0 rewrites it
  int woff = 0;
  for (int q = 0; q < w; ++q) woff += wsum[q];
  int run = segbase + woff + x - s4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (r0 + j < n) ptr[r0 + j] = run;
    run += c[j];
  }
}

__global__ __launch_bounds__(256) void plan_fill_kernel(const PlanBatch pb) {
  const int64_t total = pb.edge_start[pb.n];
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int j = find_job(pb.edge_start, pb.n, g);
    const PlanJob& J = pb.j[j];
    const int64_t e = g - pb.edge_start[j];
    const int64_t s = J.ei[e], d = J.ei[J.E + e];
    if (s < 0 || s >= J.n_src || d < 0 || d >= J.n_dst) continue;
    // a list ordered by destination (source) with no invalid edge IS its CSR (CSC): position = edge id, written in place, coalesced
    if (pb.flags_all[2 * j] != pb.build_id) {
      J.col[e] = (int)s;
      J.eid[e] = (int)e;
    } else {
      const int pi = J.rowptr[d] + J.pos_of_eid[e];   // arrival slots parked by the histogram pass
      J.tmp_in[pi] = (int)e;
      J.tmpc_in[pi] = (int)s;  // the other endpoint travels with the edge id: no random re-read of ei in the rank pass
    }
    if (pb.flags_all[2 * j + 1] != pb.build_id) {
      J.t_col[e] = (int)d;
      J.t_eid[e] = (int)e;
    } else {
      const int po = J.t_rowptr[s] + J.t_eid[e];
      J.tmp_out[po] = (int)e;
      J.tmpc_out[po] = (int)d;
    }
  }
}

// 16 lanes per row of (job, dir); rank-by-counting inside the row restores ascending edge order (the arrival order of the
// atomics is not reproducible).  Rows of <= 16 entries -- the scene-graph regime -- compare through lane shuffles, longer rows
// re-read the row from L1.  (One full wavefront per row left 3/4 of the lanes idle at in-degree 16: 2.3 ms for config 5.)
__global__ __launch_bounds__(256) void plan_rank_kernel(const PlanBatch pb) {
  const int64_t total_rows = pb.row_start[2 * pb.n];
  const int lane = threadIdx.x & 15;
  const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int64_t n_grps = ((int64_t)gridDim.x * blockDim.x) >> 4;
  // trip count is uniform over the wavefront (groups past the end idle), so the shuffles below see converged lanes
  const int64_t trips = (total_rows + n_grps - 1) / n_grps;
  for (int64_t it = 0; it < trips; ++it) {
    const int64_t r = grp + it * n_grps;
    const bool live = r < total_rows;
    int b = 0, deg = 0, dir = 0;
    bool ordered = false;
    const PlanJob* Jp = &pb.j[0];
    if (live) {
      const int jd = find_job(pb.row_start, 2 * pb.n, r);
      ordered = pb.flags_all[jd] != pb.build_id;
      Jp = &pb.j[jd >> 1];
      dir = jd & 1;
      const int row = (int)(r - pb.row_start[jd]);
      const int* ptr = dir ? Jp->t_rowptr : Jp->rowptr;
      b = ptr[row];
      deg = ptr[row + 1] - b;
      if (lane == 0) {
        // this group is the last reader of the row's counters: leave them zero for the next build
        if (dir) { Jp->cnt_out[row] = 0; Jp->cur_out[row] = 0; }
        else { Jp->cnt_in[row] = 0; Jp->cur_in[row] = 0; if (Jp->degf) Jp->degf[row] = 1.f / (float)(deg > 1 ? deg : 1); }
      }
    }
    const PlanJob& J = *Jp;
    if (ordered) {
      // ordered list: the fill pass wrote this direction in place (pos_of_eid[e] = e for the link pass)
      if (dir == 0 && pb.need_tpos)
        for (int c = lane; c < deg; c += 16) J.pos_of_eid[b + c] = b + c;
      deg = 0;
    }
    const int* tmp = dir ? J.tmp_out : J.tmp_in;
    const int* tmpc = dir ? J.tmpc_out : J.tmpc_in;
    // wave-uniform choice: every row of this wavefront's 4 groups is short
    const bool all_short = __all(deg <= 16);
    if (all_short) {
      const int mine = lane < deg ? tmp[b + lane] : 0x7fffffff;
      const int other = lane < deg ? tmpc[b + lane] : 0;
      int rank = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) rank += (__shfl(mine, i, 16) < mine) ? 1 : 0;
      if (lane < deg) {
        const int pos = b + rank;
        if (dir == 0) {
          J.eid[pos] = mine;
          J.col[pos] = other;  // source endpoint
          if (pb.need_tpos) J.pos_of_eid[mine] = pos;  // 4-byte scatter over all edges: only the link pass (t_pos) reads it
        } else {
          J.t_eid[pos] = mine;
          J.t_col[pos] = other;  // destination endpoint
        }
      }
    } else if (__all(deg <= 32)) {
      // rows of 17..32 entries (the out-degrees of a 10^6-object graph scatter around its mean of 16: nearly every wavefront holds
      // one): two entries per lane, still ranked through shuffles -- the loop below re-reads the row once per entry
      const int m0 = lane < deg ? tmp[b + lane] : 0x7fffffff, m1 = lane + 16 < deg ? tmp[b + lane + 16] : 0x7fffffff;
      const int o0 = lane < deg ? tmpc[b + lane] : 0, o1 = lane + 16 < deg ? tmpc[b + lane + 16] : 0;
      int r0 = 0, r1 = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int v0 = __shfl(m0, i, 16), v1 = __shfl(m1, i, 16);
        r0 += ((v0 < m0) ? 1 : 0) + ((v1 < m0) ? 1 : 0);
        r1 += ((v0 < m1) ? 1 : 0) + ((v1 < m1) ? 1 : 0);
      }
      if (lane < deg) {
        const int pos = b + r0;
        if (dir == 0) {
          J.eid[pos] = m0;
          J.col[pos] = o0;
          if (pb.need_tpos) J.pos_of_eid[m0] = pos;
        } else {
          J.t_eid[pos] = m0;
          J.t_col[pos] = o0;
        }
      }
      if (lane + 16 < deg) {
        const int pos = b + r1;
        if (dir == 0) {
          J.eid[pos] = m1;
          J.col[pos] = o1;
          if (pb.need_tpos) J.pos_of_eid[m1] = pos;
        } else {
          J.t_eid[pos] = m1;
          J.t_col[pos] = o1;
        }
      }
    } else {
      for (int c = lane; c < deg; c += 16) {
        const int mine = tmp[b + c];
        int rank = 0;
        for (int i = 0; i < deg; ++i) rank += (tmp[b + i] < mine) ? 1 : 0;
        const int pos = b + rank;
        if (dir == 0) {
          J.eid[pos] = mine;
          J.col[pos] = tmpc[b + c];
          if (pb.need_tpos) J.pos_of_eid[mine] = pos;  // 4-byte scatter over all edges: only the link pass (t_pos) reads it
        } else {
          J.t_eid[pos] = mine;
          J.t_col[pos] = tmpc[b + c];
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void plan_link_kernel(const PlanBatch pb) {
  const int64_t total = pb.edge_start[pb.n];
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    const int j = find_job(pb.edge_start, pb.n, g);
    const PlanJob& J = pb.j[j];
    const int64_t k = g - pb.edge_start[j];
    if (k < J.t_rowptr[J.n_src]) J.t_pos[k] = J.pos_of_eid[J.t_eid[k]];
  }
}


template <bool RC>
__global__ __launch_bounds__(1024) void plan_small_kernel(const PlanSmallArgs a, int* status) {
  __shared__ __attribute__((aligned(16))) char lds[PS_LDS_BYTES];
  KT(0);
  plan_small_block<RC>(a, status, (int)blockIdx.x, lds);
  KT(5);
}

size_t plan_scratch_ints(int64_t E, int n_src, int n_dst) {
  // cnt_in, cur_in [n_dst]; cnt_out, cur_out [n_src]; flags [64]; tmp_in, tmp_out, t_eid, pos_of_eid, tmpc_in, tmpc_out [E]; each 64-int aligned
  auto a = [](int64_t x) { return (size_t)((x + 63) & ~(int64_t)63); };
  return 2 * a(n_dst) + 2 * a(n_src) + 64 + 6 * a(E);
}

void plan_carve(PlanJob& job, int* s) {
  auto a = [](int64_t x) { return (size_t)((x + 63) & ~(int64_t)63); };
  job.cnt_in = s;  s += a(job.n_dst);
  job.cur_in = s;  s += a(job.n_dst);
  job.cnt_out = s; s += a(job.n_src);
  job.cur_out = s; s += a(job.n_src);
  job.flags = s;   s += 64;
  job.tmp_in = s;  s += a(job.E);
  job.tmp_out = s; s += a(job.E);
  job.t_eid = s;   s += a(job.E);
  job.pos_of_eid = s; s += a(job.E);
  job.tmpc_in = s; s += a(job.E);
  job.tmpc_out = s;
}

// the link pass alone (t_pos: CSR position of every CSC entry), for a plan whose single-launch build ran as a role of another
// launch (front.hip) with need_tpos set
int plan_link_launch(PlanBatch& pb, hipStream_t st) {
  pb.edge_start[0] = 0;
  for (int j = 0; j < pb.n; ++j) pb.edge_start[j + 1] = pb.edge_start[j] + pb.j[j].E;
  const int64_t E = pb.edge_start[pb.n];
  if (E <= 0) return HMP_OK;
  const int lg = (int)(cdiv(E, 256) < 2048 ? cdiv(E, 256) : 2048);
  hipLaunchKernelGGL(plan_link_kernel, dim3(lg), dim3(256), 0, st, pb);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

int plan_launch(PlanBatch& pb, int* d_status, hipStream_t st) {
  HMP_CHECK_ARG(pb.n >= 0 && pb.n <= HMP_MAX_EDGE_TYPES, "plan: %d jobs", pb.n);
  if (pb.n == 0) return HMP_OK;
  pb.edge_start[0] = 0;
  pb.row_start[0] = 0;
  for (int j = 0; j < pb.n; ++j) {
    PlanJob& J = pb.j[j];
    HMP_CHECK_ARG(J.E >= 0 && J.E < (int64_t)2147483647 && J.n_src >= 0 && J.n_dst >= 0, "plan: bad sizes");
    HMP_CHECK_ARG(J.E == 0 || J.ei != nullptr, "plan: null edge_index with E > 0");
    pb.edge_start[j + 1] = pb.edge_start[j] + J.E;
    pb.row_start[2 * j + 1] = pb.row_start[2 * j] + J.n_dst;
    pb.row_start[2 * j + 2] = pb.row_start[2 * j + 1] + J.n_src;
    // counters + cursors of one job are one contiguous block (see plan_carve); the rank kernel re-zeroes them
    const size_t zero_ints = (size_t)(J.tmp_in - J.cnt_in);
    if (pb.clear_first && zero_ints > 0) HMP_HIP(hipMemsetAsync(J.cnt_in, 0, zero_ints * sizeof(int), st));
  }
  const int64_t E = pb.edge_start[pb.n];
  const int64_t rows = pb.row_start[2 * pb.n];
  const char* sm = getenv("HMP_PLAN_SMALL");  // 0 / 1 pins the multi-launch / single-launch build (tests); unset: by size
  const int small_mode = sm ? (sm[0] == '1' ? 1 : 0) : -1;
  pb.built_small = 0;
  if (small_mode == 1 || (small_mode < 0 && E <= PS_MAX_EDGES)) {
    pb.built_small = 1;
    PlanSmallArgs a;
    a.pb = pb;
    const int blocks = plan_small_layout(pb, a.part_start, a.rows_per_part);
    if (plan_small_fits_rc(pb)) hipLaunchKernelGGL(plan_small_kernel<true>, dim3(blocks), dim3(1024), 0, st, a, d_status);
    else hipLaunchKernelGGL(plan_small_kernel<false>, dim3(blocks), dim3(1024), 0, st, a, d_status);
    HMP_LAUNCH_CHECK();
    if (pb.need_tpos && E > 0) {
      const int lg = (int)(cdiv(E, 256) < 2048 ? cdiv(E, 256) : 2048);
      hipLaunchKernelGGL(plan_link_kernel, dim3(lg), dim3(256), 0, st, pb);
      HMP_LAUNCH_CHECK();
    }
    return HMP_OK;
  }
  const int eg = (int)(E > 0 ? (cdiv(E, 256) < 2048 ? cdiv(E, 256) : 2048) : 1);
  {
    static std::atomic<int> build_counter{0};
    int id = ++build_counter;
    if (id == 0) id = ++build_counter;  // 0 = the zeroed workspace: never a stamp
    pb.build_id = id;
    pb.flags_all = pb.j[0].flags;
  }
  if (E > 0) {
    hipLaunchKernelGGL(plan_hist_kernel, dim3(eg), dim3(256), 0, st, pb, d_status);
    HMP_LAUNCH_CHECK();
  }
  {
    ScanSegs sg;
    int blocks = 0;
    for (int jd = 0; jd < 2 * pb.n; ++jd) {
      const int nrows = (jd & 1) ? pb.j[jd >> 1].n_src : pb.j[jd >> 1].n_dst;
      sg.start[jd] = blocks;
      blocks += cdiv(nrows, SCAN_SEG);
    }
    sg.start[2 * pb.n] = blocks;
    if (blocks > 0) hipLaunchKernelGGL(plan_scan_partial_kernel, dim3(blocks), dim3(1024), 0, st, pb, sg);
    hipLaunchKernelGGL(plan_scan_tops_kernel, dim3(2 * pb.n), dim3(1024), 0, st, pb);
    if (blocks > 0) hipLaunchKernelGGL(plan_scan_final_kernel, dim3(blocks), dim3(1024), 0, st, pb, sg);
    HMP_LAUNCH_CHECK();
  }
  if (E > 0) {
    hipLaunchKernelGGL(plan_fill_kernel, dim3(eg), dim3(256), 0, st, pb);
    HMP_LAUNCH_CHECK();
    const int64_t want = cdiv(rows, 16);  // 16 rows per block and trip
    const int rg = (int)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
    hipLaunchKernelGGL(plan_rank_kernel, dim3(rg), dim3(256), 0, st, pb);
    HMP_LAUNCH_CHECK();
    if (pb.need_tpos) {
      hipLaunchKernelGGL(plan_link_kernel, dim3(eg), dim3(256), 0, st, pb);
      HMP_LAUNCH_CHECK();
    }
  }
  return HMP_OK;
}

}  // namespace hmp

extern "C" size_t hmp_plan_scratch_bytes(int64_t n_edges, int32_t n_src, int32_t n_dst) {
  return hmp::plan_scratch_ints(n_edges, n_src, n_dst) * sizeof(int);
}

extern "C" int hmp_plan_build(const int64_t* d_edge_index, hmp_plan plan, void* d_scratch, int32_t* d_status, void* stream) {
  using namespace hmp;
  HMP_CHECK_ARG(plan.d_rowptr && plan.d_t_rowptr, "hmp_plan_build: null rowptr");
  HMP_CHECK_ARG(plan.n_edges == 0 || (plan.d_col && plan.d_eid && plan.d_t_col && plan.d_t_pos && d_scratch),
                "hmp_plan_build: null output/scratch with E > 0");
  PlanBatch pb;
  memset(&pb, 0, sizeof(pb));
  pb.n = 1;
  pb.need_tpos = 1;
  pb.clear_first = 1;
  PlanJob& J = pb.j[0];
  J.ei = d_edge_index;
  J.E = plan.n_edges;
  J.n_src = plan.n_src;
  J.n_dst = plan.n_dst;
  J.rowptr = plan.d_rowptr; J.col = plan.d_col; J.eid = plan.d_eid;
  J.t_rowptr = plan.d_t_rowptr; J.t_col = plan.d_t_col; J.t_pos = plan.d_t_pos;
  HMP_CHECK_ARG(d_scratch != nullptr, "hmp_plan_build: scratch required");
  plan_carve(J, (int*)d_scratch);
  return plan_launch(pb, d_status, (hipStream_t)stream);
}
