// Object connectivity of a room-object scene graph: the pairwise geometric predicates of the
// reference's add_object_connectivity (src/hydra_gnn/preprocess_dsgs.py:89-225), which the inference server evaluates for every
// incoming frame (bin/room_classification_server:253-257) in an O(n^2)-per-room Python loop over numpy 3-vectors.
//
// Objects arrive in the reference's visiting order (ascending node id).  Object i is compared with every EARLIER object j of
// the same room; an edge (i, j) exists when i is on / under / near j (the reference's `is_above` is commented out there).  All
// arithmetic is the reference's, in float64 like numpy, without contraction, so the edge SET is bit-exact
// (tests/golden/dsg_x8F5xyUWy9e_expected.npz was produced by the reference's own code).  Output order: by i, then by j ascending
// = the reference's insertion order.
//
// Work decomposition: one block per object i, lanes stride the candidates j < i; pass 1 counts, a single-block exclusive scan
// turns counts into offsets, pass 2 re-evaluates and compacts in order (wave ballots + a running base).  A frame has 10^1..10^3
// objects: both passes are launch-latency bound, which is the point (the host loop takes milliseconds).
#include "common.h"

namespace hmp {

struct ObjGeom {
  const double* pos;   // [n][3]
  const double* size;  // [n][3]  bounding_box.max - bounding_box.min
  const int* room;     // [n]     room index, < 0: no room
  int n;
  double threshold_near, max_near, max_on;
};

#pragma clang fp contract(off)
__device__ __forceinline__ bool obj_edge(const ObjGeom& g, int i, int j) {
  if (g.room[i] < 0 || g.room[i] != g.room[j]) return false;
  double p1[3], p2[3], s1[3], s2[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    p1[k] = g.pos[3 * i + k]; p2[k] = g.pos[3 * j + k];
    s1[k] = g.size[3 * i + k]; s2[k] = g.size[3 * j + k];
  }
  const double dx = fabs(p1[0] - p2[0]), dy = fabs(p1[1] - p2[1]), dz = fabs(p1[2] - p2[2]);
  const bool in2 = dx <= s2[0] / 2 && dy <= s2[1] / 2;  // centre of 1 inside 2 on the xy plane
  const bool in1 = dx <= s1[0] / 2 && dy <= s1[1] / 2;
  // _is_on (:89-110)
  const bool above = p1[2] > p2[2];
  const double on_thresh = g.max_on + (s1[2] + s2[2]) / 2;
  const bool is_on = (in2 && above && dz <= on_thresh) || (in1 && !above && dz <= on_thresh);
  // _is_under (:139-158)
  const bool is_under = (in1 || in2) && (p1[2] < p2[2] || p2[2] < p1[2]);
  // _is_near (:161-180)
  bool is_near = true;
  const double d[3] = {dx, dy, dz};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double avg = (s1[k] + s2[k]) / 2.0;
    is_near = is_near && d[k] <= avg * g.threshold_near && d[k] - avg <= g.max_near;
  }
  return is_on || is_under || is_near;
}

__global__ __launch_bounds__(256) void object_edge_count_kernel(const ObjGeom g, int* __restrict__ count) {
  __shared__ int ws[4];
  const int i = blockIdx.x;
  int c = 0;
  for (int j = threadIdx.x; j < i; j += 256) c += obj_edge(g, i, j) ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) count[i] = ws[0] + ws[1] + ws[2] + ws[3];
}

// offset[0..n] = exclusive scan of count[0..n); one block (n is a frame's object count)
__global__ __launch_bounds__(1024) void scan_i32_kernel(const int* __restrict__ count, int n, int* __restrict__ offset) {
  __shared__ int part[1024];
  const int per = (n + 1023) / 1024;
  const int b = threadIdx.x * per, e = min(n, b + per);
  int s = 0;
  for (int k = b; k < e; ++k) s += count[k];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int t = 0; t < 1024; ++t) { const int v = part[t]; part[t] = run; run += v; }
    offset[n] = run;
  }
  __syncthreads();
  int run = part[threadIdx.x];
  for (int k = b; k < e; ++k) { offset[k] = run; run += count[k]; }
}

__global__ __launch_bounds__(256) void object_edge_fill_kernel(const ObjGeom g, const int* __restrict__ offset, int* __restrict__ edges,
                                                               int total) {
  __shared__ int ws[4];
  const int i = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int base = offset[i];
  for (int j0 = 0; j0 < i; j0 += 256) {  // uniform trip count: every thread reaches the barriers
    const int j = j0 + threadIdx.x;
    const bool f = j < i && obj_edge(g, i, j);
    const unsigned long long m = __ballot(f);
    if (lane == 0) ws[w] = __popcll(m);
    __syncthreads();
    int before = __popcll(m & ((1ull << lane) - 1ull));
    for (int q = 0; q < w; ++q) before += ws[q];
    const int chunk = ws[0] + ws[1] + ws[2] + ws[3];
    if (f) {
      const int o = base + before;
      if (o < total) { edges[o] = i; edges[total + o] = j; }
    }
    base += chunk;
    __syncthreads();
  }
}

}  // namespace hmp

using namespace hmp;

static int geom_check(const double* d_pos, const double* d_size, const int32_t* d_room, int32_t n) {
  HMP_CHECK_ARG(n >= 0 && (n == 0 || (d_pos && d_size && d_room)), "hmp_object_edges: bad argument");
  return HMP_OK;
}

extern "C" int hmp_object_edges_count(const double* d_pos, const double* d_size, const int32_t* d_room, int32_t n, double threshold_near,
                                      double max_near, double max_on, int32_t* d_count, int32_t* d_offset, void* stream) {
  HMP_TRY(geom_check(d_pos, d_size, d_room, n));
  HMP_CHECK_ARG(d_offset && (n == 0 || d_count), "hmp_object_edges_count: null output");
  hipStream_t st = (hipStream_t)stream;
  ObjGeom g{d_pos, d_size, d_room, n, threshold_near, max_near, max_on};
  if (n > 0) hipLaunchKernelGGL(object_edge_count_kernel, dim3(n), dim3(256), 0, st, g, d_count);
  hipLaunchKernelGGL(scan_i32_kernel, dim3(1), dim3(1024), 0, st, d_count, n, d_offset);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

extern "C" int hmp_object_edges_fill(const double* d_pos, const double* d_size, const int32_t* d_room, int32_t n, double threshold_near,
                                     double max_near, double max_on, const int32_t* d_offset, int32_t* d_edges, int32_t total, void* stream) {
  HMP_TRY(geom_check(d_pos, d_size, d_room, n));
  HMP_CHECK_ARG(d_offset && total >= 0 && (total == 0 || d_edges), "hmp_object_edges_fill: bad argument");
  if (n == 0 || total == 0) return HMP_OK;
  ObjGeom g{d_pos, d_size, d_room, n, threshold_near, max_near, max_on};
  hipLaunchKernelGGL(object_edge_fill_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, g, d_offset, d_edges, total);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}
