// K3 -- GAT edge softmax + weighted aggregation (forward and the two backward passes).
//
// Replaces PyG GATConv.edge_update / softmax / message / aggregate: per
// destination row i and head h:  e_k = leaky_relu(a_s[j_k,h] + a_d[i,h] + <edge_attr_k, v_edge[:,h]>, 0.2),
// alpha = softmax_k(e) (max-subtracted, denominator + 1e-16), alpha' = dropout(alpha),
// out[i,h,:] = sum_k alpha'_k h_s[j_k,h,:];  then concat or head-mean, + bias, HeteroConv sum over the edge types
// reaching the node type, ELU, feature dropout -- all in ONE kernel.
//
// Mapping: a row group of GS lanes per destination row; lane g owns channels 4g..4g+3 of EVERY head (heads are
// padded to Cp = align4(C) floats in the projected rows), so head-mean needs no cross-lane traffic and every
// neighbour costs H coalesced 16-byte-per-lane loads.  The softmax is ONLINE (running max / running sum with
// rescaling of the accumulator), so a row is swept once and nothing per-edge is stored in the forward pass;
// the backward recomputes alpha from the saved per-row (max, denominator).  Logits are computed redundantly by
// every lane of the group (H exps per neighbour per lane: cheap next to the gather).
//
// Self loops (add_self_loops=True for same-type edges): entries with col == row are skipped ("remove_self_loops")
// and one loop i->i is appended after the row's edges for i < min(N_src, N_dst); its edge-attribute term is 0
// (the reference's fill_value for GAT_edge is zeros(3): heterogeneous_network.py:77-78).
//
// No atomics: backward pass 1 runs destination-major (d logits, d a_dst), pass 2 source-major over the CSC lists
// (d h_s, d a_src) reading what pass 1 stored per edge.
#include "kernels.h"

namespace hmp {

KT_DEFINE(gat)

namespace {

constexpr float NEG_SLOPE = 0.2f;

// The descriptors live in a device-memory table, so the pointers they hold are GENERIC to the compiler: it would emit
// flat_load (counted on vmcnt AND lgkmcnt, i.e. every wait drains everything).  glob() tells it they are global.
typedef float f4v __attribute__((ext_vector_type(4)));
template <typename T>
using gptr = const T __attribute__((address_space(1)))*;
template <typename T>
__device__ __forceinline__ gptr<T> glob(const T* p) { return (gptr<T>)p; }
template <typename T>
using gwptr = T __attribute__((address_space(1)))*;
template <typename T>
__device__ __forceinline__ gwptr<T> globw(T* p) { return (gwptr<T>)p; }
__device__ __forceinline__ void st4(float* p, const float4& v) {
  f4v t;
  t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
  *(gwptr<f4v>)p = t;
}
__device__ __forceinline__ float4 ld4(const float* p) {
  const f4v v = *(gptr<f4v>)p;
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void fma4(float4& a, float w, const float4& v) {
  a.x += w * v.x; a.y += w * v.y; a.z += w * v.z; a.w += w * v.w;
}
__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

template <int GS>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = GS / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ DropCfg make_cfg(const GatDyn& dyn, float p, uint32_t stream) {
  DropCfg c;
  c.k0 = dyn.k0; c.k1 = dyn.k1; c.step = dyn.step; c.stream = stream;
  c.thresh = drop_thresh(p); c.scale = 1.f / (1.f - p);
  c.step_dev = dyn.step_dev;
  return drop_resolve(c);
}

// keep flags of the HM heads of edge position `pos` (element (pos, h) of an [*, GAT_HMAX] tensor)
template <int HM>
__device__ __forceinline__ void alpha_keep(const DropCfg& cfg, int64_t pos, bool (&keep)[HM]) {
  bool k4[4];
  drop_keep4(cfg, (uint32_t)(pos * 2), k4);
#pragma unroll
  for (int h = 0; h < HM && h < 4; ++h) keep[h] = k4[h];
  if (HM > 4) {
    drop_keep4(cfg, (uint32_t)(pos * 2 + 1), k4);
#pragma unroll
    for (int h = 4; h < HM; ++h) keep[h] = k4[h - 4];
  }
}

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == HMP_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == HMP_ACT_ELU) return v > 0.f ? v : expm1f(v);
  return v;
}

// raw (pre-leaky) logits of one neighbour for all heads.  `ea` is uniform (null: the conv has no edge attributes); `use`
// masks the edge term of this slot (virtual self loops carry zero attributes, dead batch slots nothing) WITHOUT a branch
// around the loads: eid is clamped to a valid edge by the caller.
template <int HM>
__device__ __forceinline__ void edge_logits(const GatInS& I, const float* ea, int H, const float* aj, const float (&ad)[HM],
                                            const float (&ve)[GAT_MAX_EDIM][HM], bool use, int eid, float (&raw)[HM]) {
  // every load is unconditional (head / attribute index clamped into range, the value masked afterwards): a load under a
  // run-time `h < H` is a branch, and the compiler then waits for each load before it issues the next one
  float x[GAT_MAX_EDIM] = {0.f, 0.f, 0.f, 0.f};
  if (ea) {  // uniform
#pragma unroll
    for (int d = 0; d < GAT_MAX_EDIM; ++d) {
      const float t = glob(ea)[(int64_t)eid * I.edim + min(d, I.edim - 1)];
      x[d] = (use && d < I.edim) ? t : 0.f;
    }
  }
  float as[HM];
#pragma unroll
  for (int h = 0; h < HM; ++h) as[h] = glob(aj)[I.asoff + min(h, H - 1)];
#pragma unroll
  for (int h = 0; h < HM; ++h) {
    float r = as[h] + ad[h];
    if (ea) {
#pragma unroll
      for (int d = 0; d < GAT_MAX_EDIM; ++d) r += x[d] * ve[d][h];
    }
    raw[h] = h < H ? r : 0.f;
  }
}

template <int HM>
__device__ __forceinline__ void load_row_consts(const GatInS& I, int H, int row, float (&ad)[HM], float (&ve)[GAT_MAX_EDIM][HM]) {
#pragma unroll
  for (int h = 0; h < HM; ++h) {
    const float t = glob(I.zd)[(int64_t)row * I.ldzd + I.adoff + min(h, H - 1)];
    ad[h] = (h < H) ? t : 0.f;
  }
  if (I.vedge && I.edim > 0) {  // uniform
#pragma unroll
    for (int d = 0; d < GAT_MAX_EDIM; ++d)
#pragma unroll
      for (int h = 0; h < HM; ++h) {
        const float t = glob(I.vedge)[min(d, I.edim - 1) * GAT_HMAX + min(h, H - 1)];
        ve[d][h] = (d < I.edim && h < H) ? t : 0.f;
      }
  } else {
#pragma unroll
    for (int d = 0; d < GAT_MAX_EDIM; ++d)
#pragma unroll
      for (int h = 0; h < HM; ++h) ve[d][h] = 0.f;
  }
}

// A batch of UB consecutive neighbour slots k0 .. k0+UB-1 of one destination row (real edges [b, e), then the virtual self
// loop at k == e when kend > e): neighbour ids and edge ids first (one round trip), then everything that depends on them
// -- attention logits (a_s of the neighbour, edge attributes) and the neighbour's H head slices of this lane -- in flight
// together (second round trip).  The per-edge arithmetic that follows is the sequential code's, edge by edge, in the same
// order: results are bit-identical to the unbatched kernels, the dependent chain is 2 round trips per UB edges instead of
// 2 per edge.  Slots outside the row are clamped to valid addresses and flagged dead.
template <int HM, int UB>
struct EdgeBatch {
  int j[UB], eid[UB];
  bool live[UB], loop[UB], removed[UB];  // removed: an explicit self loop that add_self_loops replaces (PyG remove_self_loops)
  float raw[UB][HM];
  float4 v[UB][HM];
};

template <int HM, int UB>
__device__ __forceinline__ void fetch_batch(const GatInS& I, int Cp, const float* __restrict__ ea, int H, int row, int b, int e, int kend,
                                            int k0, int cc, const float (&ad)[HM], const float (&ve)[GAT_MAX_EDIM][HM],
                                            EdgeBatch<HM, UB>& B) {
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    const int k = k0 + u;
    const bool in = k < kend;
    const bool lp = in && k >= e;
    const int kc = (e > b) ? min(k, e - 1) : 0;
    const int cj = glob(I.col)[kc];
    const int ce = ea ? glob(I.eid)[kc] : 0;
    const bool real = in && !lp;
    B.loop[u] = lp;
    B.j[u] = lp ? row : (real ? cj : 0);  // dead slots read row 0 (always allocated), never used
    B.eid[u] = real ? ce : 0;
    B.removed[u] = real && I.self_loops && cj == row;
    B.live[u] = in && !B.removed[u];
  }
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    edge_logits<HM>(I, ea, H, I.za + (int64_t)B.j[u] * I.ldza, ad, ve, !B.loop[u] && B.live[u], B.eid[u], B.raw[u]);
    const float* zj = I.z + (int64_t)B.j[u] * I.ldz + I.hoff + cc;
#pragma unroll
    for (int h = 0; h < HM; ++h) B.v[u][h] = ld4(zj + min(h, H - 1) * Cp);  // heads >= H: a duplicate nobody reads
  }
}

// =================================================================================================
// forward
// =================================================================================================
template <int HM, int GS>
__global__ __launch_bounds__(256) void gat_fwd_kernel(const GatLayerS* __restrict__ tab, const GatDyn dyn) {
  int di = 0;
  while (di + 1 < tab->n_dst && (int)blockIdx.x >= dyn.block_start[di + 1]) ++di;
  const GatDstS& D = tab->d[di];
  const int rpb = 256 / GS;
  const int row = ((int)blockIdx.x - dyn.block_start[di]) * rpb + (int)threadIdx.x / GS;
  if (row >= dyn.n_nodes[D.t]) return;
  const int gl = threadIdx.x % GS, c0 = gl * 4;
  const bool cact = c0 < D.Cp;
  const int H = D.H;

  float4 tot[HM];
#pragma unroll
  for (int h = 0; h < HM; ++h) tot[h] = make_float4(0.f, 0.f, 0.f, 0.f);

  for (int ii = 0; ii < D.n_in; ++ii) {
    const GatInS& I = D.in[ii];
    const int64_t E = dyn.n_edges[I.et];
    const int n_loop = I.self_loops ? min(dyn.n_nodes[I.src_t], dyn.n_nodes[D.t]) : 0;
    const float* ea = (I.edim > 0 && E > 0) ? dyn.edge_attr[I.et] : nullptr;  // dead batch slots read attribute row 0
    float ad[HM], ve[GAT_MAX_EDIM][HM];
    load_row_consts<HM>(I, H, row, ad, ve);
    const bool adrop = dyn.training && I.adrop_p > 0.f;
    DropCfg acfg;
    if (adrop) acfg = make_cfg(dyn, I.adrop_p, I.adrop_stream);

    float m[HM], s[HM];
    float4 acc[HM];
#pragma unroll
    for (int h = 0; h < HM; ++h) { m[h] = -INFINITY; s[h] = 0.f; acc[h] = make_float4(0.f, 0.f, 0.f, 0.f); }

    const int b = glob(I.rowptr)[row], e = glob(I.rowptr)[row + 1];
    const int kend = e + ((row < n_loop) ? 1 : 0);
    constexpr int UB = (HM <= 4) ? 4 : 2;
    const int cc = cact ? c0 : 0;
    for (int k0 = b; k0 < kend; k0 += UB) {
      EdgeBatch<HM, UB> B;
      fetch_batch<HM, UB>(I, D.Cp, ea, H, row, b, e, kend, k0, cc, ad, ve, B);
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        if (!B.live[u]) continue;
        bool keep[HM];
#pragma unroll
        for (int h = 0; h < HM; ++h) keep[h] = true;
        if (adrop) alpha_keep<HM>(acfg, B.loop[u] ? (E + row) : (int64_t)(k0 + u), keep);
#pragma unroll
        for (int h = 0; h < HM; ++h) {
          if (h >= H) continue;
          const float ev = B.raw[u][h] > 0.f ? B.raw[u][h] : NEG_SLOPE * B.raw[u][h];
          const float mn = fmaxf(m[h], ev);
          const float sc = expf(m[h] - mn);
          const float p = expf(ev - mn);
          s[h] = s[h] * sc + p;
          float w = p;
          if (adrop) w = keep[h] ? p * acfg.scale : 0.f;
          if (cact) {
            const float4 v = B.v[u][h];
            acc[h].x = acc[h].x * sc + w * v.x;
            acc[h].y = acc[h].y * sc + w * v.y;
            acc[h].z = acc[h].z * sc + w * v.z;
            acc[h].w = acc[h].w * sc + w * v.w;
          }
          m[h] = mn;
        }
      }
    }
#pragma unroll
    for (int h = 0; h < HM; ++h) {
      if (h >= H) continue;
      const float den = s[h] + 1e-16f;
      tot[h].x += acc[h].x / den; tot[h].y += acc[h].y / den; tot[h].z += acc[h].z / den; tot[h].w += acc[h].w / den;
      if (gl == 0) {
        globw(I.smax)[(int64_t)row * GAT_HMAX + h] = m[h];
        globw(I.sden)[(int64_t)row * GAT_HMAX + h] = den;
      }
    }
  }
  if (!cact) return;

  const bool fdrop = dyn.training && D.drop_p > 0.f;
  DropCfg fcfg;
  if (fdrop) fcfg = make_cfg(dyn, D.drop_p, D.drop_stream);
  auto finish = [&](float v, int col) -> float {
    v = (v + (D.bias ? glob(D.bias)[col] : 0.f)) * D.group_scale;
    v = act_apply(v, D.act);
    if (fdrop) {
      bool k4[4];
      drop_keep4(fcfg, (uint32_t)row * (uint32_t)(D.ldo >> 2) + (uint32_t)(col >> 2), k4);
      v = k4[col & 3] ? (v * fcfg.scale + 0.0f) : -0.0f;  // dropped: -0.0f (sign bit = "dropped", see gemm.hip epilogue)
    }
    return v;
  };
  auto orow = globw(D.out) + (int64_t)row * D.ldo;
  if (D.concat) {
#pragma unroll
    for (int h = 0; h < HM; ++h) {
      if (h >= H) continue;
      const float t4[4] = {tot[h].x, tot[h].y, tot[h].z, tot[h].w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = c0 + i;
        if (c < D.C) orow[h * D.C + c] = finish(t4[i], h * D.C + c);
      }
    }
  } else {
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int h = 0; h < HM; ++h)
      if (h < H) { sum.x += tot[h].x; sum.y += tot[h].y; sum.z += tot[h].z; sum.w += tot[h].w; }
    const float inv = 1.f / (float)H;
    const float t4[4] = {sum.x * inv, sum.y * inv, sum.z * inv, sum.w * inv};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + i;
      if (c < D.C) orow[c] = finish(t4[i], c);
    }
  }
}

// =================================================================================================
// backward pass 1: destination major.  For every incoming edge k of row i and head h:
//   d alpha'_k = <g[i,h,:], h_s[j_k,h,:]>, d alpha_k = d alpha'_k * keep/(1-p),
//   d e_k = alpha_k (d alpha_k - sum_k' alpha_k' d alpha_k'),  d raw_k = d e_k * leaky'(raw_k)
// writes alpha'_k and d raw_k per edge (CSR position; loops at E + i), d a_dst[i,h] = sum_k d raw_k.
// =================================================================================================
// gradient slice of this lane for every head.  Vector path (C % 4 == 0, 16-byte aligned rows -- the executor's buffers):
// one unconditional 16-byte load per head, head index clamped; otherwise element by element.
template <int HM>
__device__ __forceinline__ void load_g(const GatDstS& D, const GatDyn& dyn, int row, int c0, bool cact, float4 (&g)[HM]) {
  const float* gp = D.g ? D.g : dyn.g_top;
  const int ldg = D.g ? D.ldg : dyn.ld_gtop;
  const float* gr = gp + (int64_t)row * ldg;
  const bool vec = (D.C & 3) == 0 && (ldg & 3) == 0 && (reinterpret_cast<uintptr_t>(gp) & 15) == 0;  // uniform
  const float sc = D.concat ? D.group_scale : D.group_scale / (float)D.H;
  if (vec) {
    const int cc = cact ? c0 : 0;
#pragma unroll
    for (int h = 0; h < HM; ++h) {
      const float4 t = ld4(gr + (D.concat ? min(h, D.H - 1) * D.C : 0) + cc);
      const bool on = cact && h < D.H;
      g[h] = D.concat ? make_float4(on ? t.x * sc : 0.f, on ? t.y * sc : 0.f, on ? t.z * sc : 0.f, on ? t.w * sc : 0.f)
                      : make_float4(on ? t.x / (float)D.H * D.group_scale : 0.f, on ? t.y / (float)D.H * D.group_scale : 0.f,
                                    on ? t.z / (float)D.H * D.group_scale : 0.f, on ? t.w / (float)D.H * D.group_scale : 0.f);
    }
    return;
  }
#pragma unroll
  for (int h = 0; h < HM; ++h) {
    float t4[4] = {0.f, 0.f, 0.f, 0.f};
    if (cact && h < D.H) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = c0 + i;
        if (c < D.C) t4[i] = D.concat ? glob(gr)[h * D.C + c] : glob(gr)[c] / (float)D.H;
      }
    }
    g[h] = make_float4(t4[0] * D.group_scale, t4[1] * D.group_scale, t4[2] * D.group_scale, t4[3] * D.group_scale);
  }
}

template <int HM, int GS>
__global__ __launch_bounds__(256) void gat_bwd1_kernel(const GatLayerS* __restrict__ tab, const GatDyn dyn) {
  constexpr int CAP = 32;  // edges per row whose sweep-0 scalars are parked in LDS (multiple of the batch size; 16 KB at 8 rows x 4 heads: rooms with 17..32 objects no longer take the recompute path)
  __shared__ float cache[256 / GS][CAP][HM][4];
  int di = 0;
  while (di + 1 < tab->n_dst && (int)blockIdx.x >= dyn.block_start[di + 1]) ++di;
  const GatDstS& D = tab->d[di];
  const int rpb = 256 / GS;
  const int row = ((int)blockIdx.x - dyn.block_start[di]) * rpb + (int)threadIdx.x / GS;
  if (row >= dyn.n_nodes[D.t]) return;  // whole groups leave together: shuffles below stay inside a group
  const int gl = threadIdx.x % GS, c0 = gl * 4;
  const bool cact = c0 < D.Cp;
  const int H = D.H;
  float4 g[HM];
  KT(0);
  load_g<HM>(D, dyn, row, c0, cact, g);

  for (int ii = 0; ii < D.n_in; ++ii) {
    KT(1 + 4 * ii);
    const GatInS& I = D.in[ii];
    const int64_t E = dyn.n_edges[I.et];
    const int n_loop = I.self_loops ? min(dyn.n_nodes[I.src_t], dyn.n_nodes[D.t]) : 0;
    const float* ea = (I.edim > 0 && E > 0) ? dyn.edge_attr[I.et] : nullptr;  // dead batch slots read attribute row 0
    float ad[HM], ve[GAT_MAX_EDIM][HM], m[HM], den[HM];
    load_row_consts<HM>(I, H, row, ad, ve);
#pragma unroll
    for (int h = 0; h < HM; ++h) {
      m[h] = glob(I.smax)[(int64_t)row * GAT_HMAX + min(h, H - 1)];  // heads >= H: a duplicate nobody reads
      den[h] = glob(I.sden)[(int64_t)row * GAT_HMAX + min(h, H - 1)];
    }
    const bool adrop = dyn.training && I.adrop_p > 0.f;
    DropCfg acfg;
    if (adrop) acfg = make_cfg(dyn, I.adrop_p, I.adrop_stream);
    const int b = glob(I.rowptr)[row], e = glob(I.rowptr)[row + 1];
    const int kend = e + ((row < n_loop) ? 1 : 0);

    float tsum[HM], dsum[HM];
#pragma unroll
    for (int h = 0; h < HM; ++h) { tsum[h] = 0.f; dsum[h] = 0.f; }

    constexpr int UB = (HM <= 4) ? 4 : 2;
    const int cc = cact ? c0 : 0;
    // Sweep 0 needs every neighbour row (d alpha' = <g, h_s[j]>) to form sum_k alpha_k d alpha_k; sweep 1 needs only the
    // per-(edge, head) scalars again.  The first CAP edges of a row park them in LDS {alpha, d alpha, leaky slope, dropout
    // scale} (alpha < 0 marks a removed self loop), so sweep 1 re-reads NO neighbour rows for them: one lane per edge
    // finishes d logit and stores it.  Edges past CAP take the old path (fetch + recompute).
    float (*crow)[HM][4] = cache[threadIdx.x / GS];
    const int n_all = kend - b;
    const int n_cached = n_all < CAP ? n_all : CAP;
    KT(2 + 4 * ii);
    for (int k0 = b; k0 < kend; k0 += UB) {  // ---- sweep 0
      EdgeBatch<HM, UB> B;
      fetch_batch<HM, UB>(I, D.Cp, ea, H, row, b, e, kend, k0, cc, ad, ve, B);
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int k = k0 + u;
        const int idx = k - b;
        if (B.removed[u]) {
          if (gl == 0 && idx < CAP) {
#pragma unroll
            for (int h = 0; h < HM; ++h) crow[idx][h][0] = -1.f;
          }
          continue;
        }
        if (!B.live[u]) continue;
        const int64_t pos = B.loop[u] ? (E + row) : (int64_t)k;
        bool keep[HM];
#pragma unroll
        for (int h = 0; h < HM; ++h) keep[h] = true;
        if (adrop) alpha_keep<HM>(acfg, pos, keep);
#pragma unroll
        for (int h = 0; h < HM; ++h) {
          if (h >= H) continue;
          float part = 0.f;
          if (cact) part = dot4(g[h], B.v[u][h]);
          const float dap = group_sum<GS>(part);  // d alpha'_k,h (identical in every lane of the group)
          const float rw = B.raw[u][h];
          const float ev = rw > 0.f ? rw : NEG_SLOPE * rw;
          const float alpha = expf(ev - m[h]) / den[h];
          const float dscale = adrop ? (keep[h] ? acfg.scale : 0.f) : 1.f;
          const float da = dap * dscale;
          tsum[h] += alpha * da;
          if (gl == 0 && idx < CAP) {
            crow[idx][h][0] = alpha; crow[idx][h][1] = da; crow[idx][h][2] = (rw > 0.f ? 1.f : NEG_SLOPE); crow[idx][h][3] = dscale;
          }
        }
      }
    }
    // writer (lane 0 of the group) and readers share a wavefront: order the LDS traffic, no block barrier needed
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    KT(3 + 4 * ii);
    {  // ---- sweep 1, cached edges: lane gl takes edges gl, gl + GS, ...
      float dpart[HM];
#pragma unroll
      for (int h = 0; h < HM; ++h) dpart[h] = 0.f;
      for (int idx = gl; idx < n_cached; idx += GS) {
        const int k = b + idx;
        const bool is_loop = k >= e;
        const int64_t pos = is_loop ? (E + row) : (int64_t)k;
        const int eo = (I.dlogit_orig && !is_loop) ? glob(I.eid)[k] : 0;
#pragma unroll
        for (int h = 0; h < HM; ++h) {
          if (h >= H) continue;
          const float alpha = crow[idx][h][0];
          float apd = 0.f, dl = 0.f;
          if (alpha >= 0.f) {
            const float de = alpha * (crow[idx][h][1] - tsum[h]);
            dl = de * crow[idx][h][2];
            apd = alpha * crow[idx][h][3];
            dpart[h] += dl;
          }
          globw(I.alpha_drop)[pos * GAT_HMAX + h] = apd;
          globw(I.dlogit)[pos * GAT_HMAX + h] = dl;
          if (I.dlogit_orig && !is_loop) globw(I.dlogit_orig)[(int64_t)eo * GAT_HMAX + h] = dl;
        }
      }
#pragma unroll
      for (int h = 0; h < HM; ++h) dsum[h] = group_sum<GS>(dpart[h]);
    }
    for (int k0 = b + CAP; k0 < kend; k0 += UB) {  // ---- sweep 1, edges past the cache: recompute
      EdgeBatch<HM, UB> B;
      fetch_batch<HM, UB>(I, D.Cp, ea, H, row, b, e, kend, k0, cc, ad, ve, B);
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int k = k0 + u;
        const int64_t pos = B.loop[u] ? (E + row) : (int64_t)k;
        if (B.removed[u]) {  // removed self loop: contributes nothing
          if (gl == 0) {
#pragma unroll
            for (int h = 0; h < HM; ++h)
              if (h < H) {
                globw(I.alpha_drop)[pos * GAT_HMAX + h] = 0.f;
                globw(I.dlogit)[pos * GAT_HMAX + h] = 0.f;
                if (I.dlogit_orig) globw(I.dlogit_orig)[(int64_t)glob(I.eid)[k] * GAT_HMAX + h] = 0.f;
              }
          }
          continue;
        }
        if (!B.live[u]) continue;
        bool keep[HM];
#pragma unroll
        for (int h = 0; h < HM; ++h) keep[h] = true;
        if (adrop) alpha_keep<HM>(acfg, pos, keep);
#pragma unroll
        for (int h = 0; h < HM; ++h) {
          if (h >= H) continue;
          float part = 0.f;
          if (cact) part = dot4(g[h], B.v[u][h]);
          const float dap = group_sum<GS>(part);
          const float rw = B.raw[u][h];
          const float ev = rw > 0.f ? rw : NEG_SLOPE * rw;
          const float alpha = expf(ev - m[h]) / den[h];
          const float dscale = adrop ? (keep[h] ? acfg.scale : 0.f) : 1.f;
          const float da = dap * dscale;
          const float de = alpha * (da - tsum[h]);
          const float dl = de * (rw > 0.f ? 1.f : NEG_SLOPE);
          dsum[h] += dl;
          if (gl == 0) {
            globw(I.alpha_drop)[pos * GAT_HMAX + h] = alpha * dscale;
            globw(I.dlogit)[pos * GAT_HMAX + h] = dl;
            if (I.dlogit_orig && !B.loop[u]) globw(I.dlogit_orig)[(int64_t)B.eid[u] * GAT_HMAX + h] = dl;
          }
        }
      }
    }
    if (gl == 0) {
#pragma unroll
      for (int h = 0; h < HM; ++h)
        if (h < H) globw(I.dz_dst)[(int64_t)row * I.lddz_dst + I.adoff + h] = dsum[h];
    }
    KT(4 + 4 * ii);
  }
}

// =================================================================================================
// backward pass 2: source major over the CSC lists.
//   d h_s[j,h,:] = sum_{k in out(j)} alpha'_k,h * g[i_k,h,:],   d a_src[j,h] = sum_k d raw_k,h
// =================================================================================================
template <int HM, int GS>
__global__ __launch_bounds__(256) void gat_bwd2_kernel(const GatLayerS* __restrict__ tab, const GatDyn dyn) {
  int si = 0;
  while (si + 1 < tab->n_src && (int)blockIdx.x >= dyn.block_start[si + 1]) ++si;
  const GatSrcS& S = tab->s[si];
  const int rpb = 256 / GS;
  const int row = ((int)blockIdx.x - dyn.block_start[si]) * rpb + (int)threadIdx.x / GS;
  if (row >= dyn.n_nodes[S.t]) return;
  const int gl = threadIdx.x % GS, c0 = gl * 4;

  for (int oi = 0; oi < S.n_out; ++oi) {
    const GatDstS& D = tab->d[S.out[oi].d];
    const GatInS& I = D.in[S.out[oi].i];
    const bool cact = c0 < D.Cp;
    const int H = D.H;
    const int64_t E = dyn.n_edges[I.et];
    const int n_loop = I.self_loops ? min(dyn.n_nodes[I.src_t], dyn.n_nodes[D.t]) : 0;
    const float* gp = D.g ? D.g : dyn.g_top;
    const int ldg = D.g ? D.ldg : dyn.ld_gtop;
    float4 acc[HM];
    float das[HM];
#pragma unroll
    for (int h = 0; h < HM; ++h) { acc[h] = make_float4(0.f, 0.f, 0.f, 0.f); das[h] = 0.f; }
    const int b = glob(I.t_rowptr)[row], e = glob(I.t_rowptr)[row + 1];
    const int kend = e + ((row < n_loop) ? 1 : 0);
    // vector path for the gradient rows (C % 4 == 0, 16-byte aligned rows: the executor's buffers); uniform
    const bool vec = (D.C & 3) == 0 && (ldg & 3) == 0 && (reinterpret_cast<uintptr_t>(gp) & 15) == 0;
    const int cc = cact ? c0 : 0;
    // out-edges in batches of 4: destination ids + CSR positions first, then the per-edge scalars and gradient rows of the
    // whole batch in flight together (all loads unconditional: head index clamped, see edge_logits); the adds keep the edge order
    constexpr int UB = (HM <= 4) ? 4 : 2;
    for (int k0 = b; k0 < kend; k0 += UB) {
      int ii[UB];
      int64_t pp[UB];
      bool in[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int k = k0 + u;
        in[u] = k < kend;
        const bool lp = in[u] && k >= e;
        const int kc = (e > b) ? min(k, e - 1) : 0;
        const int ti = glob(I.t_col)[kc];
        const int tp = glob(I.t_pos)[kc];
        const bool real = in[u] && !lp;
        ii[u] = lp ? row : (real ? ti : 0);
        pp[u] = lp ? (E + row) : (real ? (int64_t)tp : 0);
      }
      float ap[UB][HM], dl[UB][HM];
      float4 gv[UB][HM];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const float* gr = gp + (int64_t)ii[u] * ldg;
#pragma unroll
        for (int h = 0; h < HM; ++h) {
          const int hc = min(h, H - 1);
          ap[u][h] = globw(I.alpha_drop)[pp[u] * GAT_HMAX + hc];
          dl[u][h] = globw(I.dlogit)[pp[u] * GAT_HMAX + hc];
        }
        if (vec) {
#pragma unroll
          for (int h = 0; h < HM; ++h) gv[u][h] = ld4(gr + (D.concat ? min(h, H - 1) * D.C : 0) + cc);
        } else {
#pragma unroll
          for (int h = 0; h < HM; ++h) {
            float t4[4] = {0.f, 0.f, 0.f, 0.f};
            if (cact && h < H && (D.concat || h == 0)) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const int c = c0 + q;
                if (c < D.C) t4[q] = D.concat ? glob(gr)[h * D.C + c] : glob(gr)[c];
              }
            }
            gv[u][h] = make_float4(t4[0], t4[1], t4[2], t4[3]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        if (!in[u]) continue;
#pragma unroll
        for (int h = 0; h < HM; ++h) {
          if (h >= H) continue;
          das[h] += dl[u][h];
          if (cact) {
            const float4 gq = D.concat ? gv[u][h] : gv[u][0];
            const float w = ap[u][h] * D.group_scale;
            if (D.concat) {
              acc[h].x += w * gq.x; acc[h].y += w * gq.y; acc[h].z += w * gq.z; acc[h].w += w * gq.w;
            } else {
              const float fh = (float)H;
              acc[h].x += w * (gq.x / fh); acc[h].y += w * (gq.y / fh); acc[h].z += w * (gq.z / fh); acc[h].w += w * (gq.w / fh);
            }
          }
        }
      }
    }
    float* dzr = S.dz + (int64_t)row * S.lddz;
#pragma unroll
    for (int h = 0; h < HM; ++h) {
      if (h >= H) continue;
      if (cact) st4(dzr + I.hoff + h * D.Cp + c0, acc[h]);
      if (gl == 0) globw(I.dza_src)[(int64_t)row * I.lddza_src + I.asoff + h] = das[h];
    }
  }
}

template <typename F>
int dispatch(int H, int Cp, F&& f) {
  const int lanes = Cp / 4;
  int gs = 8;
  while (gs < 64 && gs < lanes) gs <<= 1;
  HMP_CHECK_ARG(lanes <= 64, "gat: channels per head %d > 256 not supported", Cp);
  HMP_CHECK_ARG(H >= 1 && H <= GAT_HMAX, "gat: heads %d not in [1, %d]", H, GAT_HMAX);
  const int hm = H <= 1 ? 1 : (H <= 2 ? 2 : (H <= 4 ? 4 : 8));
  return f(hm, gs);
}

#define GAT_LAUNCH(KERNEL, hm, gs, blocks, st, d_tab, dyn)                                                   \
  do {                                                                                                       \
    switch ((hm)*100 + (gs)) {                                                                               \
      case 108: hipLaunchKernelGGL((KERNEL<1, 8>), dim3(blocks), dim3(256), 0, st, d_tab, dyn); break;       \
      case 116: hipLaunchKernelGGL((KERNEL<1, 16>), dim3(blocks), dim3(256), 0, st, d_tab, dyn); break;      \
      case 132: hipLaunchKernelGGL((KERNEL<1, 32>), dim3(blocks), dim3(256), 0, st, d_tab, dyn); break;      \
      case 164: hipLaunchKernelGGL((KERNEL<1, 64>), dim3(blocks), dim3(256), 0, st, d_tab, dyn); break;      \
      case 208: hipLaunchKernelGGL((KERNEL<2, 8>), dim3(blocks), dim3(256), 0, st, d_tab, dyn); break;       \
      case 216: hipLaunchKernelGGL((KERNEL<2, 16>), dim3(blocks), dim3(256), 0, st, d_tab, dyn); break;      \
      case 232: hipLaunchKernelGGL((KERNEL<2, 32>), dim3(blocks), dim3(256), 0, st, d_tab, dyn); break;      \
      case 264: hipLaunchKernelGGL((KERNEL<2, 64>), dim3(blocks), dim3(256), 0, st, d_tab, dyn); break;      \
      case 408: hipLaunchKernelGGL((KERNEL<4, 8>), dim3(blocks), dim3(256), 0, st, d_tab, dyn); break;       \
      case 416: hipLaunchKernelGGL((KERNEL<4, 16>), dim3(blocks), dim3(256), 0, st, d_tab, dyn); break;      \
      case 432: hipLaunchKernelGGL((KERNEL<4, 32>), dim3(blocks), dim3(256), 0, st, d_tab, dyn); break;      \
      case 464: hipLaunchKernelGGL((KERNEL<4, 64>), dim3(blocks), dim3(256), 0, st, d_tab, dyn); break;      \
      case 808: hipLaunchKernelGGL((KERNEL<8, 8>), dim3(blocks), dim3(256), 0, st, d_tab, dyn); break;       \
      case 816: hipLaunchKernelGGL((KERNEL<8, 16>), dim3(blocks), dim3(256), 0, st, d_tab, dyn); break;      \
      case 832: hipLaunchKernelGGL((KERNEL<8, 32>), dim3(blocks), dim3(256), 0, st, d_tab, dyn); break;      \
      case 864: hipLaunchKernelGGL((KERNEL<8, 64>), dim3(blocks), dim3(256), 0, st, d_tab, dyn); break;      \
      default: HMP_FAIL(HMP_E_ARG, "gat: no kernel for heads-class %d, group %d", hm, gs);                   \
    }                                                                                                        \
  } while (0)

// all destination entries of a layer share H (the reference builds every edge type of a layer alike); the channel count may
// differ between destination types (last layer of the two-headed task: output_dim_dict): the kernels read Cp per destination
// and mask their lanes with it, so the launch takes the row-group width of the widest one
int layer_shape(const GatLayerS& h_tab, int& H, int& Cp) {
  HMP_CHECK_ARG(h_tab.n_dst > 0, "gat: empty layer table");
  H = h_tab.d[0].H; Cp = h_tab.d[0].Cp;
  for (int i = 1; i < h_tab.n_dst; ++i) {
    HMP_CHECK_ARG(h_tab.d[i].H == H, "gat: destination types of one layer must share the number of heads");
    Cp = h_tab.d[i].Cp > Cp ? h_tab.d[i].Cp : Cp;
  }
  return HMP_OK;
}

}  // namespace

int gat_fwd_launch(const GatLayerS* d_tab, const GatLayerS& h_tab, GatDyn& dyn, hipStream_t st) {
  int H, Cp;
  HMP_TRY(layer_shape(h_tab, H, Cp));
  return dispatch(H, Cp, [&](int hm, int gs) -> int {
    int blocks = 0;
    for (int i = 0; i < h_tab.n_dst; ++i) {
      dyn.block_start[i] = blocks;
      blocks += cdiv(dyn.n_nodes[h_tab.d[i].t], 256 / gs);
    }
    dyn.block_start[h_tab.n_dst] = blocks;
    if (blocks == 0) return HMP_OK;
    GAT_LAUNCH(gat_fwd_kernel, hm, gs, blocks, st, d_tab, dyn);
    HMP_LAUNCH_CHECK();
    return HMP_OK;
  });
}

int gat_bwd1_launch(const GatLayerS* d_tab, const GatLayerS& h_tab, GatDyn& dyn, hipStream_t st) {
  int H, Cp;
  HMP_TRY(layer_shape(h_tab, H, Cp));
  return dispatch(H, Cp, [&](int hm, int gs) -> int {
    int blocks = 0;
    for (int i = 0; i < h_tab.n_dst; ++i) {
      dyn.block_start[i] = blocks;
      blocks += cdiv(dyn.n_nodes[h_tab.d[i].t], 256 / gs);
    }
    dyn.block_start[h_tab.n_dst] = blocks;
    if (blocks == 0) return HMP_OK;
    GAT_LAUNCH(gat_bwd1_kernel, hm, gs, blocks, st, d_tab, dyn);
    HMP_LAUNCH_CHECK();
    return HMP_OK;
  });
}

int gat_bwd2_launch(const GatLayerS* d_tab, const GatLayerS& h_tab, GatDyn& dyn, hipStream_t st) {
  int H, Cp;
  HMP_TRY(layer_shape(h_tab, H, Cp));
  return dispatch(H, Cp, [&](int hm, int gs) -> int {
    int blocks = 0;
    for (int i = 0; i < h_tab.n_src; ++i) {
      dyn.block_start[i] = blocks;
      blocks += cdiv(dyn.n_nodes[h_tab.s[i].t], 256 / gs);
    }
    dyn.block_start[h_tab.n_src] = blocks;
    if (blocks == 0) return HMP_OK;
    GAT_LAUNCH(gat_bwd2_kernel, hm, gs, blocks, st, d_tab, dyn);
    HMP_LAUNCH_CHECK();
    return HMP_OK;
  });
}

}  // namespace hmp

// =================================================================================================
// unit entry points (one conv, concat layout, no bias / activation): build a one-entry table, run, synchronise
// =================================================================================================
namespace {

using namespace hmp;

struct UnitTab {
  GatLayerS h;
  GatLayerS* d = nullptr;
  ~UnitTab() {
    if (d) (void)hipFree(d);
  }
};

int unit_setup(UnitTab& u, GatDyn& dyn, const hmp_plan& plan, const hmp_gat_args& a, const float* h_src, int ldh, const float* a_src,
               int lda_s, const float* a_dst, int lda_d, const float* edge_attr, const float* v_edge) {
  HMP_CHECK_ARG(a.heads >= 1 && a.heads <= GAT_HMAX && a.channels >= 1 && a.channels <= 256, "hmp_gat: heads/channels out of range");
  HMP_CHECK_ARG(a.edge_dim >= 0 && a.edge_dim <= GAT_MAX_EDIM, "hmp_gat: edge_dim out of range");
  HMP_CHECK_ARG(a.edge_dim == 0 || (edge_attr && v_edge) || plan.n_edges == 0, "hmp_gat: edge_attr / v_edge missing");
  HMP_CHECK_ARG((ldh & 3) == 0 && ldh >= a.heads * align4(a.channels), "hmp_gat: ldh must be a multiple of 4 and >= H*Cp");
  HMP_CHECK_ARG(a.dropout_p >= 0.f && a.dropout_p < 1.f, "hmp_gat: dropout_p");
  memset(&u.h, 0, sizeof(u.h));
  memset(&dyn, 0, sizeof(dyn));
  GatLayerS& G = u.h;
  G.n_dst = 1; G.n_src = 1;
  GatDstS& D = G.d[0];
  D.t = 1; D.H = a.heads; D.C = a.channels; D.Cp = align4(a.channels); D.concat = 1;
  D.act = HMP_ACT_NONE; D.group_scale = 1.f; D.n_in = 1;
  GatInS& I = D.in[0];
  I.et = 0; I.src_t = 0;
  I.rowptr = plan.d_rowptr; I.col = plan.d_col; I.eid = plan.d_eid;
  I.t_rowptr = plan.d_t_rowptr; I.t_col = plan.d_t_col; I.t_pos = plan.d_t_pos;
  I.z = h_src; I.ldz = ldh; I.hoff = 0;
  I.za = a_src; I.ldza = lda_s; I.asoff = 0;
  I.zd = a_dst; I.ldzd = lda_d; I.adoff = 0;
  I.vedge = a.edge_dim > 0 ? v_edge : nullptr;
  I.edim = a.edge_dim;
  I.self_loops = a.self_loops;
  I.adrop_p = a.dropout_p; I.adrop_stream = a.rng_stream;
  G.s[0].t = 0; G.s[0].n_out = 1; G.s[0].out[0].d = 0; G.s[0].out[0].i = 0;
  dyn.n_nodes[0] = plan.n_src; dyn.n_nodes[1] = plan.n_dst;
  dyn.n_edges[0] = plan.n_edges;
  dyn.edge_attr[0] = edge_attr;
  dyn.training = a.dropout_p > 0.f ? 1 : 0;
  dyn.k0 = (uint32_t)a.seed; dyn.k1 = (uint32_t)(a.seed >> 32);
  dyn.step = a.rng_step;
  return HMP_OK;
}

int unit_upload(UnitTab& u) {
  HMP_HIP(hipMalloc(&u.d, sizeof(GatLayerS)));
  HMP_HIP(hipMemcpy(u.d, &u.h, sizeof(GatLayerS), hipMemcpyHostToDevice));
  return HMP_OK;
}

}  // namespace

extern "C" int hmp_gat_fwd(const float* d_h_src, int32_t ldh, const float* d_a_src, int32_t lda_src, const float* d_a_dst,
                           int32_t lda_dst, const float* d_edge_attr, const float* d_v_edge, hmp_plan plan, hmp_gat_args args,
                           float* d_smax, float* d_sden, float* d_out, int32_t ldo, void* stream) {
  HMP_CHECK_ARG(d_h_src && d_a_src && d_a_dst && d_smax && d_sden && d_out, "hmp_gat_fwd: null pointer");
  HMP_CHECK_ARG(ldo >= args.heads * args.channels, "hmp_gat_fwd: ldo < H*C");
  UnitTab u;
  GatDyn dyn;
  HMP_TRY(unit_setup(u, dyn, plan, args, d_h_src, ldh, d_a_src, lda_src, d_a_dst, lda_dst, d_edge_attr, d_v_edge));
  u.h.d[0].out = d_out; u.h.d[0].ldo = ldo;
  u.h.d[0].in[0].smax = d_smax; u.h.d[0].in[0].sden = d_sden;
  HMP_TRY(unit_upload(u));
  HMP_TRY(gat_fwd_launch(u.d, u.h, dyn, (hipStream_t)stream));
  HMP_HIP(hipStreamSynchronize((hipStream_t)stream));
  return HMP_OK;
}

extern "C" int hmp_gat_bwd(const float* d_gout, int32_t ldg, const float* d_h_src, int32_t ldh, const float* d_a_src, int32_t lda_src,
                           const float* d_a_dst, int32_t lda_dst, const float* d_edge_attr, const float* d_v_edge, hmp_plan plan,
                           hmp_gat_args args, const float* d_smax, const float* d_sden, float* d_alpha_drop, float* d_dlogit,
                           float* d_dlogit_orig, float* d_g_h_src, int32_t ldgh, float* d_g_a_src, int32_t ldgas, float* d_g_a_dst,
                           int32_t ldgad, void* stream) {
  HMP_CHECK_ARG(d_gout && d_h_src && d_a_src && d_a_dst && d_smax && d_sden && d_alpha_drop && d_dlogit && d_g_h_src && d_g_a_src && d_g_a_dst,
                "hmp_gat_bwd: null pointer");
  HMP_CHECK_ARG((ldgh & 3) == 0 && (reinterpret_cast<uintptr_t>(d_g_h_src) & 15) == 0, "hmp_gat_bwd: g_h_src must be 16-byte aligned, ld %% 4 == 0");
  UnitTab u;
  GatDyn dyn;
  HMP_TRY(unit_setup(u, dyn, plan, args, d_h_src, ldh, d_a_src, lda_src, d_a_dst, lda_dst, d_edge_attr, d_v_edge));
  GatInS& I = u.h.d[0].in[0];
  I.smax = const_cast<float*>(d_smax); I.sden = const_cast<float*>(d_sden);
  I.alpha_drop = d_alpha_drop; I.dlogit = d_dlogit; I.dlogit_orig = d_dlogit_orig;
  I.dza_src = d_g_a_src; I.lddza_src = ldgas;
  I.dz_dst = d_g_a_dst; I.lddz_dst = ldgad;
  u.h.s[0].dz = d_g_h_src; u.h.s[0].lddz = ldgh;
  dyn.g_top = d_gout; dyn.ld_gtop = ldg;
  HMP_TRY(unit_upload(u));
  HMP_TRY(gat_bwd1_launch(u.d, u.h, dyn, (hipStream_t)stream));
  HMP_TRY(gat_bwd2_launch(u.d, u.h, dyn, (hipStream_t)stream));
  HMP_HIP(hipStreamSynchronize((hipStream_t)stream));
  return HMP_OK;
}
