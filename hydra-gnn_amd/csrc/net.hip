// Network executor: the HeteroConv layer stack of Hydra-GNN's room classifier as one native launch
// sequence (plan -> pack -> per layer {grouped MFMA projection, fused aggregation} -> loss -> per layer
// {transposed aggregation, grouped input-gradient GEMM, grouped split-K weight-gradient GEMM} ->
// gradient un-pack -> Adam).  Shapes come from the batch at call time, every buffer lives in one
// caller-provided workspace, nothing allocates or synchronises, so the whole step can be captured into a
// hipGraph and replayed.
//
// SAGE algebra (exact up to fp32 summation order):
//   HeteroConv-sum over the convs reaching node type t:
//     out_t = sum_e [ W_l,e * mean_e(x_src) + b_e ] + (sum_e W_r,e) * x_t
//   mean and projection commute, so every source type s is projected ONCE per layer by the stacked
//   operand  Wp[l][s] = [ W_l,e1 ; W_l,e2 ; ... ; sum_e W_r,e ]  (Z = x_s * Wp^T), and the aggregation
//   kernel gathers the (narrow) projected rows.  Weight gradients come back de-stacked; every W_r,e of a
//   node type receives the same gradient, every b_e the column sum of the destination gradient.
#include <vector>

#include "kernels.h"

using namespace hmp;

namespace {

struct ConvLayout {
  int coff;  // column of this conv's segment in Z[l][src]
};

struct LayerLayout {
  int n_live;
  int live[HMP_MAX_CONVS];            // indices of live convs
  ConvLayout conv[HMP_MAX_CONVS];
  int ncols[HMP_MAX_NODE_TYPES];      // width of Z[l][s] (0: type is not a live source / root)
  int roff[HMP_MAX_NODE_TYPES];       // root segment column in Z[l][t], -1 if t receives nothing
  int64_t wp_off[HMP_MAX_NODE_TYPES]; // packed weights [ncols][ldw]
  int ldw[HMP_MAX_NODE_TYPES];
  int64_t bias_off[HMP_MAX_NODE_TYPES];
  int64_t slab_off[HMP_MAX_NODE_TYPES];  // packed-gradient slabs [max_slabs][ncols][lddw]
  int lddw[HMP_MAX_NODE_TYPES];
};

constexpr int MAX_SLABS = 16;

struct ProfRec {
  int cls;
  hipEvent_t a, b;
};

}  // namespace

struct hmp_net {
  hmp_net_spec spec;
  int T, ET, L;
  int dim[HMP_MAX_LAYERS + 1][HMP_MAX_NODE_TYPES];  // feature width of H[l][t]
  int ld[HMP_MAX_LAYERS + 1][HMP_MAX_NODE_TYPES];   // leading dimension of our own H/G buffers (l >= 1)
  LayerLayout lay[HMP_MAX_LAYERS];
  int64_t packed_floats, slab_floats;
  int out_dim, out_ld;

  // static device tables (owned)
  PackSeg* d_pack_segs = nullptr;
  int64_t* d_pack_row_start = nullptr;
  int n_pack = 0;
  int64_t pack_rows = 0;
  GradSeg* d_grad_segs = nullptr;
  int64_t* d_grad_elem_start = nullptr;
  int n_grad = 0;
  int64_t grad_elems = 0;

  // workspace binding
  bool bound = false;
  int cap_nodes[HMP_MAX_NODE_TYPES];
  int64_t cap_edges[HMP_MAX_EDGE_TYPES];
  NetState* d_state = nullptr;
  hmp_plan plan[HMP_MAX_EDGE_TYPES];
  int* plan_scratch[HMP_MAX_EDGE_TYPES];
  float* d_packed = nullptr;
  float* d_slabs = nullptr;
  float* H[HMP_MAX_LAYERS + 1][HMP_MAX_NODE_TYPES];
  float* G[HMP_MAX_LAYERS + 1][HMP_MAX_NODE_TYPES];
  float* Z[HMP_MAX_LAYERS][HMP_MAX_NODE_TYPES];
  float* dZ[HMP_MAX_LAYERS][HMP_MAX_NODE_TYPES];
  float* d_out = nullptr;   // pooled output [cap_out, out_ld] (only with pool_edge_type)
  float* d_gout = nullptr;  // loss gradient [cap_out, out_ld]
  int cap_out = 0;

  // last forward
  bool have_fwd = false;
  hmp_batch batch;
  int training = 0;
  uint64_t seed = 0;
  uint32_t rng_step = 0;
  bool step_dev = false;  // dropout step offset read from d_state->step
  GradReduceDyn dyn;

  // profiling
  bool prof = false;
  std::vector<ProfRec> recs;
};

namespace {

inline int fpad(int f) { return align4(f); }

int build_layout(hmp_net* n) {
  const hmp_net_spec& S = n->spec;
  n->T = S.n_node_types; n->ET = S.n_edge_types; n->L = S.n_layers;
  HMP_CHECK_ARG(n->T >= 1 && n->T <= HMP_MAX_NODE_TYPES, "net: n_node_types %d", n->T);
  HMP_CHECK_ARG(n->ET >= 1 && n->ET <= HMP_MAX_EDGE_TYPES, "net: n_edge_types %d", n->ET);
  HMP_CHECK_ARG(n->L >= 1 && n->L <= HMP_MAX_LAYERS, "net: n_layers %d", n->L);
  HMP_CHECK_ARG(S.readout_type >= 0 && S.readout_type < n->T, "net: readout_type");
  HMP_CHECK_ARG(S.pool_edge_type >= -1 && S.pool_edge_type < n->ET, "net: pool_edge_type");
  for (int e = 0; e < n->ET; ++e)
    HMP_CHECK_ARG(S.edge_src[e] >= 0 && S.edge_src[e] < n->T && S.edge_dst[e] >= 0 && S.edge_dst[e] < n->T, "net: edge type %d endpoints", e);
  for (int t = 0; t < n->T; ++t) {
    n->dim[0][t] = S.in_dim[t];
    n->ld[0][t] = 0;  // comes with the batch
  }
  int64_t packed = 0, slabs = 0;
  for (int l = 0; l < n->L; ++l) {
    const hmp_layer_spec& Ls = S.layers[l];
    LayerLayout& Y = n->lay[l];
    HMP_CHECK_ARG(Ls.n_convs >= 1 && Ls.n_convs <= HMP_MAX_CONVS, "net: layer %d has %d convs", l, Ls.n_convs);
    HMP_CHECK_ARG(Ls.group_mean == 0, "net: HeteroConv aggr=mean is only supported for GAT pre_mp");
    for (int t = 0; t < n->T; ++t) {
      n->dim[l + 1][t] = Ls.out_dim[t];
      n->ld[l + 1][t] = fpad(Ls.out_dim[t]);
      Y.ncols[t] = 0;
      Y.roff[t] = -1;
    }
    Y.n_live = 0;
    int n_in[HMP_MAX_NODE_TYPES] = {0}, n_outgoing[HMP_MAX_NODE_TYPES] = {0};
    for (int c = 0; c < Ls.n_convs; ++c) {
      const hmp_conv_spec& C = Ls.convs[c];
      HMP_CHECK_ARG(C.kind == HMP_CONV_SAGE, "net: layer %d conv %d: only SAGE convs run in this executor build", l, c);
      HMP_CHECK_ARG(C.edge_type >= 0 && C.edge_type < n->ET, "net: conv edge_type");
      HMP_CHECK_ARG(C.src == S.edge_src[C.edge_type] && C.dst == S.edge_dst[C.edge_type], "net: conv endpoints disagree with edge type");
      HMP_CHECK_ARG(C.f_out == Ls.out_dim[C.dst], "net: SAGE f_out must equal the layer's out_dim of the destination type");
      HMP_CHECK_ARG(n->dim[l][C.src] > 0 && n->dim[l][C.dst] > 0, "net: layer %d conv %d reads a node type with no features", l, c);
      if (!C.active) continue;
      HMP_CHECK_ARG(C.w0 >= 0 && C.b0 >= 0 && C.w1 >= 0, "net: SAGE conv needs lin_l.weight, lin_l.bias, lin_r.weight");
      Y.live[Y.n_live++] = c;
      Y.conv[c].coff = Y.ncols[C.src];
      Y.ncols[C.src] += fpad(C.f_out);
      ++n_in[C.dst];
      ++n_outgoing[C.src];
      HMP_CHECK_ARG(n_in[C.dst] <= AGG_MAX_IN && n_outgoing[C.src] <= AGG_MAX_IN, "net: more than %d convs share a node type", AGG_MAX_IN);
    }
    HMP_CHECK_ARG(Y.n_live > 0, "net: layer %d has no live conv", l);
    for (int t = 0; t < n->T; ++t) {
      if (n_in[t] > 0) {
        Y.roff[t] = Y.ncols[t];
        Y.ncols[t] += fpad(Ls.out_dim[t]);
      }
    }
    for (int t = 0; t < n->T; ++t) {
      Y.ldw[t] = fpad(n->dim[l][t]);
      Y.lddw[t] = fpad(n->dim[l][t] + 1);
      Y.wp_off[t] = packed;
      packed += (int64_t)Y.ncols[t] * Y.ldw[t];
      Y.bias_off[t] = packed;
      packed += (Y.roff[t] >= 0) ? fpad(Ls.out_dim[t]) : 0;
      Y.slab_off[t] = slabs;
      slabs += (int64_t)MAX_SLABS * Y.ncols[t] * Y.lddw[t];
    }
  }
  // liveness must be closed: whatever a live conv produces is consumed by the next layer / the readout
  for (int l = 0; l < n->L; ++l) {
    const LayerLayout& Y = n->lay[l];
    for (int i = 0; i < Y.n_live; ++i) {
      const hmp_conv_spec& C = S.layers[l].convs[Y.live[i]];
      if (l == n->L - 1)
        HMP_CHECK_ARG(C.dst == S.readout_type, "net: last-layer conv %d is active but does not feed the readout type", Y.live[i]);
      else
        HMP_CHECK_ARG(n->lay[l + 1].ncols[C.dst] > 0, "net: layer %d conv %d is active but layer %d never reads its output", l, Y.live[i], l + 1);
    }
  }
  n->packed_floats = packed;
  n->slab_floats = slabs;
  n->out_dim = n->dim[n->L][S.readout_type];
  n->out_ld = fpad(n->out_dim);
  HMP_CHECK_ARG(n->out_dim > 0, "net: readout type has no output in the last layer");
  return HMP_OK;
}

// static pack / grad tables ------------------------------------------------------------------------
int build_tables(hmp_net* n) {
  const hmp_net_spec& S = n->spec;
  std::vector<PackSeg> ps;
  std::vector<int64_t> prs;
  std::vector<GradSeg> gs;
  std::vector<int64_t> ges;
  int64_t prow = 0, gel = 0;
  for (int l = 0; l < n->L; ++l) {
    const hmp_layer_spec& Ls = S.layers[l];
    const LayerLayout& Y = n->lay[l];
    for (int i = 0; i < Y.n_live; ++i) {  // W_l segments
      const int c = Y.live[i];
      const hmp_conv_spec& C = Ls.convs[c];
      const int fs = n->dim[l][C.src];
      PackSeg s;
      memset(&s, 0, sizeof(s));
      s.dst = Y.wp_off[C.src] + (int64_t)Y.conv[c].coff * Y.ldw[C.src];
      s.rows = C.f_out; s.rows_pad = fpad(C.f_out); s.cols = fs; s.ld_dst = Y.ldw[C.src]; s.ld_src = fs;
      s.nsrc = 1; s.src[0] = C.w0;
      ps.push_back(s); prs.push_back(prow); prow += s.rows_pad;
      GradSeg g;
      g.dst = C.w0; g.rows = C.f_out; g.cols = fs;
      g.src = Y.slab_off[C.src] + (int64_t)Y.conv[c].coff * Y.lddw[C.src];
      g.ld_src = Y.lddw[C.src]; g.slab_id = l * HMP_MAX_NODE_TYPES + C.src;
      gs.push_back(g); ges.push_back(gel); gel += (int64_t)g.rows * g.cols;
    }
    for (int t = 0; t < n->T; ++t) {  // root + bias segments
      if (Y.roff[t] < 0) continue;
      const int fo = Ls.out_dim[t], ft = n->dim[l][t];
      PackSeg s, b;
      memset(&s, 0, sizeof(s));
      memset(&b, 0, sizeof(b));
      s.dst = Y.wp_off[t] + (int64_t)Y.roff[t] * Y.ldw[t];
      s.rows = fo; s.rows_pad = fpad(fo); s.cols = ft; s.ld_dst = Y.ldw[t]; s.ld_src = ft;
      b.dst = Y.bias_off[t];
      b.rows = 1; b.rows_pad = 1; b.cols = fo; b.ld_dst = fpad(fo); b.ld_src = fo;
      for (int i = 0; i < Y.n_live; ++i) {
        const hmp_conv_spec& C = Ls.convs[Y.live[i]];
        if (C.dst != t) continue;
        s.src[s.nsrc++] = C.w1;
        b.src[b.nsrc++] = C.b0;
        GradSeg g;
        g.dst = C.w1; g.rows = fo; g.cols = ft;
        g.src = Y.slab_off[t] + (int64_t)Y.roff[t] * Y.lddw[t];
        g.ld_src = Y.lddw[t]; g.slab_id = l * HMP_MAX_NODE_TYPES + t;
        gs.push_back(g); ges.push_back(gel); gel += (int64_t)g.rows * g.cols;
        GradSeg gb;  // bias: the ones-column (index ft) of the root rows
        gb.dst = C.b0; gb.rows = fo; gb.cols = 1;
        gb.src = g.src + ft; gb.ld_src = Y.lddw[t]; gb.slab_id = g.slab_id;
        gs.push_back(gb); ges.push_back(gel); gel += fo;
      }
      ps.push_back(s); prs.push_back(prow); prow += s.rows_pad;
      ps.push_back(b); prs.push_back(prow); prow += 1;
    }
  }
  prs.push_back(prow);
  ges.push_back(gel);
  n->n_pack = (int)ps.size(); n->pack_rows = prow;
  n->n_grad = (int)gs.size(); n->grad_elems = gel;
  HMP_HIP(hipMalloc(&n->d_pack_segs, ps.size() * sizeof(PackSeg)));
  HMP_HIP(hipMalloc(&n->d_pack_row_start, prs.size() * sizeof(int64_t)));
  HMP_HIP(hipMalloc(&n->d_grad_segs, gs.size() * sizeof(GradSeg)));
  HMP_HIP(hipMalloc(&n->d_grad_elem_start, ges.size() * sizeof(int64_t)));
  HMP_HIP(hipMemcpy(n->d_pack_segs, ps.data(), ps.size() * sizeof(PackSeg), hipMemcpyHostToDevice));
  HMP_HIP(hipMemcpy(n->d_pack_row_start, prs.data(), prs.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  HMP_HIP(hipMemcpy(n->d_grad_segs, gs.data(), gs.size() * sizeof(GradSeg), hipMemcpyHostToDevice));
  HMP_HIP(hipMemcpy(n->d_grad_elem_start, ges.data(), ges.size() * sizeof(int64_t), hipMemcpyHostToDevice));
  return HMP_OK;
}

// workspace carving: returns bytes; when base != null also assigns pointers ---------------------------
size_t carve(hmp_net* n, char* base, const int32_t* cn, const int64_t* ce) {
  size_t off = 0;
  auto take = [&](size_t bytes) -> char* {
    char* p = base ? base + off : nullptr;
    off += align256(bytes);
    return p;
  };
  const hmp_net_spec& S = n->spec;
  n->d_state = (NetState*)take(sizeof(NetState));
  for (int e = 0; e < n->ET; ++e) {
    const int ns = cn[S.edge_src[e]], nd = cn[S.edge_dst[e]];
    hmp_plan& P = n->plan[e];
    P.d_rowptr = (int32_t*)take((size_t)(nd + 1) * 4);
    P.d_t_rowptr = (int32_t*)take((size_t)(ns + 1) * 4);
    P.d_col = (int32_t*)take((size_t)ce[e] * 4);
    P.d_eid = (int32_t*)take((size_t)ce[e] * 4);
    P.d_t_col = (int32_t*)take((size_t)ce[e] * 4);
    P.d_t_pos = (int32_t*)take((size_t)ce[e] * 4);
    n->plan_scratch[e] = (int*)take(plan_scratch_ints(ce[e], ns, nd) * 4);
  }
  n->d_packed = (float*)take((size_t)n->packed_floats * 4);
  n->d_slabs = (float*)take((size_t)n->slab_floats * 4);
  for (int l = 0; l <= n->L; ++l)
    for (int t = 0; t < n->T; ++t) {
      n->H[l][t] = n->G[l][t] = nullptr;
      if (l >= 1 && n->dim[l][t] > 0) {
        n->H[l][t] = (float*)take((size_t)cn[t] * n->ld[l][t] * 4);
        n->G[l][t] = (float*)take((size_t)cn[t] * n->ld[l][t] * 4);
      }
    }
  for (int l = 0; l < n->L; ++l)
    for (int t = 0; t < n->T; ++t) {
      n->Z[l][t] = n->dZ[l][t] = nullptr;
      if (n->lay[l].ncols[t] > 0) {
        n->Z[l][t] = (float*)take((size_t)cn[t] * n->lay[l].ncols[t] * 4);
        n->dZ[l][t] = (float*)take((size_t)cn[t] * n->lay[l].ncols[t] * 4);
      }
    }
  int cap_out = cn[S.readout_type];
  if (S.pool_edge_type >= 0) cap_out = cn[S.edge_dst[S.pool_edge_type]];
  n->cap_out = cap_out;
  n->d_out = (float*)take((size_t)cap_out * n->out_ld * 4);
  n->d_gout = (float*)take((size_t)cap_out * n->out_ld * 4);
  return off;
}

// profiling helpers ------------------------------------------------------------------------------------
struct Scope {
  hmp_net* n;
  hipStream_t st;
  int idx = -1;
  Scope(hmp_net* n_, int cls, hipStream_t st_) : n(n_), st(st_) {
    if (!n->prof) return;
    ProfRec r;
    r.cls = cls;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    hipEventRecord(r.a, st);
    n->recs.push_back(r);
    idx = (int)n->recs.size() - 1;
  }
  ~Scope() {
    if (idx >= 0) hipEventRecord(n->recs[idx].b, st);
  }
};

enum { KC_PLAN = 0, KC_PACK, KC_GEMM_FWD, KC_AGG_FWD, KC_LOSS, KC_AGG_BWD, KC_GEMM_BWD, KC_GRAD_REDUCE, KC_ADAM, KC_GAT_FWD, KC_GAT_BWD, KC_POOL };

DropCfg make_drop(const hmp_net* n, float p, uint32_t stream) {
  DropCfg d;
  d.k0 = (uint32_t)n->seed; d.k1 = (uint32_t)(n->seed >> 32);
  d.step = n->rng_step; d.stream = stream;
  d.thresh = drop_thresh(p); d.scale = 1.f / (1.f - p);
  d.step_dev = n->step_dev ? &n->d_state->step : nullptr;
  return d;
}

int check_batch(const hmp_net* n, const hmp_batch* b) {
  HMP_CHECK_ARG(n->bound, "net: workspace not bound (call hmp_net_bind_workspace)");
  const hmp_net_spec& S = n->spec;
  for (int t = 0; t < n->T; ++t) {
    HMP_CHECK_ARG(b->n_nodes[t] >= 0 && b->n_nodes[t] <= n->cap_nodes[t], "batch: node type %d has %d nodes, capacity %d", t, b->n_nodes[t], n->cap_nodes[t]);
    if (n->lay[0].ncols[t] > 0 && b->n_nodes[t] > 0)
      HMP_CHECK_ARG(b->d_x[t] != nullptr && b->ldx[t] >= n->dim[0][t], "batch: node type %d features missing or ld < %d", t, n->dim[0][t]);
  }
  for (int e = 0; e < n->ET; ++e) {
    HMP_CHECK_ARG(b->n_edges[e] >= 0 && b->n_edges[e] <= n->cap_edges[e], "batch: edge type %d has %lld edges, capacity %lld", e, (long long)b->n_edges[e], (long long)n->cap_edges[e]);
    HMP_CHECK_ARG(b->n_edges[e] == 0 || b->d_edge_index[e] != nullptr, "batch: edge type %d edge_index missing", e);
  }
  const int n_out_expect = S.pool_edge_type >= 0 ? b->n_nodes[S.edge_dst[S.pool_edge_type]] : b->n_nodes[S.readout_type];
  HMP_CHECK_ARG(b->n_out == n_out_expect, "batch: n_out %d != %d", b->n_out, n_out_expect);
  return HMP_OK;
}

// ---- forward -------------------------------------------------------------------------------------------
int run_plan(hmp_net* n, const hmp_batch* b, hipStream_t st) {
  Scope sc(n, KC_PLAN, st);
  PlanBatch pb;
  memset(&pb, 0, sizeof(pb));
  pb.n = n->ET;
  for (int e = 0; e < n->ET; ++e) {
    PlanJob& J = pb.j[e];
    hmp_plan& P = n->plan[e];
    P.n_src = b->n_nodes[n->spec.edge_src[e]];
    P.n_dst = b->n_nodes[n->spec.edge_dst[e]];
    P.n_edges = b->n_edges[e];
    J.ei = b->d_edge_index[e];
    J.E = P.n_edges; J.n_src = P.n_src; J.n_dst = P.n_dst;
    J.rowptr = P.d_rowptr; J.col = P.d_col; J.eid = P.d_eid;
    J.t_rowptr = P.d_t_rowptr; J.t_col = P.d_t_col; J.t_pos = P.d_t_pos;
    plan_carve(J, n->plan_scratch[e]);
  }
  return plan_launch(pb, &n->d_state->status, st);
}

const float* h_ptr(const hmp_net* n, int l, int t) { return l == 0 ? n->batch.d_x[t] : n->H[l][t]; }
int h_ld(const hmp_net* n, int l, int t) { return l == 0 ? n->batch.ldx[t] : n->ld[l][t]; }

int forward_impl(hmp_net* n, const hmp_batch* b, const float* d_params, hipStream_t st) {
  HMP_TRY(check_batch(n, b));
  n->batch = *b;
  const hmp_net_spec& S = n->spec;
  HMP_TRY(run_plan(n, b, st));
  {
    Scope sc(n, KC_PACK, st);
    HMP_TRY(pack_launch(n->d_pack_segs, n->n_pack, n->pack_rows, n->d_pack_row_start, d_params, n->d_packed, st));
  }
  for (int l = 0; l < n->L; ++l) {
    const hmp_layer_spec& Ls = S.layers[l];
    const LayerLayout& Y = n->lay[l];
    {  // grouped projection
      Scope sc(n, KC_GEMM_FWD, st);
      GemmBatch gb;
      memset(&gb, 0, sizeof(gb));
      for (int s = 0; s < n->T; ++s) {
        if (Y.ncols[s] == 0 || b->n_nodes[s] == 0) continue;
        GemmProblem& p = gb.p[gb.n++];
        p.A = h_ptr(n, l, s); p.lda = h_ld(n, l, s); p.trans_a = 0;
        p.B = n->d_packed + Y.wp_off[s]; p.ldb = Y.ldw[s]; p.trans_b = 1;
        p.C = n->Z[l][s]; p.ldc = Y.ncols[s];
        p.M = b->n_nodes[s]; p.N = Y.ncols[s]; p.K = n->dim[l][s];
        p.n_real = p.N;
        p.epi = EPI_NONE;
      }
      HMP_TRY(gemm_launch(gb, false, 1, st));
    }
    {  // fused aggregation + root + bias + activation + dropout
      Scope sc(n, KC_AGG_FWD, st);
      AggArgs a;
      memset(&a, 0, sizeof(a));
      a.mean = 1;
      for (int t = 0; t < n->T; ++t) {
        if (Y.roff[t] < 0 || b->n_nodes[t] == 0) continue;
        AggDst& D = a.d[a.n++];
        D.n_rows = b->n_nodes[t];
        D.F = fpad(Ls.out_dim[t]);
        D.out = n->H[l + 1][t]; D.ldo = n->ld[l + 1][t];
        D.zroot = n->Z[l][t]; D.ldzr = Y.ncols[t]; D.roff = Y.roff[t];
        D.bias = n->d_packed + Y.bias_off[t];
        D.act = Ls.act;
        D.drop_on = (n->training && Ls.dropout > 0.f) ? 1 : 0;
        if (D.drop_on) D.drop = make_drop(n, Ls.dropout, (uint32_t)(l * HMP_MAX_NODE_TYPES + t));
        for (int i = 0; i < Y.n_live; ++i) {
          const int c = Y.live[i];
          const hmp_conv_spec& C = Ls.convs[c];
          if (C.dst != t) continue;
          if (b->n_nodes[C.src] == 0 || b->n_edges[C.edge_type] == 0) continue;
          AggIn& I = D.in[D.n_in++];
          I.rowptr = n->plan[C.edge_type].d_rowptr;
          I.col = n->plan[C.edge_type].d_col;
          I.z = n->Z[l][C.src]; I.ldz = Y.ncols[C.src]; I.coff = Y.conv[c].coff;
        }
      }
      HMP_TRY(agg_fwd_launch(a, st));
    }
  }
  if (S.pool_edge_type >= 0) {
    Scope sc(n, KC_POOL, st);
    const int rt = S.readout_type;
    HMP_TRY(hmp_segment_mean_fwd(n->H[n->L][rt], n->ld[n->L][rt], n->out_ld, n->plan[S.pool_edge_type], n->d_out, n->out_ld, st));
  }
  n->have_fwd = true;
  return HMP_OK;
}

const float* out_ptr(const hmp_net* n) {
  return n->spec.pool_edge_type >= 0 ? n->d_out : n->H[n->L][n->spec.readout_type];
}

// ---- backward ------------------------------------------------------------------------------------------
int backward_impl(hmp_net* n, const float* d_gout, int ld_gout, float* d_grads, float* const* d_gx, hipStream_t st) {
  HMP_CHECK_ARG(n->have_fwd, "net: backward without a forward");
  HMP_CHECK_ARG((ld_gout & 3) == 0 && ld_gout >= n->out_ld && (reinterpret_cast<uintptr_t>(d_gout) & 15) == 0,
                "net: output gradient must be 16-byte aligned with ld %% 4 == 0 and ld >= %d", n->out_ld);
  const hmp_net_spec& S = n->spec;
  const hmp_batch* b = &n->batch;
  const int rt = S.readout_type;
  const float* gtop = d_gout;
  int ld_gtop = ld_gout;
  if (S.pool_edge_type >= 0) {
    Scope sc(n, KC_POOL, st);
    HMP_TRY(hmp_segment_mean_bwd(d_gout, ld_gout, n->out_ld, n->plan[S.pool_edge_type], n->G[n->L][rt], n->ld[n->L][rt], st));
    gtop = n->G[n->L][rt];
    ld_gtop = n->ld[n->L][rt];
  }
  memset(&n->dyn, 0, sizeof(n->dyn));
  for (int l = n->L - 1; l >= 0; --l) {
    const hmp_layer_spec& Ls = S.layers[l];
    const LayerLayout& Y = n->lay[l];
    auto g_of = [&](int t, int& ldg) -> const float* {
      if (l == n->L - 1 && t == rt) { ldg = ld_gtop; return gtop; }
      ldg = n->ld[l + 1][t];
      return n->G[l + 1][t];
    };
    {  // transposed aggregation: gradient of the projected rows
      Scope sc(n, KC_AGG_BWD, st);
      TAggArgs a;
      memset(&a, 0, sizeof(a));
      a.mean = 1;
      for (int s = 0; s < n->T; ++s) {
        if (Y.ncols[s] == 0 || b->n_nodes[s] == 0) continue;
        TAggSrc& T = a.s[a.n++];
        T.n_rows = b->n_nodes[s];
        T.dz = n->dZ[l][s]; T.lddz = Y.ncols[s]; T.ncols = Y.ncols[s];
        if (Y.roff[s] >= 0) {
          int ldg;
          T.groot = g_of(s, ldg);
          T.ldgr = ldg; T.roff = Y.roff[s]; T.Froot = fpad(Ls.out_dim[s]);
        }
        for (int i = 0; i < Y.n_live; ++i) {
          const int c = Y.live[i];
          const hmp_conv_spec& C = Ls.convs[c];
          if (C.src != s) continue;
          TAggOut& O = T.out[T.n_out++];
          const hmp_plan& P = n->plan[C.edge_type];
          O.t_rowptr = P.d_t_rowptr; O.t_col = P.d_t_col; O.rowptr = P.d_rowptr;
          int ldg;
          O.g = g_of(C.dst, ldg);
          O.ldg = ldg; O.coff = Y.conv[c].coff; O.F = fpad(C.f_out);
        }
      }
      HMP_TRY(agg_bwd_launch(a, st));
    }
    const bool need_dx = (l > 0) || (d_gx != nullptr);
    if (need_dx) {  // input gradient, masked by the previous layer's activation/dropout derivative
      Scope sc(n, KC_GEMM_BWD, st);
      GemmBatch gb;
      memset(&gb, 0, sizeof(gb));
      for (int s = 0; s < n->T; ++s) {
        if (Y.ncols[s] == 0 || b->n_nodes[s] == 0) continue;
        float* dst = l > 0 ? n->G[l][s] : d_gx[s];
        if (!dst) continue;
        GemmProblem& p = gb.p[gb.n++];
        p.A = n->dZ[l][s]; p.lda = Y.ncols[s]; p.trans_a = 0;
        p.B = n->d_packed + Y.wp_off[s]; p.ldb = Y.ldw[s]; p.trans_b = 0;
        p.C = dst; p.ldc = l > 0 ? n->ld[l][s] : b->ldx[s];
        p.M = b->n_nodes[s]; p.N = n->dim[l][s]; p.K = Y.ncols[s];
        p.n_real = p.N;
        if (l > 0) {
          const hmp_layer_spec& Lp = S.layers[l - 1];
          p.epi = EPI_ACTMASK;
          p.H = n->H[l][s]; p.ldh = n->ld[l][s]; p.act = Lp.act;
          p.drop_on = (n->training && Lp.dropout > 0.f) ? 1 : 0;
          if (p.drop_on) p.drop = make_drop(n, Lp.dropout, (uint32_t)((l - 1) * HMP_MAX_NODE_TYPES + s));
          if (p.act == HMP_ACT_NONE && !p.drop_on) p.epi = EPI_NONE;
        }
      }
      HMP_TRY(gemm_launch(gb, false, 1, st));
    }
    {  // weight + bias gradient: dWp = dZ^T * [H | 1], split over node chunks
      Scope sc(n, KC_GEMM_BWD, st);
      GemmBatch gb;
      memset(&gb, 0, sizeof(gb));
      int ids[GEMM_MAX_PROB];
      for (int s = 0; s < n->T; ++s) {
        if (Y.ncols[s] == 0) continue;
        const int sid = l * HMP_MAX_NODE_TYPES + s;
        n->dyn.n_slabs[sid] = 0;
        n->dyn.slab_stride[sid] = (int64_t)Y.ncols[s] * Y.lddw[s];
        if (b->n_nodes[s] == 0) continue;  // no nodes: zero gradient (n_slabs = 0)
        ids[gb.n] = sid;
        GemmProblem& p = gb.p[gb.n++];
        p.A = n->dZ[l][s]; p.lda = Y.ncols[s]; p.trans_a = 1;
        p.B = h_ptr(n, l, s); p.ldb = h_ld(n, l, s); p.trans_b = 0;
        p.C = n->d_slabs + Y.slab_off[s]; p.ldc = Y.lddw[s];
        p.slab_stride = (int64_t)Y.ncols[s] * Y.lddw[s];
        p.M = Y.ncols[s]; p.N = n->dim[l][s] + 1; p.K = b->n_nodes[s];
        p.n_real = n->dim[l][s]; p.aug_ones = 1;
        p.epi = EPI_NONE;
      }
      HMP_TRY(gemm_launch(gb, true, MAX_SLABS, st));
      for (int i = 0; i < gb.n; ++i) n->dyn.n_slabs[ids[i]] = gb.p[i].ksplit;
    }
  }
  {
    Scope sc(n, KC_GRAD_REDUCE, st);
    HMP_TRY(grad_reduce_launch(n->d_grad_segs, n->n_grad, n->grad_elems, n->d_grad_elem_start, n->dyn, n->d_slabs, d_grads, st));
  }
  return HMP_OK;
}

}  // namespace

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" int hmp_net_create(const hmp_net_spec* spec, hmp_net** out) {
  HMP_CHECK_ARG(spec && out, "hmp_net_create: null argument");
  HMP_CHECK_ARG(hmp_device_count() > 0, "hmp_net_create: no gfx950 device visible");
  hmp_net* n = new hmp_net;
  n->spec = *spec;
  int r = build_layout(n);
  if (r == HMP_OK) r = build_tables(n);
  if (r != HMP_OK) {
    hmp_net_destroy(n);
    return r;
  }
  *out = n;
  return HMP_OK;
}

extern "C" void hmp_net_destroy(hmp_net* n) {
  if (!n) return;
  for (auto& r : n->recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
  if (n->d_pack_segs) hipFree(n->d_pack_segs);
  if (n->d_pack_row_start) hipFree(n->d_pack_row_start);
  if (n->d_grad_segs) hipFree(n->d_grad_segs);
  if (n->d_grad_elem_start) hipFree(n->d_grad_elem_start);
  delete n;
}

extern "C" size_t hmp_net_workspace_bytes(const hmp_net* net, const int32_t* cap_nodes, const int64_t* cap_edges) {
  if (!net || !cap_nodes || !cap_edges) return 0;
  hmp_net tmp = *net;  // carve() writes pointer fields; keep the real object untouched
  tmp.recs.clear();
  return carve(&tmp, nullptr, cap_nodes, cap_edges);
}

extern "C" int hmp_net_bind_workspace(hmp_net* n, void* d_workspace, size_t bytes, const int32_t* cap_nodes, const int64_t* cap_edges) {
  HMP_CHECK_ARG(n && d_workspace && cap_nodes && cap_edges, "hmp_net_bind_workspace: null argument");
  HMP_CHECK_ARG((reinterpret_cast<uintptr_t>(d_workspace) & 255) == 0, "hmp_net_bind_workspace: workspace must be 256-byte aligned");
  const size_t need = carve(n, (char*)d_workspace, cap_nodes, cap_edges);
  HMP_CHECK_ARG(bytes >= need, "hmp_net_bind_workspace: %zu bytes given, %zu needed", bytes, need);
  for (int t = 0; t < n->T; ++t) n->cap_nodes[t] = cap_nodes[t];
  for (int e = 0; e < n->ET; ++e) n->cap_edges[e] = cap_edges[e];
  // zero once: padding columns of every buffer stay zero for the lifetime of the binding
  HMP_HIP(hipMemset(d_workspace, 0, need));
  n->bound = true;
  n->have_fwd = false;
  return HMP_OK;
}

extern "C" int hmp_net_forward(hmp_net* n, const hmp_batch* batch, const float* d_params, int32_t training, uint64_t seed,
                               uint32_t rng_step, const float** d_out, int32_t* ld_out, void* stream) {
  HMP_CHECK_ARG(n && batch && d_params && d_out && ld_out, "hmp_net_forward: null argument");
  n->training = training; n->seed = seed; n->rng_step = rng_step; n->step_dev = false;
  HMP_TRY(forward_impl(n, batch, d_params, (hipStream_t)stream));
  *d_out = out_ptr(n);
  *ld_out = n->out_ld;
  return HMP_OK;
}

extern "C" int hmp_net_backward(hmp_net* n, const float* d_gout, int32_t ld_gout, const float* d_params, float* d_grads,
                                float* const* d_gx, void* stream) {
  HMP_CHECK_ARG(n && d_gout && d_grads && d_params, "hmp_net_backward: null argument");
  return backward_impl(n, d_gout, ld_gout, d_grads, d_gx, (hipStream_t)stream);
}

extern "C" int hmp_net_step_fwd_bwd(hmp_net* n, const hmp_batch* batch, const float* d_params, float* d_grads,
                                    const hmp_train_args* args, void* stream) {
  HMP_CHECK_ARG(n && batch && d_params && d_grads && args, "hmp_net_step_fwd_bwd: null argument");
  HMP_CHECK_ARG(batch->d_labels != nullptr, "hmp_net_step_fwd_bwd: labels required");
  hipStream_t st = (hipStream_t)stream;
  n->training = args->training; n->seed = args->seed; n->rng_step = 0; n->step_dev = true;
  HMP_TRY(forward_impl(n, batch, d_params, st));
  const int64_t na = n->spec.n_active_params;
  {
    Scope sc(n, KC_LOSS, st);
    HMP_TRY(masked_ce_launch(out_ptr(n), n->out_ld, batch->n_out, n->out_dim, batch->d_labels, args->ignored_label, n->d_gout,
                             n->out_ld, d_grads + na, n->d_state, st));
  }
  return backward_impl(n, n->d_gout, n->out_ld, d_grads, nullptr, st);
}

extern "C" int hmp_net_step_adam(hmp_net* n, float* d_params, const float* d_grads, float* d_m, float* d_v,
                                 const hmp_train_args* args, void* stream) {
  HMP_CHECK_ARG(n && d_params && d_grads && d_m && d_v && args, "hmp_net_step_adam: null argument");
  HMP_CHECK_ARG(n->bound, "hmp_net_step_adam: workspace not bound");
  hipStream_t st = (hipStream_t)stream;
  const int64_t na = n->spec.n_active_params;
  Scope sc(n, KC_ADAM, st);
  HMP_TRY(adam_launch(d_params, d_grads, d_m, d_v, na, args->lr, args->beta1, args->beta2, args->eps, args->weight_decay, 0,
                      &n->d_state->step, d_grads + na + 1, st));
  return step_increment_launch(n->d_state, st);
}

extern "C" int hmp_net_read_state(hmp_net* n, int32_t* step, int32_t* status, void* stream) {
  HMP_CHECK_ARG(n && n->bound, "hmp_net_read_state: net not bound");
  NetState h;
  HMP_HIP(hipStreamSynchronize((hipStream_t)stream));
  HMP_HIP(hipMemcpy(&h, n->d_state, sizeof(h), hipMemcpyDeviceToHost));
  if (step) *step = h.step;
  if (status) *status = h.status;
  return HMP_OK;
}

extern "C" int hmp_net_profile(hmp_net* n, int32_t enable) {
  HMP_CHECK_ARG(n, "hmp_net_profile: null net");
  n->prof = enable != 0;
  return HMP_OK;
}

extern "C" int hmp_net_profile_read(hmp_net* n, float* ms_sum, int32_t* launches) {
  HMP_CHECK_ARG(n && ms_sum && launches, "hmp_net_profile_read: null argument");
  for (int i = 0; i < HMP_N_KCLASS; ++i) { ms_sum[i] = 0.f; launches[i] = 0; }
  for (auto& r : n->recs) {
    HMP_HIP(hipEventSynchronize(r.b));
    float ms = 0.f;
    HMP_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    if (r.cls >= 0 && r.cls < HMP_N_KCLASS) { ms_sum[r.cls] += ms; launches[r.cls] += 1; }
    hipEventDestroy(r.a);
    hipEventDestroy(r.b);
  }
  n->recs.clear();
  return HMP_OK;
}
