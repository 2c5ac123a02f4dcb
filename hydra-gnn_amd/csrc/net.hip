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
//
// GAT algebra: the attention logits a_s = <lin_src(x), att_src> = x * V_src with V_src = fold(att_src, W_src)
// (and likewise a_d, and the edge term <lin_edge(e), att_edge> = e * V_edge), so the stacked operand of a source
// type also carries the H rows of V_src^T (and of V_dst^T for the destination role): ONE projection per node
// type yields h_s, a_s and a_d; the chain rule back to lin/att runs in the gradient un-pack kernel.
#include <cstdlib>
#include <vector>

#include "plan_small.h"

using namespace hmp;

namespace {

struct ConvLayout {
  int coff;               // SAGE: column of W_l * x in Z[l][src];  GAT: column of h_s (H*Cp wide)
  int asoff, adoff;       // GAT: columns of a_s in Z[l][src] / a_d in Z[l][dst]
  int H, C, Cp;
  int64_t vedge_off;      // GAT_edge: packed V_edge [GAT_MAX_EDIM][GAT_HMAX]
  int64_t vslab_off;      // GAT_edge: slabs of d V_edge
  bool wd_is_src;         // GAT: the destination role uses lin_src's weights
  // per-conv workspace (GAT)
  float *smax, *sden, *alpha_drop, *dlogit, *dlogit_orig;
  // SAGE, aggregate-first (hmp_conv_spec::agg_first): M = segment mean of the SOURCE rows [n_dst][ld_m] (fp32), projected by
  // W_l [f_out][ldw(src)] at wc_off into columns [aoff, aoff + f_out) of Z[l][dst]; dM its gradient; weight-gradient slabs at cslab_off
  bool agg_first;
  int aoff, ld_m;
  int64_t wc_off, cslab_off;
  float *mrows, *dmrows;
};

struct LayerLayout {
  int kind;
  int n_live;
  int live[HMP_MAX_CONVS];            // indices of live convs
  ConvLayout conv[HMP_MAX_CONVS];
  int n_in[HMP_MAX_NODE_TYPES];       // live convs reaching the node type
  bool reads[HMP_MAX_NODE_TYPES];     // the layer reads this node type's state (ncols > 0, or the source of an aggregate-first conv)
  int aggf_of_src[HMP_MAX_NODE_TYPES];  // the aggregate-first conv whose SOURCE is this node type, -1: none
  int ncols[HMP_MAX_NODE_TYPES];      // width of Z[l][s] (0: type is not read by this layer)
  int roff[HMP_MAX_NODE_TYPES];       // SAGE root segment column in Z[l][t], -1 if none
  int64_t wp_off[HMP_MAX_NODE_TYPES]; // packed weights [ncols][ldw]
  int ldw[HMP_MAX_NODE_TYPES];
  int64_t bias_off[HMP_MAX_NODE_TYPES];
  int64_t slab_off[HMP_MAX_NODE_TYPES];   // packed-gradient slabs [max_slabs][ncols][lddw]
  int lddw[HMP_MAX_NODE_TYPES];
  int64_t bslab_off[HMP_MAX_NODE_TYPES];  // GAT: slabs of the bias column sums [max_slabs][out_dim][4]
  GatLayerS* d_gat;                       // device table (GAT layers)
  GatLayerS h_gat;
};

constexpr int MAX_SLABS = 192;       // split-K slabs per weight-gradient problem (large batches: 256x256 tiles x ~170 node chunks)
static_assert(MAX_SLABS >= GEMM_TALL_SLABS, "the slab buffer must hold what the tall weight-gradient kernel may write");
constexpr int TN_DIRECT_SLABS = 16;  // small batches: more slabs only lengthen the gradient un-pack

struct ProfRec {
  int cls;
  hipEvent_t a, b;
};

}  // namespace

// Test / A-B switches of the executor (environment variables), read ONCE per entry call -- forward, step -- into this block
// (tests flip them between two calls on one net) instead of by getenv() wherever a decision is taken
struct EnvSwitches {
  bool plan_sliced = true;  // HMP_PLAN_SLICED=0: plan parts read the whole edge list (tests compare both builds)
  bool front = true;        // HMP_FRONT=0: separate pack / projection / plan launches
  bool ell = false;         // HMP_ELL=1 (experiment builds only): neighbour ids from the plan's ELL tables
  bool z16 = true;          // HMP_Z16=0: keep fp32 projected rows / gradients in bf16 compute mode
  bool h16 = true;          // HMP_H16=0: keep fp32 activations in bf16 compute mode
  bool rootcopy = false;    // HMP_ROOTCOPY=1: the transposed aggregation copies the root block of dZ
  bool tn_direct = true;    // HMP_TN=0: LDS-staged split-K kernel for the small-batch weight gradients
  bool bf16_all = false;    // HMP_BF16_ALL=1: every GEMM of a bf16-mode net takes the bf16 kernel whatever its size
};

struct hmp_net {
  hmp_net_spec spec;
  EnvSwitches env;
  int T, ET, L;
  int dim[HMP_MAX_LAYERS + 1][HMP_MAX_NODE_TYPES];  // feature width of H[l][t]
  int ld[HMP_MAX_LAYERS + 1][HMP_MAX_NODE_TYPES];   // leading dimension of our own H/G buffers (l >= 1)
  LayerLayout lay[HMP_MAX_LAYERS];
  int64_t packed_floats, slab_floats;
  int out_dim, out_ld;
  bool any_gat = false;
  bool pass0[HMP_MAX_NODE_TYPES] = {false};  // layer 1 reads the INPUT features of these types (pre_mp passthrough)

  // static device tables (owned)
  PackSeg* d_pack_segs = nullptr;
  int n_pack = 0;
  int64_t pack_rows = 0;
  GradSeg* d_grad_segs = nullptr;
  int n_grad = 0;
  int64_t grad_elems = 0;
  int max_pack_rows = 0;
  int64_t max_grad_elems = 0;
  SegBlocks pack_sb, grad_sb;
  int2* d_pack_map = nullptr;  // front kernel: pack block (16 items) -> {segment, first item}
  int n_pack_blocks16 = 0;
  char* ws_base = nullptr;
  size_t ws_bytes = 0;
  float* degf[HMP_MAX_EDGE_TYPES];  // 1 / max(in-degree,1) per destination node, by-product of the plan
  int* ell[HMP_MAX_EDGE_TYPES];     // [n_dst][ELL_W] / [n_src][ELL_W] first neighbour ids per row (kernels.h), by-product of
  int* t_ell[HMP_MAX_EDGE_TYPES];   // the single-launch plan build only: valid while ell_ok
  bool ell_ok = false;
  bool ell_on = false;              // this call: HMP_ELL=1
  bool any_agg_first = false;       // some conv is evaluated aggregate-first (large batches only: never with the small-batch sequence)
  int* d_iota = nullptr;            // 0, 1, 2, ..: row extents AND ids of the identity lists that hand an aggregate-first block to
  float* d_ones = nullptr;          // the aggregation kernels (in-degree 1, weight 1)
  float* d_sadd = nullptr;          // fallback of the gather-add epilogue: the source rows' gradient term as an fp32 matrix (GemmProblem::Cadd)
  int iota_cap = 0;
  int64_t sadd_floats = 0;
  float* d_row_lv = nullptr;        // per output row {loss, valid} of the fused step's loss kernel
  bool fin_loss = false;            // the next gradient un-pack also finalises {loss_sum, count}

  // workspace binding
  bool bound = false;
  int cap_nodes[HMP_MAX_NODE_TYPES];
  int64_t cap_edges[HMP_MAX_EDGE_TYPES];
  NetState* d_state = nullptr;  // owned by the net (NOT part of the re-bindable workspace): status bits and the net's own step counter
  int* d_step = nullptr;        // the step counter of the running step: the optimiser's (hmp_train_args::d_step) or &d_state->step
  hmp_plan plan[HMP_MAX_EDGE_TYPES];
  int* plan_scratch[HMP_MAX_EDGE_TYPES];
  float* d_packed = nullptr;
  float* d_slabs = nullptr;
  float* H[HMP_MAX_LAYERS + 1][HMP_MAX_NODE_TYPES];
  float* G[HMP_MAX_LAYERS + 1][HMP_MAX_NODE_TYPES];
  bool h16[HMP_MAX_LAYERS + 1][HMP_MAX_NODE_TYPES] = {};  // last forward: H[l][t] was written as bf16 (see forward_impl)
  float* Z[HMP_MAX_LAYERS][HMP_MAX_NODE_TYPES];
  float* dZ[HMP_MAX_LAYERS][HMP_MAX_NODE_TYPES];
  float* d_out = nullptr;   // pooled output [cap_out, out_ld] (only with pool_edge_type)
  float* d_gout = nullptr;  // loss gradient [cap_out, out_ld]
  int cap_out = 0;

  // last forward
  bool have_fwd = false;
  bool plan_ok = false;  // the plan arrays hold the topology of the last batch (cleared by a workspace re-bind)
  hmp_batch batch;
  int training = 0;
  uint64_t seed = 0;
  uint32_t rng_step = 0;
  bool step_dev = false;  // dropout step offset read from *d_step
  GradReduceDyn dyn;

  // parallel branches (side streams; under capture they become branches of the hipGraph)
  bool use_branches = false;
  int branch_mask = 0;     // HMP_BRANCH bits: 1 = pack + layer-0 projection next to the plan, 2 = weight gradients next to the backward chain
  bool dw_branch = false;  // weight-gradient GEMMs per layer on a side stream instead of one merged launch
  int dw_mode = -1;        // HMP_DW_BRANCH override (0 / 1), -1 = automatic
  int compute_bf16 = 0;    // hmp_net_set_compute: large grouped GEMMs on the bf16 matrix pipe (fp32 storage and accumulation)
  int fuse_mode = -1;      // HMP_FUSE override (0 / 1), -1 = automatic: row-local GEMMs ride in the aggregation kernels
  bool fuse_now = false;   // decision for the current batch
  bool reuse_plan = false; // this call: hmp_batch::plan_valid accepted
  // fused training step: labels for the cross entropy riding in the last aggregation, Adam riding in the gradient un-pack
  const int64_t* ce_labels = nullptr;
  int64_t ce_ignored = 0;
  bool ce_done = false;
  AdamFuse adam_fuse = {};
  bool adam_done = false;
  hipStream_t side[2] = {nullptr, nullptr};
  hipEvent_t evs[32];
  int n_evs = 0, ev_i = 0;

  // graph-local chain (aggregate.hip: chain_kernel): the aggregation launches of a fused training step are DEFERRED while they
  // qualify; at the weight-gradient point they run as one launch (or, if anything in between did not qualify, in order as before)
  struct Deferred {
    int kind;  // 0 agg_proj_fwd, 1 agg_fwd, 2 agg_bwd_dx, 3 agg_bwd
    AggArgs fa;
    TAggArgs ba;
  };
  bool chain_try = false;        // this step: defer
  std::vector<Deferred> deferred;
  int chain_mode = -1;           // HMP_CHAIN override (0 / 1), -1 = automatic
  ChainArgs* chain_h = nullptr;  // host copy of what the device holds (pinned)
  ChainArgs* chain_stage = nullptr;
  ChainArgs* d_chain = nullptr;
  hipEvent_t chain_copied = nullptr;
  bool chain_valid = false;

  // profiling
  bool prof = false;
  std::vector<ProfRec> recs;
};

namespace {

void read_env(hmp_net* n) {
  auto is = [](const char* name, char c) { const char* v = getenv(name); return v && v[0] == c; };
  EnvSwitches e;
  e.plan_sliced = !is("HMP_PLAN_SLICED", '0');
  e.front = !is("HMP_FRONT", '0');
#ifdef HMP_EXPERIMENTS
  e.ell = is("HMP_ELL", '1');
#endif
  e.z16 = !is("HMP_Z16", '0');
  e.h16 = !is("HMP_H16", '0');
  e.rootcopy = is("HMP_ROOTCOPY", '1');
  e.tn_direct = !is("HMP_TN", '0');
  e.bf16_all = is("HMP_BF16_ALL", '1');
  n->env = e;
}

inline int fpad(int f) { return align4(f); }

int build_layout(hmp_net* n) {
  const hmp_net_spec& S = n->spec;
  n->T = S.n_node_types; n->ET = S.n_edge_types; n->L = S.n_layers;
  HMP_CHECK_ARG(n->T >= 1 && n->T <= HMP_MAX_NODE_TYPES, "net: n_node_types %d", n->T);
  HMP_CHECK_ARG(n->ET >= 1 && n->ET <= HMP_MAX_EDGE_TYPES, "net: n_edge_types %d", n->ET);
  HMP_CHECK_ARG(n->L >= 1 && n->L <= HMP_MAX_LAYERS, "net: n_layers %d", n->L);
  HMP_CHECK_ARG(S.readout_type >= 0 && S.readout_type < n->T, "net: readout_type");
  HMP_CHECK_ARG(S.pool_edge_type >= -1 && S.pool_edge_type < n->ET, "net: pool_edge_type");
  HMP_CHECK_ARG(S.aux_readout_type >= -1 && S.aux_readout_type < n->T && S.aux_readout_type != S.readout_type &&
                    (S.aux_readout_type < 0 || S.pool_edge_type < 0),
                "net: aux_readout_type (a second node type, not with a pooled output)");
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
    Y.kind = Ls.convs[0].kind;
    Y.d_gat = nullptr;
    HMP_CHECK_ARG(Y.kind == HMP_CONV_SAGE || Y.kind == HMP_CONV_GAT, "net: layer %d: unknown conv kind %d", l, Y.kind);
    HMP_CHECK_ARG(Ls.group_mean == 0 || Y.kind == HMP_CONV_GAT, "net: HeteroConv aggr=mean is only supported for GAT layers");
    for (int t = 0; t < n->T; ++t) {
      n->dim[l + 1][t] = Ls.out_dim[t];
      n->ld[l + 1][t] = fpad(Ls.out_dim[t]);
      if (Ls.passthrough[t]) {
        HMP_CHECK_ARG(l == 0 && n->L > 1, "net: passthrough node types are supported on the first layer only");
        HMP_CHECK_ARG(Ls.out_dim[t] == S.in_dim[t], "net: passthrough type %d must keep its input width", t);
        n->pass0[t] = true;
      }
      Y.ncols[t] = 0;
      Y.roff[t] = -1;
      Y.n_in[t] = 0;
      Y.reads[t] = false;
      Y.aggf_of_src[t] = -1;
    }
    Y.n_live = 0;
    int n_outgoing[HMP_MAX_NODE_TYPES] = {0};
    for (int c = 0; c < Ls.n_convs; ++c) {
      const hmp_conv_spec& C = Ls.convs[c];
      ConvLayout& Q = Y.conv[c];
      memset(&Q, 0, sizeof(Q));
      HMP_CHECK_ARG(C.kind == Y.kind, "net: layer %d mixes conv kinds", l);
      HMP_CHECK_ARG(C.edge_type >= 0 && C.edge_type < n->ET, "net: conv edge_type");
      HMP_CHECK_ARG(C.src == S.edge_src[C.edge_type] && C.dst == S.edge_dst[C.edge_type], "net: conv endpoints disagree with edge type");
      HMP_CHECK_ARG(n->dim[l][C.src] > 0 && n->dim[l][C.dst] > 0, "net: layer %d conv %d reads a node type with no features", l, c);
      if (Y.kind == HMP_CONV_SAGE) {
        HMP_CHECK_ARG(C.f_out == Ls.out_dim[C.dst], "net: SAGE f_out must equal the layer's out_dim of the destination type");
      } else {
        HMP_CHECK_ARG(C.heads >= 1 && C.heads <= GAT_HMAX && C.f_out >= 1 && C.f_out <= 256, "net: GAT heads %d / channels %d unsupported", C.heads, C.f_out);
        HMP_CHECK_ARG(Ls.out_dim[C.dst] == (C.concat ? C.heads * C.f_out : C.f_out), "net: GAT out_dim mismatch on layer %d conv %d", l, c);
        HMP_CHECK_ARG(C.edge_dim >= 0 && C.edge_dim <= GAT_MAX_EDIM, "net: GAT edge_dim %d > %d", C.edge_dim, GAT_MAX_EDIM);
        HMP_CHECK_ARG(!(C.edge_dim > 0 && C.fill_mean && C.self_loops), "net: fill_value='mean' with edge attributes is not supported (the reference passes zeros)");
        HMP_CHECK_ARG(!C.self_loops || C.src == C.dst, "net: self loops need src == dst (the reference sets add_self_loops = (src == dst))");
      }
      HMP_CHECK_ARG(!C.agg_first || (Y.kind == HMP_CONV_SAGE && C.src != C.dst), "net: agg_first is for SAGE convs between two node types");
      if (!C.active) continue;
      Y.live[Y.n_live++] = c;
      ++Y.n_in[C.dst];
      const bool aggf = C.agg_first != 0;
      ++n_outgoing[aggf ? C.dst : C.src];  // an aggregate-first block is gathered (identity lists) from the DESTINATION type's own Z
      HMP_CHECK_ARG(Y.n_in[C.dst] <= AGG_MAX_IN && n_outgoing[C.src] <= AGG_MAX_IN && n_outgoing[C.dst] <= AGG_MAX_IN,
                    "net: more than %d convs share a node type", AGG_MAX_IN);
      Y.reads[C.src] = Y.reads[C.dst] = true;
      if (Y.kind == HMP_CONV_SAGE) {
        HMP_CHECK_ARG(C.w0 >= 0 && C.b0 >= 0 && C.w1 >= 0, "net: SAGE conv needs lin_l.weight, lin_l.bias, lin_r.weight");
        if (aggf) {
          HMP_CHECK_ARG(Y.aggf_of_src[C.src] < 0, "net: layer %d has two aggregate-first convs from node type %d (at most one)", l, C.src);
          Y.aggf_of_src[C.src] = c;
          Q.agg_first = true;
          Q.ld_m = fpad(n->dim[l][C.src]);
          n->any_agg_first = true;
        } else {
          Q.coff = Y.ncols[C.src];
          Y.ncols[C.src] += fpad(C.f_out);
        }
      } else {
        HMP_CHECK_ARG(C.w0 >= 0 && C.a0 >= 0 && C.a1 >= 0 && C.b0 >= 0, "net: GAT conv needs lin_src, att_src, att_dst, bias");
        HMP_CHECK_ARG(C.edge_dim == 0 || (C.w2 >= 0 && C.a2 >= 0), "net: GAT_edge conv needs lin_edge, att_edge");
        Q.H = C.heads; Q.C = C.f_out; Q.Cp = fpad(C.f_out);
        Q.wd_is_src = (C.src == C.dst) || (C.w1 == C.w0) || (C.w1 < 0);
        HMP_CHECK_ARG(Q.wd_is_src ? (n->dim[l][C.src] == n->dim[l][C.dst]) : true, "net: shared lin with different widths");
        Q.coff = Y.ncols[C.src];
        Y.ncols[C.src] += Q.H * Q.Cp;
        Q.asoff = Y.ncols[C.src];
        Y.ncols[C.src] += fpad(Q.H);
        n->any_gat = true;
      }
    }
    HMP_CHECK_ARG(Y.n_live > 0, "net: layer %d has no live conv", l);
    if (Y.kind == HMP_CONV_SAGE) {
      // column order of Z[l][t]: [blocks of the convs sourced from t] [aggregate-first blocks of convs that END in t] [root]:
      // the root block stays last, where the backward GEMMs can take it from the output gradient in place
      for (int i = 0; i < Y.n_live; ++i) {
        const int c = Y.live[i];
        const hmp_conv_spec& C = Ls.convs[c];
        if (!Y.conv[c].agg_first) continue;
        Y.conv[c].aoff = Y.ncols[C.dst];
        Y.ncols[C.dst] += fpad(C.f_out);
      }
      for (int t = 0; t < n->T; ++t)
        if (Y.n_in[t] > 0) {
          Y.roff[t] = Y.ncols[t];
          Y.ncols[t] += fpad(Ls.out_dim[t]);
        }
    } else {
      for (int i = 0; i < Y.n_live; ++i) {
        const int c = Y.live[i];
        const hmp_conv_spec& C = Ls.convs[c];
        Y.conv[c].adoff = Y.ncols[C.dst];
        Y.ncols[C.dst] += fpad(C.heads);
      }
    }
    for (int t = 0; t < n->T; ++t) {
      Y.ldw[t] = fpad(n->dim[l][t]);
      Y.lddw[t] = fpad(n->dim[l][t] + 1);
      Y.wp_off[t] = packed;
      packed += (int64_t)Y.ncols[t] * Y.ldw[t];
      Y.bias_off[t] = packed;
      packed += (Y.n_in[t] > 0) ? fpad(Ls.out_dim[t]) : 0;
      Y.slab_off[t] = slabs;
      slabs += (int64_t)MAX_SLABS * Y.ncols[t] * Y.lddw[t];
      Y.bslab_off[t] = slabs;
      if (Y.kind == HMP_CONV_GAT && Y.n_in[t] > 0) slabs += (int64_t)MAX_SLABS * fpad(Ls.out_dim[t]) * 4;
    }
    for (int i = 0; i < Y.n_live; ++i) {  // aggregate-first convs: their W_l packed on its own, their weight-gradient slabs
      const int c = Y.live[i];
      if (!Y.conv[c].agg_first) continue;
      const hmp_conv_spec& C = Ls.convs[c];
      Y.conv[c].wc_off = packed;
      packed += (int64_t)fpad(C.f_out) * Y.ldw[C.src];
      Y.conv[c].cslab_off = slabs;
      slabs += (int64_t)MAX_SLABS * fpad(C.f_out) * Y.lddw[C.src];
    }
    if (Y.kind == HMP_CONV_GAT) {
      for (int i = 0; i < Y.n_live; ++i) {
        const int c = Y.live[i];
        if (Ls.convs[c].edge_dim > 0) {
          Y.conv[c].vedge_off = packed;
          packed += GAT_MAX_EDIM * GAT_HMAX;
          Y.conv[c].vslab_off = slabs;
          slabs += (int64_t)MAX_SLABS * GAT_MAX_EDIM * GAT_HMAX;
        }
      }
    }
  }
  // liveness must be closed: whatever a live conv produces is consumed by the next layer / the readout
  for (int l = 0; l < n->L; ++l) {
    const LayerLayout& Y = n->lay[l];
    for (int i = 0; i < Y.n_live; ++i) {
      const hmp_conv_spec& C = S.layers[l].convs[Y.live[i]];
      if (l == n->L - 1)
        HMP_CHECK_ARG(C.dst == S.readout_type || C.dst == S.aux_readout_type,
                      "net: last-layer conv %d is active but does not feed a readout type", Y.live[i]);
      else
        HMP_CHECK_ARG(n->lay[l + 1].reads[C.dst], "net: layer %d conv %d is active but layer %d never reads its output", l, Y.live[i], l + 1);
    }
  }
  n->packed_floats = packed;
  n->slab_floats = slabs;
  n->out_dim = n->dim[n->L][S.readout_type];
  n->out_ld = fpad(n->out_dim);
  HMP_CHECK_ARG(n->out_dim > 0, "net: readout type has no output in the last layer");
  HMP_CHECK_ARG(S.aux_readout_type < 0 || n->dim[n->L][S.aux_readout_type] > 0, "net: second readout type has no output in the last layer");
  return HMP_OK;
}

inline int slab_id_w(int l, int t) { return l * SLAB_IDS_PER_LAYER + t; }
inline int slab_id_b(int l, int t) { return l * SLAB_IDS_PER_LAYER + HMP_MAX_NODE_TYPES + t; }
inline int slab_id_v(int l, int c) { return l * SLAB_IDS_PER_LAYER + 2 * HMP_MAX_NODE_TYPES + c; }

GradTerm term(int kind, int slab_id, int64_t src, int ld, int C, int Cp, int inner = 0, int64_t att = 0, int64_t w = 0, int ldw = 0,
              float scale = 1.f) {
  GradTerm t;
  memset(&t, 0, sizeof(t));
  t.kind = kind; t.slab_id = slab_id; t.src = src; t.ld = ld; t.C = C; t.Cp = Cp; t.inner = inner;
  t.att = att; t.w = w; t.ldw = ldw; t.scale = scale;
  return t;
}

// static pack / grad tables ------------------------------------------------------------------------
int build_tables(hmp_net* n) {
  const hmp_net_spec& S = n->spec;
  std::vector<PackSeg> ps;
  std::vector<int64_t> prs;
  std::vector<GradSeg> gs;
  std::vector<int64_t> ges;
  int64_t prow = 0, gel = 0;
  auto push_pack = [&](const PackSeg& s) {
    ps.push_back(s); prs.push_back(prow); prow += s.rows_pad;
    if (s.rows_pad > n->max_pack_rows) n->max_pack_rows = s.rows_pad;
  };
  auto push_grad = [&](GradSeg g) {
    gs.push_back(g); ges.push_back(gel); gel += (int64_t)g.rows * g.cols;
    if ((int64_t)g.rows * g.cols > n->max_grad_elems) n->max_grad_elems = (int64_t)g.rows * g.cols;
  };
  for (int l = 0; l < n->L; ++l) {
    const hmp_layer_spec& Ls = S.layers[l];
    const LayerLayout& Y = n->lay[l];
    const float gscale = Ls.group_mean ? 1.f : 1.f;  // weights see the scale through the output gradient; bias below
    (void)gscale;
    for (int i = 0; i < Y.n_live; ++i) {
      const int c = Y.live[i];
      const hmp_conv_spec& C = Ls.convs[c];
      const ConvLayout& Q = Y.conv[c];
      const int fs = n->dim[l][C.src], ft = n->dim[l][C.dst];
      PackSeg s;
      memset(&s, 0, sizeof(s));
      if (Y.kind == HMP_CONV_SAGE) {
        s.kind = PACK_SUM;
        s.dst = Q.agg_first ? Q.wc_off : Y.wp_off[C.src] + (int64_t)Q.coff * Y.ldw[C.src];
        s.rows = C.f_out; s.rows_pad = fpad(C.f_out); s.cols = fs; s.ld_dst = Y.ldw[C.src]; s.ld_src = fs;
        s.nsrc = 1; s.src[0] = C.w0;
        push_pack(s);
        GradSeg g;
        memset(&g, 0, sizeof(g));
        g.dst = C.w0; g.rows = C.f_out; g.cols = fs; g.n_terms = 1;
        if (Q.agg_first) g.t[0] = term(GT_COPY, slab_id_v(l, c), Q.cslab_off, Y.lddw[C.src], C.f_out, C.f_out);
        else g.t[0] = term(GT_COPY, slab_id_w(l, C.src), Y.slab_off[C.src] + (int64_t)Q.coff * Y.lddw[C.src], Y.lddw[C.src], C.f_out, C.f_out);
        push_grad(g);
        continue;
      }
      // ---- GAT conv -------------------------------------------------------------------------------
      const int HC = Q.H * Q.C;
      const int64_t wd = Q.wd_is_src ? C.w0 : C.w1;
      // h_s rows (heads padded to Cp)
      s.kind = PACK_HEADS; s.H = Q.H; s.C = Q.C; s.Cp = Q.Cp;
      s.dst = Y.wp_off[C.src] + (int64_t)Q.coff * Y.ldw[C.src];
      s.rows = Q.H * Q.Cp; s.rows_pad = Q.H * Q.Cp; s.cols = fs; s.ld_dst = Y.ldw[C.src]; s.ld_src = fs;
      s.nsrc = 1; s.src[0] = C.w0;
      push_pack(s);
      // V_src^T rows
      s.kind = PACK_ATTDOT; s.dst = Y.wp_off[C.src] + (int64_t)Q.asoff * Y.ldw[C.src];
      s.rows = Q.H; s.rows_pad = fpad(Q.H); s.att = C.a0; s.src[0] = C.w0;
      push_pack(s);
      // V_dst^T rows (destination role)
      s.dst = Y.wp_off[C.dst] + (int64_t)Q.adoff * Y.ldw[C.dst];
      s.cols = ft; s.ld_dst = Y.ldw[C.dst]; s.ld_src = ft; s.att = C.a1; s.src[0] = wd;
      push_pack(s);
      if (C.edge_dim > 0) {
        s.kind = PACK_ATTDOT_T; s.dst = Q.vedge_off; s.rows = C.edge_dim; s.rows_pad = GAT_MAX_EDIM; s.cols = Q.H;
        s.ld_dst = GAT_HMAX; s.ld_src = C.edge_dim; s.att = C.a2; s.src[0] = C.w2;
        push_pack(s);
      }
      const int64_t S_hs = Y.slab_off[C.src] + (int64_t)Q.coff * Y.lddw[C.src];
      const int64_t S_as = Y.slab_off[C.src] + (int64_t)Q.asoff * Y.lddw[C.src];
      const int64_t S_ad = Y.slab_off[C.dst] + (int64_t)Q.adoff * Y.lddw[C.dst];
      GradSeg g;
      memset(&g, 0, sizeof(g));
      // lin_src.weight
      g.dst = C.w0; g.rows = HC; g.cols = fs;
      g.t[g.n_terms++] = term(GT_COPY, slab_id_w(l, C.src), S_hs, Y.lddw[C.src], Q.C, Q.Cp);
      g.t[g.n_terms++] = term(GT_ATT_OUTER, slab_id_w(l, C.src), S_as, Y.lddw[C.src], Q.C, Q.C, 0, C.a0);
      if (Q.wd_is_src) g.t[g.n_terms++] = term(GT_ATT_OUTER, slab_id_w(l, C.dst), S_ad, Y.lddw[C.dst], Q.C, Q.C, 0, C.a1);
      push_grad(g);
      if (!Q.wd_is_src) {  // separate lin_dst.weight
        memset(&g, 0, sizeof(g));
        g.dst = C.w1; g.rows = HC; g.cols = ft; g.n_terms = 1;
        g.t[0] = term(GT_ATT_OUTER, slab_id_w(l, C.dst), S_ad, Y.lddw[C.dst], Q.C, Q.C, 0, C.a1);
        push_grad(g);
      }
      memset(&g, 0, sizeof(g));  // att_src
      g.dst = C.a0; g.rows = HC; g.cols = 1; g.n_terms = 1;
      g.t[0] = term(GT_ATT_DOT, slab_id_w(l, C.src), S_as, Y.lddw[C.src], Q.C, Q.C, fs, 0, C.w0, fs);
      push_grad(g);
      memset(&g, 0, sizeof(g));  // att_dst
      g.dst = C.a1; g.rows = HC; g.cols = 1; g.n_terms = 1;
      g.t[0] = term(GT_ATT_DOT, slab_id_w(l, C.dst), S_ad, Y.lddw[C.dst], Q.C, Q.C, ft, 0, wd, ft);
      push_grad(g);
      if (C.edge_dim > 0) {
        memset(&g, 0, sizeof(g));  // lin_edge.weight [HC, D]
        g.dst = C.w2; g.rows = HC; g.cols = C.edge_dim; g.n_terms = 1;
        g.t[0] = term(GT_ATT_OUTER_T, slab_id_v(l, c), Q.vslab_off, GAT_HMAX, Q.C, Q.C, 0, C.a2);
        push_grad(g);
        memset(&g, 0, sizeof(g));  // att_edge
        g.dst = C.a2; g.rows = HC; g.cols = 1; g.n_terms = 1;
        g.t[0] = term(GT_ATT_DOT_T, slab_id_v(l, c), Q.vslab_off, GAT_HMAX, Q.C, Q.C, C.edge_dim, 0, C.w2, C.edge_dim);
        push_grad(g);
      }
      memset(&g, 0, sizeof(g));  // bias: column sums of the destination gradient
      const int wout = Ls.out_dim[C.dst];
      g.dst = C.b0; g.rows = wout; g.cols = 1; g.n_terms = 1;
      g.t[0] = term(GT_COPY, slab_id_b(l, C.dst), Y.bslab_off[C.dst], 4, wout, wout, 0, 0, 0, 0,
                    Ls.group_mean ? 1.f / (float)Y.n_in[C.dst] : 1.f);
      push_grad(g);
    }
    for (int t = 0; t < n->T; ++t) {  // per destination type: SAGE root + bias sums, GAT bias sums
      if (Y.n_in[t] == 0) continue;
      const int fo = Ls.out_dim[t], ft = n->dim[l][t];
      PackSeg s, b;
      memset(&s, 0, sizeof(s));
      memset(&b, 0, sizeof(b));
      s.kind = PACK_SUM; b.kind = PACK_SUM;
      s.dst = Y.wp_off[t] + (int64_t)(Y.roff[t] < 0 ? 0 : Y.roff[t]) * Y.ldw[t];
      s.rows = fo; s.rows_pad = fpad(fo); s.cols = ft; s.ld_dst = Y.ldw[t]; s.ld_src = ft;
      b.dst = Y.bias_off[t];
      b.rows = 1; b.rows_pad = 1; b.cols = fo; b.ld_dst = fpad(fo); b.ld_src = fo;
      for (int i = 0; i < Y.n_live; ++i) {
        const hmp_conv_spec& C = Ls.convs[Y.live[i]];
        if (C.dst != t) continue;
        b.src[b.nsrc++] = C.b0;
        if (Y.kind != HMP_CONV_SAGE) continue;
        s.src[s.nsrc++] = C.w1;
        GradSeg g;
        memset(&g, 0, sizeof(g));
        g.dst = C.w1; g.rows = fo; g.cols = ft; g.n_terms = 1;
        const int64_t root = Y.slab_off[t] + (int64_t)Y.roff[t] * Y.lddw[t];
        g.t[0] = term(GT_COPY, slab_id_w(l, t), root, Y.lddw[t], fo, fo);
        push_grad(g);
        memset(&g, 0, sizeof(g));  // bias: the ones-column (index ft) of the root rows
        g.dst = C.b0; g.rows = fo; g.cols = 1; g.n_terms = 1;
        g.t[0] = term(GT_COPY, slab_id_w(l, t), root + ft, Y.lddw[t], fo, fo);
        push_grad(g);
      }
      if (Y.kind == HMP_CONV_SAGE) push_pack(s);
      push_pack(b);
    }
  }
  prs.push_back(prow);
  ges.push_back(gel);
  n->n_pack = (int)ps.size(); n->pack_rows = prow;
  n->n_grad = (int)gs.size(); n->grad_elems = gel;
  HMP_CHECK_ARG(ps.size() <= (size_t)SEG_MAX && gs.size() <= (size_t)SEG_MAX, "net: too many parameter segments (%zu / %zu > %d)",
                ps.size(), gs.size(), SEG_MAX);
  n->pack_sb.n = (int)ps.size();
  n->pack_sb.start[0] = 0;
  for (size_t i = 0; i < ps.size(); ++i)  // one wave per (row, 64-column chunk), 4 waves per block
    n->pack_sb.start[i + 1] = n->pack_sb.start[i] + cdiv((int64_t)ps[i].rows_pad * cdiv(ps[i].ld_dst, 64), 4);
  {
    std::vector<int2> pm;
    for (size_t i = 0; i < ps.size(); ++i) {
      const int items = ps[i].rows_pad * cdiv(ps[i].ld_dst, 64);
      for (int it = 0; it < items; it += 16) pm.push_back(make_int2((int)i, it));
    }
    n->n_pack_blocks16 = (int)pm.size();
    if (!pm.empty()) {
      HMP_HIP(hipMalloc(&n->d_pack_map, pm.size() * sizeof(int2)));
      HMP_HIP(hipMemcpy(n->d_pack_map, pm.data(), pm.size() * sizeof(int2), hipMemcpyHostToDevice));
    }
  }
  n->grad_sb.n = (int)gs.size();
  n->grad_sb.start[0] = 0;
  for (size_t i = 0; i < gs.size(); ++i) {
    const bool wave_mode = gs[i].n_terms == 1 && (gs[i].t[0].kind == GT_ATT_DOT || gs[i].t[0].kind == GT_ATT_DOT_T);
    const int64_t el = (int64_t)gs[i].rows * gs[i].cols;
    n->grad_sb.start[i + 1] = n->grad_sb.start[i] + cdiv(el, wave_mode ? 4 : 256);
  }
  HMP_HIP(hipMalloc(&n->d_pack_segs, ps.size() * sizeof(PackSeg)));
  HMP_HIP(hipMalloc(&n->d_grad_segs, gs.size() * sizeof(GradSeg)));
  HMP_HIP(hipMemcpy(n->d_pack_segs, ps.data(), ps.size() * sizeof(PackSeg), hipMemcpyHostToDevice));
  HMP_HIP(hipMemcpy(n->d_grad_segs, gs.data(), gs.size() * sizeof(GradSeg), hipMemcpyHostToDevice));
  for (int l = 0; l < n->L; ++l)
    if (n->lay[l].kind == HMP_CONV_GAT) HMP_HIP(hipMalloc(&n->lay[l].d_gat, sizeof(GatLayerS)));
  return HMP_OK;
}

// workspace carving: returns bytes; when base != null also assigns pointers ---------------------------
size_t carve(hmp_net* n, char* base, const int32_t* cn, const int64_t* ce) {
  size_t off = 0;
  n->sadd_floats = 0;
  auto take = [&](size_t bytes) -> char* {
    char* p = base ? base + off : nullptr;
    off += align256(bytes);
    return p;
  };
  const hmp_net_spec& S = n->spec;
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
    n->degf[e] = (float*)take((size_t)nd * 4);
    n->ell[e] = (int*)take((size_t)nd * ELL_W * 4);
    n->t_ell[e] = (int*)take((size_t)ns * ELL_W * 4);
  }
  n->d_packed = (float*)take((size_t)n->packed_floats * 4);
  n->d_slabs = (float*)take((size_t)n->slab_floats * 4);
  for (int l = 0; l <= n->L; ++l)
    for (int t = 0; t < n->T; ++t) {
      n->H[l][t] = n->G[l][t] = nullptr;
      if (l >= 1 && n->dim[l][t] > 0 && !(l == 1 && n->pass0[t])) {
        n->H[l][t] = (float*)take((size_t)cn[t] * n->ld[l][t] * 4);
        n->G[l][t] = (float*)take((size_t)cn[t] * n->ld[l][t] * 4);
      }
    }
  for (int l = 0; l < n->L; ++l) {
    LayerLayout& Y = n->lay[l];
    for (int t = 0; t < n->T; ++t) {
      n->Z[l][t] = n->dZ[l][t] = nullptr;
      if (Y.ncols[t] > 0) {
        n->Z[l][t] = (float*)take((size_t)cn[t] * Y.ncols[t] * 4);
        n->dZ[l][t] = (float*)take((size_t)cn[t] * Y.ncols[t] * 4);
      }
    }
    int64_t sf = 0;  // gather-add scratch: one matrix per aggregate-first conv of the layer, side by side
    for (int i = 0; i < Y.n_live; ++i) {
      const int c = Y.live[i];
      ConvLayout& Q = Y.conv[c];
      if (!Q.agg_first) continue;
      const hmp_conv_spec& C = S.layers[l].convs[c];
      Q.mrows = (float*)take((size_t)cn[C.dst] * Q.ld_m * 4);
      Q.dmrows = (float*)take((size_t)cn[C.dst] * Q.ld_m * 4);
      sf += (int64_t)cn[C.src] * Q.ld_m;
    }
    n->sadd_floats = sf > n->sadd_floats ? sf : n->sadd_floats;
    if (Y.kind != HMP_CONV_GAT) continue;
    for (int i = 0; i < Y.n_live; ++i) {
      const int c = Y.live[i];
      const hmp_conv_spec& C = S.layers[l].convs[c];
      ConvLayout& Q = Y.conv[c];
      const size_t nd = (size_t)cn[C.dst], ne = (size_t)ce[C.edge_type] + (C.self_loops ? nd : 0);
      Q.smax = (float*)take(nd * GAT_HMAX * 4);
      Q.sden = (float*)take(nd * GAT_HMAX * 4);
      Q.alpha_drop = (float*)take(ne * GAT_HMAX * 4);
      Q.dlogit = (float*)take(ne * GAT_HMAX * 4);
      Q.dlogit_orig = C.edge_dim > 0 ? (float*)take((size_t)ce[C.edge_type] * GAT_HMAX * 4) : nullptr;
    }
  }
  int cap_out = cn[S.readout_type];
  if (S.pool_edge_type >= 0) cap_out = cn[S.edge_dst[S.pool_edge_type]];
  n->cap_out = cap_out;
  n->d_out = (float*)take((size_t)cap_out * n->out_ld * 4);
  n->d_gout = (float*)take((size_t)cap_out * n->out_ld * 4);
  n->d_row_lv = (float*)take((size_t)cap_out * 2 * 4);
  n->d_iota = nullptr; n->d_ones = nullptr; n->d_sadd = nullptr;
  if (n->any_agg_first) {
    int mx = 1;
    for (int t = 0; t < n->T; ++t) mx = cn[t] > mx ? cn[t] : mx;
    n->iota_cap = mx;
    n->d_iota = (int*)take((size_t)(mx + 1) * 4);
    n->d_ones = (float*)take((size_t)mx * 4);
    n->d_sadd = (float*)take((size_t)n->sadd_floats * 4);
  }
  return off;
}

// device tables of the GAT layers (hold workspace pointers => built when the workspace is bound)
int build_gat_tables(hmp_net* n) {
  const hmp_net_spec& S = n->spec;
  for (int l = 0; l < n->L; ++l) {
    LayerLayout& Y = n->lay[l];
    if (Y.kind != HMP_CONV_GAT) continue;
    const hmp_layer_spec& Ls = S.layers[l];
    GatLayerS& G = Y.h_gat;
    memset(&G, 0, sizeof(G));
    int dst_entry[HMP_MAX_NODE_TYPES];
    for (int t = 0; t < n->T; ++t) {
      dst_entry[t] = -1;
      if (Y.n_in[t] == 0) continue;
      dst_entry[t] = G.n_dst;
      GatDstS& D = G.d[G.n_dst++];
      D.t = t;
      D.out = n->H[l + 1][t]; D.ldo = n->ld[l + 1][t];
      D.bias = n->d_packed + Y.bias_off[t];
      D.act = Ls.act;
      D.drop_p = Ls.dropout;
      D.drop_stream = (uint32_t)(l * HMP_MAX_NODE_TYPES + t);
      D.group_scale = Ls.group_mean ? 1.f / (float)Y.n_in[t] : 1.f;
      const bool top = (l == n->L - 1);
      D.g = (top && t != S.aux_readout_type) ? nullptr : n->G[l + 1][t];  // second readout: its gradient is staged in G[L][aux]
      D.ldg = n->ld[l + 1][t];
      for (int i = 0; i < Y.n_live; ++i) {
        const int c = Y.live[i];
        const hmp_conv_spec& C = Ls.convs[c];
        if (C.dst != t) continue;
        const ConvLayout& Q = Y.conv[c];
        if (D.n_in == 0) { D.H = Q.H; D.C = Q.C; D.Cp = Q.Cp; D.concat = C.concat; }
        HMP_CHECK_ARG(D.H == Q.H && D.C == Q.C && D.concat == C.concat, "net: GAT convs reaching one node type must share heads/channels/concat");
        GatInS& I = D.in[D.n_in++];
        const hmp_plan& P = n->plan[C.edge_type];
        I.et = C.edge_type; I.src_t = C.src;
        I.rowptr = P.d_rowptr; I.col = P.d_col; I.eid = P.d_eid;
        I.t_rowptr = P.d_t_rowptr; I.t_col = P.d_t_col; I.t_pos = P.d_t_pos;
        I.z = n->Z[l][C.src]; I.ldz = Y.ncols[C.src]; I.hoff = Q.coff;
        I.za = I.z; I.ldza = I.ldz; I.asoff = Q.asoff;
        I.zd = n->Z[l][t]; I.ldzd = Y.ncols[t]; I.adoff = Q.adoff;
        I.edim = C.edge_dim;
        I.vedge = C.edge_dim > 0 ? n->d_packed + Q.vedge_off : nullptr;
        I.self_loops = C.self_loops;
        I.smax = Q.smax; I.sden = Q.sden;
        I.adrop_p = C.att_dropout;
        I.adrop_stream = (uint32_t)(1000 + l * HMP_MAX_CONVS + c);
        I.alpha_drop = Q.alpha_drop; I.dlogit = Q.dlogit; I.dlogit_orig = Q.dlogit_orig;
        I.dza_src = n->dZ[l][C.src]; I.lddza_src = Y.ncols[C.src];
        I.dz_dst = n->dZ[l][t]; I.lddz_dst = Y.ncols[t];
      }
    }
    for (int s = 0; s < n->T; ++s) {
      GatSrcS* Sx = nullptr;
      for (int i = 0; i < Y.n_live; ++i) {
        const int c = Y.live[i];
        const hmp_conv_spec& C = Ls.convs[c];
        if (C.src != s) continue;
        if (!Sx) {
          Sx = &G.s[G.n_src++];
          Sx->t = s; Sx->dz = n->dZ[l][s]; Sx->lddz = Y.ncols[s];
        }
        // locate the in-conv index inside the destination entry
        const GatDstS& D = G.d[dst_entry[C.dst]];
        int ii = -1;
        for (int q = 0; q < D.n_in; ++q)
          if (D.in[q].et == C.edge_type) ii = q;
        Sx->out[Sx->n_out].d = dst_entry[C.dst];
        Sx->out[Sx->n_out].i = ii;
        ++Sx->n_out;
      }
    }
    HMP_HIP(hipMemcpy(Y.d_gat, &G, sizeof(G), hipMemcpyHostToDevice));
  }
  return HMP_OK;
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
    (void)hipEventRecord(r.a, st);
    n->recs.push_back(r);
    idx = (int)n->recs.size() - 1;
  }
  void cancel() {  // nothing was launched inside (the launch was deferred): drop the record
    if (idx >= 0 && idx == (int)n->recs.size() - 1) {
      (void)hipEventDestroy(n->recs[idx].a);
      (void)hipEventDestroy(n->recs[idx].b);
      n->recs.pop_back();
    }
    idx = -1;
  }
  ~Scope() {
    if (idx >= 0) (void)hipEventRecord(n->recs[idx].b, st);
  }
};

// `to` waits for everything enqueued on `from` so far (fork when to is a side stream, join when to is the main stream)
int fork_to(hmp_net* n, hipStream_t from, hipStream_t to) {
  hipEvent_t e = n->evs[n->ev_i];
  n->ev_i = (n->ev_i + 1) % n->n_evs;
  HMP_HIP(hipEventRecord(e, from));
  HMP_HIP(hipStreamWaitEvent(to, e, 0));
  return HMP_OK;
}

enum { KC_PLAN = 0, KC_PACK, KC_GEMM_FWD, KC_AGG_FWD, KC_LOSS, KC_AGG_BWD, KC_GEMM_BWD, KC_GRAD_REDUCE, KC_ADAM, KC_GAT_FWD, KC_GAT_BWD, KC_POOL, KC_FRONT, KC_CHAIN };

DropCfg make_drop(const hmp_net* n, float p, uint32_t stream) {
  DropCfg d;
  d.k0 = (uint32_t)n->seed; d.k1 = (uint32_t)(n->seed >> 32);
  d.step = n->rng_step; d.stream = stream;
  d.thresh = drop_thresh(p); d.scale = 1.f / (1.f - p);
  d.step_dev = n->step_dev ? n->d_step : nullptr;
  return d;
}

GatDyn make_gat_dyn(const hmp_net* n, const hmp_batch* b) {
  GatDyn d;
  memset(&d, 0, sizeof(d));
  for (int t = 0; t < n->T; ++t) d.n_nodes[t] = b->n_nodes[t];
  for (int e = 0; e < n->ET; ++e) { d.n_edges[e] = b->n_edges[e]; d.edge_attr[e] = b->d_edge_attr[e]; }
  d.training = n->training;
  d.k0 = (uint32_t)n->seed; d.k1 = (uint32_t)(n->seed >> 32);
  d.step = n->rng_step;
  d.step_dev = n->step_dev ? n->d_step : nullptr;
  return d;
}

int check_batch(const hmp_net* n, const hmp_batch* b) {
  HMP_CHECK_ARG(n->bound, "net: workspace not bound (call hmp_net_bind_workspace)");
  const hmp_net_spec& S = n->spec;
  for (int t = 0; t < n->T; ++t) {
    HMP_CHECK_ARG(b->n_nodes[t] >= 0 && b->n_nodes[t] <= n->cap_nodes[t], "batch: node type %d has %d nodes, capacity %d", t, b->n_nodes[t], n->cap_nodes[t]);
    const bool reads_x = n->lay[0].ncols[t] > 0 || (n->pass0[t] && n->lay[1].ncols[t] > 0);
    if (reads_x && b->n_nodes[t] > 0)
      HMP_CHECK_ARG(b->d_x[t] != nullptr && b->ldx[t] >= n->dim[0][t], "batch: node type %d features missing or ld < %d", t, n->dim[0][t]);
  }
  for (int e = 0; e < n->ET; ++e) {
    HMP_CHECK_ARG(b->n_edges[e] >= 0 && b->n_edges[e] <= n->cap_edges[e], "batch: edge type %d has %lld edges, capacity %lld", e, (long long)b->n_edges[e], (long long)n->cap_edges[e]);
    HMP_CHECK_ARG(b->n_edges[e] == 0 || b->d_edge_index[e] != nullptr, "batch: edge type %d edge_index missing", e);
  }
  for (int l = 0; l < n->L; ++l) {
    const LayerLayout& Y = n->lay[l];
    for (int i = 0; i < Y.n_live; ++i) {
      const hmp_conv_spec& C = S.layers[l].convs[Y.live[i]];
      if (C.kind == HMP_CONV_GAT && C.edge_dim > 0 && b->n_edges[C.edge_type] > 0)
        HMP_CHECK_ARG(b->d_edge_attr[C.edge_type] != nullptr, "batch: edge type %d needs edge_attr [E, %d]", C.edge_type, C.edge_dim);
    }
  }
  const int n_out_expect = S.pool_edge_type >= 0 ? b->n_nodes[S.edge_dst[S.pool_edge_type]] : b->n_nodes[S.readout_type];
  HMP_CHECK_ARG(b->n_out == n_out_expect, "batch: n_out %d != %d", b->n_out, n_out_expect);
  return HMP_OK;
}

// ---- forward -------------------------------------------------------------------------------------------
// the ELL id tables of the current plan may be read (HMP_ELL=0: never -- the CSR-only gathers, for tests and A/B runs)
bool ell_use(const hmp_net* n) { return n->ell_on && n->ell_ok; }
void fill_plan_batch(hmp_net* n, const hmp_batch* b, PlanBatch& pb) {
  memset(&pb, 0, sizeof(pb));
  pb.n = n->ET;
  pb.need_tpos = n->any_gat ? 1 : 0;
  pb.clear_first = 0;  // counters were zeroed at bind time and every build leaves them zero
  for (int e = 0; e < n->ET; ++e) {
    PlanJob& J = pb.j[e];
    J.degf = n->degf[e];
    hmp_plan& P = n->plan[e];
    P.n_src = b->n_nodes[n->spec.edge_src[e]];
    P.n_dst = b->n_nodes[n->spec.edge_dst[e]];
    P.n_edges = b->n_edges[e];
    J.ei = b->d_edge_index[e];
    J.E = P.n_edges; J.n_src = P.n_src; J.n_dst = P.n_dst;
    J.rowptr = P.d_rowptr; J.col = P.d_col; J.eid = P.d_eid;
    J.t_rowptr = P.d_t_rowptr; J.t_col = P.d_t_col; J.t_pos = P.d_t_pos;
    {  // graph-sorted edge list vouched for by the caller (hmp_batch::d_edge_ptr)
      const int ts = n->spec.edge_src[e], td = n->spec.edge_dst[e];
      const bool on = n->env.plan_sliced;
      if (on && b->n_graphs > 0 && b->d_edge_ptr[e] && b->d_node_ptr[ts] && b->d_node_ptr[td]) {
        J.gp_edge = b->d_edge_ptr[e]; J.gp_src = b->d_node_ptr[ts]; J.gp_dst = b->d_node_ptr[td]; J.n_graphs = b->n_graphs;
      }
    }
    J.ell = n->ell_on ? n->ell[e] : nullptr;
    J.t_ell = n->ell_on ? n->t_ell[e] : nullptr;
    plan_carve(J, n->plan_scratch[e]);
  }
}

int run_plan(hmp_net* n, const hmp_batch* b, hipStream_t st) {
  Scope sc(n, KC_PLAN, st);
  PlanBatch pb;
  fill_plan_batch(n, b, pb);
  const int rc = plan_launch(pb, &n->d_state->status, st);
  n->ell_ok = rc == HMP_OK && pb.built_small != 0 && n->ell_on;
  return rc;
}

// Small (launch-latency-bound) batches (<= 65 536 nodes): the row-local GEMM that follows an aggregation (next layer's projection in the
// forward pass, the input gradient in the backward pass) runs inside the aggregation kernel on 16-row tiles.  Large
// batches keep the stand-alone 64x64-tile GEMM (16-row tiles would re-read the weights from L2 once per 16 rows).
inline bool fuse_small(const hmp_net* n, const hmp_batch* b) {
  if (n->any_agg_first) return false;  // aggregate-first convs exist in the stand-alone launch sequence only (a large-batch choice)
  if (n->fuse_mode >= 0) return n->fuse_mode == 1;
  int64_t total = 0;
  for (int t = 0; t < n->T; ++t) total += b->n_nodes[t];
  // measured crossover on MI355X (tools/fuse_sweep.py, profiles/r02_fuse_sweep.json: config-2 network, hidden 64, batch 32 .. 2048):
  // the small-batch sequence wins up to ~46 000 nodes (batch 512: 0.578 vs 0.586 ms) and loses from ~95 000 on (batch 1024:
  // 1.085 vs 1.016 ms).  What it pays is the stacked weights re-read from L2 once per 16-row tile: that crossover holds for
  // stacked operands of the hidden-64 class (<= 64 KB); wider layers keep round 1's 16 384 (config 4, hidden 128: fused
  // 0.455 ms against 0.634 ms stand-alone at ~10 000 nodes; hidden 256 is the bf16 regime's territory above that).
  int64_t wmax = 1;
  for (int l = 1; l < n->L; ++l)
    for (int t = 0; t < n->T; ++t) {
      const int64_t wb = (int64_t)n->lay[l].ncols[t] * n->lay[l].ldw[t] * 4;
      wmax = wb > wmax ? wb : wmax;
    }
  return total <= (wmax <= 65536 ? 65536 : 16384);
}

inline bool is_input(const hmp_net* n, int l, int t) { return l == 0 || (l == 1 && n->pass0[t]); }
const float* h_ptr(const hmp_net* n, int l, int t) { return is_input(n, l, t) ? n->batch.d_x[t] : n->H[l][t]; }
int h_ld(const hmp_net* n, int l, int t) { return is_input(n, l, t) ? n->batch.ldx[t] : n->ld[l][t]; }

// bf16 compute mode: only the throughput-bound regime (>= 1024 64x64 tiles in the call) leaves the exact fp32 kernel
// HMP_BF16_ALL=1 (tests): every GEMM call of a bf16-mode net takes the bf16 kernel whatever its size, so that a graph the
// float64 oracle can hold (4 x 10^4 objects) runs exactly the decisions of the 10^6-object regime (BASELINE config 5)
static thread_local bool g_bf16_all = false;  // EnvSwitches::bf16_all of the call in progress (set by read_env's callers)
inline bool bf16_all() { return g_bf16_all; }

bool gemm_takes_bf16(const std::vector<GemmProblem>& ps, bool allow_bf16) {
  if (allow_bf16 && bf16_all()) return true;
  int64_t tiles64 = 0;
  double work = 0.0;
  for (const GemmProblem& p : ps) {
    tiles64 += (int64_t)cdiv(p.M, 64) * cdiv(p.N, 64);
    work += (double)p.M * p.N * p.K;
  }
  return allow_bf16 && (tiles64 >= 1024 || work >= 1e9);
}

// launches a list of GEMM problems in groups of GEMM_MAX_PROB; ksplit_out receives the split of each problem
int gemm_many(std::vector<GemmProblem>& ps, bool want_split, hipStream_t st, std::vector<int>* ksplit_out, bool allow_bf16 = false) {
  bool bf16 = gemm_takes_bf16(ps, allow_bf16);
  for (const GemmProblem& p : ps)
    if (p.a_bf16 || p.c_bf16 || p.b_bf16 || p.h_bf16) {  // bf16-stored operands exist only for the bf16 kernel
      HMP_CHECK_ARG(allow_bf16, "net: bf16-stored GEMM operand outside bf16 compute mode");
      bf16 = true;
    }
  for (size_t base = 0; base < ps.size(); base += GEMM_MAX_PROB) {
    GemmBatch gb;
    memset(&gb, 0, sizeof(gb));
    const size_t cnt = (ps.size() - base) < (size_t)GEMM_MAX_PROB ? (ps.size() - base) : (size_t)GEMM_MAX_PROB;
    for (size_t i = 0; i < cnt; ++i) gb.p[gb.n++] = ps[base + i];
    if (bf16) HMP_TRY(gemm_bf16_launch(gb, want_split, MAX_SLABS, st));
    else HMP_TRY(gemm_launch(gb, want_split, 64, st));
    if (ksplit_out)
      for (size_t i = 0; i < cnt; ++i) ksplit_out->push_back(gb.p[i].ksplit);
  }
  return HMP_OK;
}

void fill_plan_batch(hmp_net* n, const hmp_batch* b, PlanBatch& pb);

// Front kernel arguments (layer-0 projection + plan + pack in one launch); false when the batch / network does not fit its
// limits (the caller then launches pack, projection and plan separately).
bool build_front(hmp_net* n, const hmp_batch* b, const float* d_params, FrontArgs& fa) {
  if (!n->env.front || !n->d_pack_map) return false;
  const hmp_net_spec& S = n->spec;
  const hmp_layer_spec& Ls = S.layers[0];
  const LayerLayout& Y = n->lay[0];
  // GAT layers: the projection's operand is the pack's OUTPUT (att . W rows), so only plan and pack share the launch (the plan,
  // 19 us at config 3, runs behind the 46 us pack); the link pass (t_pos) follows as its own launch
  const bool proj = Y.kind == HMP_CONV_SAGE && !n->any_gat;
  // HMP_BRANCH bit 0 forks a side stream for pack + layer-0 projection BEFORE this launch: a front launch on the main stream that
  // holds the pack (GAT: the projection's operand) would race with the projection on the side stream -> separate launches there
  if (!proj && n->use_branches && (n->branch_mask & 1)) return false;
  memset(&fa, 0, sizeof(fa));
  // ---- projection problems
  for (int s = 0; s < n->T && proj; ++s) {
    if (Y.ncols[s] == 0 || b->n_nodes[s] == 0) continue;
    if (fa.n_prob == FR_MAX_PROB) return false;
    FrontProb& P = fa.prob[fa.n_prob++];
    P.A = b->d_x[s]; P.lda = b->ldx[s];
    P.C = n->Z[0][s]; P.ldc = Y.ncols[s];
    P.M = b->n_nodes[s]; P.N = Y.ncols[s]; P.K = n->dim[0][s];
    if (P.K < 4 || P.K > 384 || (P.K & 1) || (P.lda & 1) || (reinterpret_cast<uintptr_t>(P.A) & 7)) return false;
    for (int i = 0; i < Y.n_live; ++i) {
      const int c = Y.live[i];
      const hmp_conv_spec& C = Ls.convs[c];
      if (C.src != s) continue;
      if (P.n_seg == FR_MAX_SEG) return false;
      FrontSeg& G = P.seg[P.n_seg++];
      G.col0 = Y.conv[c].coff; G.rows = C.f_out; G.nsrc = 1; G.ld = P.K;
      G.off[0] = (uint32_t)C.w0;
    }
    if (Y.roff[s] >= 0) {
      if (P.n_seg == FR_MAX_SEG) return false;
      FrontSeg& G = P.seg[P.n_seg++];
      G.col0 = Y.roff[s]; G.rows = Ls.out_dim[s]; G.nsrc = 0; G.ld = P.K;
      for (int i = 0; i < Y.n_live; ++i) {
        const hmp_conv_spec& C = Ls.convs[Y.live[i]];
        if (C.dst != s) continue;
        if (G.nsrc == FR_MAX_SRC) return false;
        G.off[G.nsrc++] = (uint32_t)C.w1;
      }
      if (G.nsrc == 0) return false;
    }
  }
  if ((proj && fa.n_prob == 0) || S.n_params >= ((int64_t)1 << 31)) return false;
  // ---- plan parts
  PlanBatch pb;
  fill_plan_batch(n, b, pb);
  int64_t E = 0;
  for (int e = 0; e < pb.n; ++e) E += pb.j[e].E;
  if (E > PS_MAX_EDGES) return false;
  fa.n_jobs = pb.n;
  fa.need_tpos = pb.need_tpos;
  fa.plan_rc = plan_small_fits_rc(pb) ? 1 : 0;
  fa.plan_blocks = plan_small_layout(pb, fa.part_start, fa.rows_per_part);
  if (n->reuse_plan) fa.plan_blocks = 0;  // same topology as the previous call: no plan role in this launch
  fa.ws = n->ws_base;
  auto woff = [&](const void* p) -> uint32_t { return (uint32_t)((reinterpret_cast<const char*>(p) - n->ws_base) >> 2); };
  if (n->ws_bytes >= ((size_t)1 << 34)) return false;  // 32-bit word offsets
  for (int e = 0; e < pb.n; ++e) {
    const PlanJob& J = pb.j[e];
    FrontJob& F = fa.job[e];
    F.ei = J.ei; F.E = (int)J.E; F.n_src = J.n_src; F.n_dst = J.n_dst;
    F.rowptr = woff(J.rowptr); F.col = woff(J.col); F.eid = woff(J.eid);
    F.t_rowptr = woff(J.t_rowptr); F.t_col = woff(J.t_col); F.t_eid = woff(J.t_eid);
    F.tmp_in = woff(J.tmp_in); F.tmp_out = woff(J.tmp_out); F.pos_of_eid = woff(J.pos_of_eid); F.degf = woff(J.degf);
    F.gp_dst = J.gp_dst; F.gp_src = J.gp_src; F.gp_edge = J.gp_edge; F.n_graphs = J.n_graphs;
    F.ell = J.ell ? woff(J.ell) : 0u;
    F.t_ell = J.t_ell ? woff(J.t_ell) : 0u;
  }
  fa.status = &n->d_state->status;
  // ---- pack blocks
  fa.params = d_params;
  fa.segs = n->d_pack_segs;
  fa.pack_map = n->d_pack_map;
  fa.pack_blocks = n->n_pack_blocks16;
  fa.packed = n->d_packed;
  fa.step_ctr = n->step_dev ? n->d_step : nullptr;
  fa.step_mirror = &n->d_state->last_step;
  return true;
}

// launches what was deferred, in order, as the multi-launch sequence would have
int chain_flush(hmp_net* n, hipStream_t st) {
  n->chain_try = false;
  for (auto& d : n->deferred) {
    if (d.kind <= 1) {
      Scope sc(n, KC_AGG_FWD, st);
      HMP_TRY(d.kind == 0 ? agg_proj_fwd_launch(d.fa, st) : agg_fwd_launch(d.fa, st));
    } else {
      Scope sc(n, KC_AGG_BWD, st);
      HMP_TRY(d.kind == 2 ? agg_bwd_dx_launch(d.ba, st) : agg_bwd_launch(d.ba, st));
    }
  }
  n->deferred.clear();
  return HMP_OK;
}

inline int gs_of(int fmax) {  // pick_shape of aggregate.hip for NV == 1 shapes
  int gs = 8;
  const int lanes = (fmax + 3) / 4;
  while (gs < 64 && gs < lanes) gs <<= 1;
  return gs;
}
inline int gs_tile(int fmax) {  // agg_proj_fwd_launch / agg_bwd_dx_launch
  int gs = 16;
  while (gs < 64 && gs * 4 < fmax) gs <<= 1;
  return gs;
}

// everything deferred -> ONE chain launch, if the recorded sequence is exactly {L-1 x agg+proj, agg+CE, L-1 x aggT+dX, aggT} of one
// tile shape; else the multi-launch sequence
int chain_run(hmp_net* n, hipStream_t st) {
  if (!n->chain_try) return HMP_OK;
  const hmp_batch* b = &n->batch;
  const int L = n->L;
  bool ok = (int)n->deferred.size() == 2 * L && L >= 1 && L <= CHAIN_MAX_LAYERS && b->n_graphs > 0;
  int gs = 0, gs_last = 0, gs_first = 0, kmax = 0;
  for (int i = 0; ok && i < 2 * L; ++i) {
    const hmp_net::Deferred& d = n->deferred[i];
    const int want = i < L - 1 ? 0 : (i == L - 1 ? 1 : (i < 2 * L - 1 ? 2 : 3));
    if (d.kind != want) { ok = false; break; }
    int fmax = 0;
    if (d.kind <= 1) {
      for (int q = 0; q < d.fa.n; ++q) fmax = d.fa.d[q].F > fmax ? d.fa.d[q].F : fmax;
      if (d.kind == 0) { const int g = gs_tile(fmax); if (gs && g != gs) ok = false; gs = g; }
      else {
        gs_last = gs_of(fmax);
        for (int q = 0; q < d.fa.n; ++q) if (!d.fa.d[q].ce_labels) ok = false;  // the readout rows carry the loss
        if (d.fa.zb16 || d.fa.hb16) ok = false;
      }
    } else {
      for (int q = 0; q < d.ba.n; ++q) {
        for (int o = 0; o < d.ba.s[q].n_out; ++o) fmax = d.ba.s[q].out[o].F > fmax ? d.ba.s[q].out[o].F : fmax;
        if (d.ba.s[q].groot) fmax = d.ba.s[q].Froot > fmax ? d.ba.s[q].Froot : fmax;
        kmax = d.ba.s[q].ncols > kmax ? d.ba.s[q].ncols : kmax;
      }
      if (d.kind == 2) { const int g = gs_tile(fmax); if (gs && g != gs) ok = false; gs = g; }
      else { gs_first = gs_of(fmax); if (d.ba.gb16 || d.ba.dzb16) ok = false; }
    }
  }
  if (L == 1) gs = 16;
  const int stride = (256 * 17 > kmax * 17) ? 256 * 17 : kmax * 17;
  const size_t lds = (size_t)2 * stride * sizeof(float);  // CHAIN_GROUPS = 2 groups of 256 threads
  if (ok) ok = (gs == 16 || gs == 32) && gs_last <= 32 && gs_first <= 32 && lds + sizeof(ChainArgs) + 16 <= 150 * 1024;
  if (!ok) return chain_flush(n, st);
  // ---- argument block
  if (!n->chain_h) {
    HMP_HIP(hipHostMalloc((void**)&n->chain_h, sizeof(ChainArgs), hipHostMallocDefault));
    HMP_HIP(hipHostMalloc((void**)&n->chain_stage, sizeof(ChainArgs), hipHostMallocDefault));
    HMP_HIP(hipMalloc((void**)&n->d_chain, sizeof(ChainArgs)));
    HMP_HIP(hipEventCreateWithFlags(&n->chain_copied, hipEventDisableTiming));
    n->chain_valid = false;
  }
  ChainArgs& C = *n->chain_stage;
  HMP_HIP(hipEventSynchronize(n->chain_copied));  // the previous upload has left the staging buffer (no-op when none is pending)
  memset(&C, 0, sizeof(C));
  C.L = L; C.lds_stride = stride; C.gs_last = gs_last; C.gs_first = gs_first;
  for (int t = 0; t < n->T; ++t) C.ptr[t] = b->d_node_ptr[t];
  for (int l = 0; l < L; ++l) {
    C.fwd[l] = n->deferred[l].fa;
    for (int q = 0; q < C.fwd[l].n; ++q) { C.fwd_type[l][q] = C.fwd[l].d[q].type; C.fwd[l].d[q].n_rows = 0; C.fwd[l].d[q].block_start = 0; }
    C.fwd[l].total_blocks = 0;
    TAggArgs& B = C.bwd[L - 1 - (l)];  // deferred[L + j] is layer L - 1 - j
    B = n->deferred[L + l].ba;
    for (int q = 0; q < B.n; ++q) { C.bwd_type[L - 1 - l][q] = B.s[q].type; B.s[q].n_rows = 0; B.s[q].block_start = 0; }
    B.total_blocks = 0;
    if (B.fin_row_lv) { C.fin_row_lv = B.fin_row_lv; C.fin_out2 = B.fin_out2; C.fin_state = B.fin_state; }
    B.fin_row_lv = nullptr; B.fin_rows = 0; B.fin_out2 = nullptr; B.fin_state = nullptr;
  }
  C.ticket = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(n->d_state) + 128);
  for (int t = 0; t < n->T; ++t) {  // every node type an entry works on needs its row offsets
    bool used = false;
    for (int l = 0; l < L; ++l) {
      for (int q = 0; q < C.fwd[l].n; ++q) used = used || C.fwd_type[l][q] == t;
      for (int q = 0; q < C.bwd[l].n; ++q) used = used || C.bwd_type[l][q] == t;
    }
    if (used && !C.ptr[t]) return chain_flush(n, st);
  }
  if (!n->chain_valid || memcmp(&C, n->chain_h, sizeof(ChainArgs)) != 0) {
    memcpy(n->chain_h, &C, sizeof(ChainArgs));
    HMP_HIP(hipMemcpyAsync(n->d_chain, n->chain_stage, sizeof(ChainArgs), hipMemcpyHostToDevice, st));
    HMP_HIP(hipEventRecord(n->chain_copied, st));
    n->chain_valid = true;
  }
  n->deferred.clear();
  n->chain_try = false;
  Scope sc(n, KC_CHAIN, st);
  return chain_launch(n->d_chain, b->n_graphs, b->n_out, gs, lds, st);
}

int forward_impl(hmp_net* n, const hmp_batch* b, const float* d_params, hipStream_t st) {
  HMP_TRY(check_batch(n, b));
  n->batch = *b;
  const hmp_net_spec& S = n->spec;
  // The parameter pack and the layer-0 projection do not depend on the plan: they run on a side stream next to the
  // (4-5 dependent launches of the) plan build and join before the first aggregation.  Under capture this becomes
  // two parallel branches of the hipGraph.
  hipStream_t main_st = st;
  hipStream_t side = (n->use_branches && (n->branch_mask & 1)) ? n->side[0] : main_st;
  if (side != main_st) HMP_TRY(fork_to(n, main_st, side));
  n->fuse_now = fuse_small(n, b);
  n->ell_on = n->env.ell;  // (experiment builds: neighbour ids from the plan's ELL tables; measured slower, see kernels.h)
  memset(n->h16, 0, sizeof(n->h16));
  n->reuse_plan = false;
  if (b->plan_valid) {
    HMP_CHECK_ARG(n->plan_ok, "batch: plan_valid without a plan from a previous call (first call, or the workspace was re-bound)");
    for (int e = 0; e < n->ET; ++e)
      HMP_CHECK_ARG(n->plan[e].n_edges == b->n_edges[e] && n->plan[e].n_src == b->n_nodes[S.edge_src[e]] && n->plan[e].n_dst == b->n_nodes[S.edge_dst[e]],
                    "batch: plan_valid but edge type %d changed shape since the previous call", e);
    n->reuse_plan = true;
  }
  // Small batches, SAGE layer 0: projection (reading the stacked weights straight from the flat parameters), plan and pack
  // are roles of ONE launch (front.hip) -- the plan, which has to read whole edge lists through single CUs, hides behind
  // the projection tiles.
  FrontArgs fa;
  const bool front = n->fuse_now && build_front(n, b, d_params, fa);
  if (front) {
    Scope sc(n, KC_FRONT, main_st);
    HMP_TRY(front_launch(fa, main_st));
    if (fa.plan_blocks > 0) n->ell_ok = n->ell_on;  // the front kernel's plan role is the single-launch build
    if (fa.need_tpos && fa.plan_blocks > 0) {
      PlanBatch pb;
      fill_plan_batch(n, b, pb);
      HMP_TRY(plan_link_launch(pb, main_st));
    }
    for (int e = 0; e < n->ET; ++e) {  // what run_plan records on the host
      hmp_plan& P = n->plan[e];
      P.n_src = b->n_nodes[S.edge_src[e]]; P.n_dst = b->n_nodes[S.edge_dst[e]]; P.n_edges = b->n_edges[e];
    }
  } else {
    Scope sc(n, KC_PACK, side);
    HMP_TRY(pack_launch(n->d_pack_segs, n->pack_sb, d_params, n->d_packed, n->step_dev ? n->d_step : nullptr, &n->d_state->last_step, side));
  }
  bool z_done = false;
  for (int l = 0; l < n->L; ++l) {
    const hmp_layer_spec& Ls = S.layers[l];
    LayerLayout& Y = n->lay[l];
    st = (l == 0) ? side : main_st;
    bool z16 = false;  // this layer's projected rows are stored as bf16 (decided with the projection, read by the aggregation)
    if (l == 0 && !front && n->any_agg_first) {  // the segment means of layer 0 read the plan: build it first (no side stream here)
      if (!n->reuse_plan) HMP_TRY(run_plan(n, b, main_st));
      if (side != main_st) HMP_TRY(fork_to(n, main_st, side));
    }
    if (l == 0 && front && fa.n_prob > 0) {
      // projection, plan and pack already ran in the front kernel
    } else if (!z_done) {  // grouped projection (skipped when the previous layer's aggregation kernel already produced Z[l])
      HMP_TRY(chain_flush(n, st));
      Scope sc(n, KC_GEMM_FWD, st);
      std::vector<GemmProblem> ps;
      for (int s = 0; s < n->T; ++s) {
        if (Y.ncols[s] == 0 || b->n_nodes[s] == 0) continue;
        GemmProblem p;
        memset(&p, 0, sizeof(p));
        p.A = h_ptr(n, l, s); p.lda = h_ld(n, l, s); p.trans_a = 0;
        p.a_bf16 = n->h16[l][s] ? 1 : 0;
        p.B = n->d_packed + Y.wp_off[s]; p.ldb = Y.ldw[s]; p.trans_b = 1;
        p.C = n->Z[l][s]; p.ldc = Y.ncols[s];
        p.M = b->n_nodes[s]; p.N = Y.ncols[s]; p.K = n->dim[l][s];
        p.n_real = p.N;
        p.epi = EPI_NONE;
        ps.push_back(p);
      }
      // bf16 compute mode, 10^6-row regime: the projected rows are only ever gathered by this layer's aggregation, so they are
      // stored as bf16 (half the projection's write and half the gather's read traffic) -- when the bf16 GEMM runs AND the
      // aggregation takes its one-wavefront-per-row shape (every output width in (128, 256]), the one that reads bf16 rows
      if (Y.kind != HMP_CONV_GAT && !n->fuse_now && gemm_takes_bf16(ps, n->compute_bf16 != 0)) {
        int fmin = 1 << 30, fmax = 0;
        for (int t = 0; t < n->T; ++t) {
          if (Y.roff[t] < 0 || b->n_nodes[t] == 0) continue;
          const int f = fpad(Ls.out_dim[t]);
          fmin = f < fmin ? f : fmin;
          fmax = f > fmax ? f : fmax;
        }
        z16 = fmax <= 256 && fmin > 128;
        for (GemmProblem& p : ps) z16 = z16 && (p.ldc & 3) == 0;
        if (!n->env.z16) z16 = false;
        if (z16)
          for (GemmProblem& p : ps) p.c_bf16 = 1;
      }
      HMP_TRY(gemm_many(ps, false, st, nullptr, n->compute_bf16 != 0));
    }
    if (n->any_agg_first && !z_done) {
      // aggregate-first convs: mean of the SOURCE rows per destination row, then the destination-sized projection into the conv's
      // block of Z[l][dst] -- a second launch: the main one wrote that block too (zero weights there)
      std::vector<GemmProblem> pa;
      for (int i = 0; i < Y.n_live; ++i) {
        const int c = Y.live[i];
        const ConvLayout& Q = Y.conv[c];
        if (!Q.agg_first) continue;
        const hmp_conv_spec& C = Ls.convs[c];
        if (b->n_nodes[C.dst] == 0 || b->n_nodes[C.src] == 0 || b->n_edges[C.edge_type] == 0) continue;
        {
          Scope sa(n, KC_AGG_FWD, st);
          const hmp_plan& Pl = n->plan[C.edge_type];
          HMP_TRY(seg_mean_rows_launch(h_ptr(n, l, C.src), h_ld(n, l, C.src), n->h16[l][C.src] ? 1 : 0, n->dim[l][C.src], Pl.d_rowptr, Pl.d_col,
                                       b->n_nodes[C.dst], Q.mrows, Q.ld_m, st));
        }
        GemmProblem p;
        memset(&p, 0, sizeof(p));
        p.A = Q.mrows; p.lda = Q.ld_m; p.trans_a = 0;
        p.B = n->d_packed + Q.wc_off; p.ldb = Y.ldw[C.src]; p.trans_b = 1;
        p.C = z16 ? reinterpret_cast<float*>(reinterpret_cast<uint16_t*>(n->Z[l][C.dst]) + Q.aoff) : n->Z[l][C.dst] + Q.aoff;
        p.c_bf16 = z16 ? 1 : 0;
        p.ldc = Y.ncols[C.dst];
        p.M = b->n_nodes[C.dst]; p.N = fpad(C.f_out); p.K = n->dim[l][C.src];
        p.n_real = p.N;
        p.epi = EPI_NONE;
        pa.push_back(p);
      }
      if (!pa.empty()) {
        Scope sg(n, KC_GEMM_FWD, st);
        HMP_TRY(gemm_many(pa, false, st, nullptr, n->compute_bf16 != 0));
      }
    }
    z_done = false;
    if (l == 0 && !front) {
      if (!n->reuse_plan && !n->any_agg_first) HMP_TRY(run_plan(n, b, main_st));
      if (side != main_st) HMP_TRY(fork_to(n, side, main_st));  // join
      st = main_st;
    }
    if (Y.kind == HMP_CONV_GAT) {
      HMP_TRY(chain_flush(n, st));
      Scope sc(n, KC_GAT_FWD, st);
      GatDyn dyn = make_gat_dyn(n, b);
      HMP_TRY(gat_fwd_launch(Y.d_gat, Y.h_gat, dyn, st));
      continue;
    }
    {  // fused aggregation + root + bias + activation + dropout
      Scope sc(n, KC_AGG_FWD, st);
      AggArgs a;
      memset(&a, 0, sizeof(a));
      a.mean = 1;
      a.state = n->d_state;
      for (int t = 0; t < n->T; ++t) {
        if (Y.roff[t] < 0 || b->n_nodes[t] == 0) continue;
        AggDst& D = a.d[a.n++];
        if (l == n->L - 1 && t == S.readout_type && n->ce_labels && S.pool_edge_type < 0 && n->fuse_now && fpad(Ls.out_dim[t]) <= 256) {
          // masked cross entropy in the epilogue of the last aggregation (one kernel less on the step's critical path)
          D.ce_labels = n->ce_labels; D.ce_ignored = n->ce_ignored; D.ce_classes = n->out_dim;
          D.ce_grad = n->d_gout; D.ce_ldg = n->out_ld; D.ce_row_lv = n->d_row_lv;
          n->ce_done = true;
        }
        D.n_rows = b->n_nodes[t];
        D.type = t;
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
          if (Y.conv[c].agg_first) {  // the conv's block of this type's own Z, through identity lists (in-degree 1)
            I.rowptr = n->d_iota; I.col = n->d_iota;
            I.z = n->Z[l][t]; I.ldz = Y.ncols[t]; I.coff = Y.conv[c].aoff;
            I.same_type = 0; I.n_src = b->n_nodes[t]; I.ell = nullptr;
            continue;
          }
          I.rowptr = n->plan[C.edge_type].d_rowptr;
          I.col = n->plan[C.edge_type].d_col;
          I.z = n->Z[l][C.src]; I.ldz = Y.ncols[C.src]; I.coff = Y.conv[c].coff;
          I.same_type = C.src == C.dst ? 1 : 0;
          I.n_src = b->n_nodes[C.src];
          I.ell = ell_use(n) ? n->ell[C.edge_type] : nullptr;
          if (n->fuse_now && b->n_edges[C.edge_type] > (int64_t)8 * b->n_nodes[C.dst]) D.tile_rows = 8;  // average in-degree > 8
        }
      }
      // fuse the projection of layer l+1 when every node type it reads is produced right here
      bool fuse = n->fuse_now && l + 1 < n->L;
      if (fuse) {
        const LayerLayout& Yn = n->lay[l + 1];
        for (int t = 0; t < n->T && fuse; ++t) {
          if (Yn.ncols[t] == 0 || b->n_nodes[t] == 0) continue;
          if (Y.roff[t] < 0 || is_input(n, l + 1, t)) fuse = false;                   // not produced by this launch
          if ((n->dim[l + 1][t] & 15) != 0 || n->dim[l + 1][t] > 256) fuse = false;  // kernel limits
        }
        for (int i = 0; i < a.n && fuse; ++i)
          if (a.d[i].F > 256) fuse = false;
      }
      if (fuse) {
        const LayerLayout& Yn = n->lay[l + 1];
        int ai = 0;
        for (int t = 0; t < n->T; ++t) {
          if (Y.roff[t] < 0 || b->n_nodes[t] == 0) continue;
          AggDst& D = a.d[ai++];
          if (Yn.ncols[t] == 0) continue;
          D.pw = n->d_packed + Yn.wp_off[t]; D.pldw = Yn.ldw[t]; D.pncols = Yn.ncols[t]; D.pK = n->dim[l + 1][t];
          D.pz = n->Z[l + 1][t]; D.pldz = Yn.ncols[t];
        }
        if (n->chain_try) {
          hmp_net::Deferred d;
          d.kind = 0; d.fa = a;
          n->deferred.push_back(d);
          sc.cancel();
        } else {
          HMP_TRY(agg_proj_fwd_launch(a, st));
        }
        z_done = true;
      } else {
        a.zb16 = z16 ? 1 : 0;
        // same regime: the activations H[l+1] of a hidden layer are read only by GEMMs of the bf16 kernel (next projection, its
        // weight gradient, the activation mask of its input gradient), which round them to bf16 on the way into LDS anyway --
        // written as bf16 here the results are bit-identical and every one of those reads, and this write, moves half the bytes.
        // Needs: every output of this launch exactly 256 wide (whole GEMM tiles), the next layer a SAGE layer whose three GEMMs
        // (same M x N x K products) take the bf16 kernel by their own size -- a bf16-stored operand would otherwise force a
        // narrow last-layer projection off the exact fp32 kernel (measured: 40 000 objects x 28 live columns)
        bool hb = z16 && l + 1 < n->L && n->lay[l + 1].kind != HMP_CONV_GAT;
        for (int i = 0; i < a.n && hb; ++i) hb = a.d[i].F == 256 && a.d[i].ldo == 256;
        if (hb) {
          double work = 0.0;
          for (int s = 0; s < n->T; ++s) work += (double)b->n_nodes[s] * n->lay[l + 1].ncols[s] * n->dim[l + 1][s];
          // an aggregate-first conv of the next layer reads its source rows with the segment-mean kernels (both storage forms):
          // counted as the projection it stands for, so that the choice does not depend on the order of evaluation
          for (int i = 0; i < n->lay[l + 1].n_live; ++i) {
            const int c = n->lay[l + 1].live[i];
            if (!n->lay[l + 1].conv[c].agg_first) continue;
            const hmp_conv_spec& C = S.layers[l + 1].convs[c];
            work += (double)b->n_nodes[C.src] * fpad(C.f_out) * n->dim[l + 1][C.src];
          }
          hb = work >= 1e9 || bf16_all();  // gemm_takes_bf16's work criterion
        }
        if (!n->env.h16) hb = false;
        if (hb) {
          a.hb16 = 1;
          for (int t = 0; t < n->T; ++t)
            if (Y.roff[t] >= 0 && b->n_nodes[t] > 0) n->h16[l + 1][t] = true;
        }
        if (n->chain_try && n->fuse_now && !z16) {
          hmp_net::Deferred d;
          d.kind = 1; d.fa = a;
          n->deferred.push_back(d);
          sc.cancel();
        } else {
          HMP_TRY(chain_flush(n, st));
          HMP_TRY(agg_fwd_launch(a, st));
        }
      }
    }
  }
  if (S.pool_edge_type >= 0) {
    HMP_TRY(chain_flush(n, st));
    Scope sc(n, KC_POOL, st);
    const int rt = S.readout_type;
    HMP_TRY(hmp_segment_mean_fwd(n->H[n->L][rt], n->ld[n->L][rt], n->out_ld, n->plan[S.pool_edge_type], n->d_out, n->out_ld, st));
  }
  n->have_fwd = true;
  n->plan_ok = true;
  return HMP_OK;
}

const float* out_ptr(const hmp_net* n) {
  return n->spec.pool_edge_type >= 0 ? n->d_out : n->H[n->L][n->spec.readout_type];
}

// ---- backward ------------------------------------------------------------------------------------------
int backward_impl(hmp_net* n, const float* d_gout, int ld_gout, float* d_grads, const float* d_params, float* const* d_gx,
                  hipStream_t st, const float* d_gaux = nullptr, int ld_gaux = 0) {
  HMP_CHECK_ARG(n->have_fwd, "net: backward without a forward");
  if (const int at = n->spec.aux_readout_type; at >= 0) {
    // second readout: stage its gradient in G[L][aux] (zero where the caller passed none, and in the padding columns)
    const int rows = n->batch.n_nodes[at], ld = n->ld[n->L][at], w = n->dim[n->L][at];
    HMP_CHECK_ARG(!d_gaux || ld_gaux >= w, "net: second output gradient narrower than the output (%d < %d)", ld_gaux, w);
    if (rows > 0) {
      HMP_HIP(hipMemsetAsync(n->G[n->L][at], 0, (size_t)rows * ld * 4, st));
      if (d_gaux)
        HMP_HIP(hipMemcpy2DAsync(n->G[n->L][at], (size_t)ld * 4, d_gaux, (size_t)ld_gaux * 4, (size_t)w * 4, rows, hipMemcpyDeviceToDevice, st));
    }
  } else {
    HMP_CHECK_ARG(!d_gaux, "net: a second output gradient for a net without aux_readout_type");
  }
  HMP_CHECK_ARG((ld_gout & 3) == 0 && ld_gout >= n->out_ld && (reinterpret_cast<uintptr_t>(d_gout) & 15) == 0,
                "net: output gradient must be 16-byte aligned with ld %% 4 == 0 and ld >= %d", n->out_ld);
  const hmp_net_spec& S = n->spec;
  const hmp_batch* b = &n->batch;
  const int rt = S.readout_type;
  const float* gtop = d_gout;
  int ld_gtop = ld_gout;
  if (S.pool_edge_type >= 0) {
    HMP_TRY(chain_flush(n, st));
    Scope sc(n, KC_POOL, st);
    HMP_TRY(hmp_segment_mean_bwd(d_gout, ld_gout, n->out_ld, n->plan[S.pool_edge_type], n->G[n->L][rt], n->ld[n->L][rt], st));
    gtop = n->G[n->L][rt];
    ld_gtop = n->ld[n->L][rt];
  }
  memset(&n->dyn, 0, sizeof(n->dyn));
  {
    // measured on MI355X (bench.py configs 2/3/4): below ~4k nodes per batch every kernel is launch-bound and one merged
    // weight-gradient launch wins (+2.5 %); above, overlapping the per-layer launches with the backward chain wins
    // (+4 % at 6k nodes, +10 % at 12k)
    int total_nodes = 0;
    for (int t = 0; t < n->T; ++t) total_nodes += b->n_nodes[t];
    n->dw_branch = n->use_branches && (n->branch_mask & 2) && (n->dw_mode >= 0 ? n->dw_mode == 1 : total_nodes > 4096);
  }
  std::vector<GemmProblem> wps;  // weight-gradient problems of all layers (merged mode)
  std::vector<int> wids;
  bool fin_early = false;
  bool g16[HMP_MAX_LAYERS + 2];  // g16[l]: the input gradients G[l][*] were stored as bf16 by layer l's input-gradient GEMM
  for (int i = 0; i < HMP_MAX_LAYERS + 2; ++i) g16[i] = false;
  bool dz16 = false;  // this layer's dZ is written as bf16 by the transposed aggregation (read by both backward GEMMs)
  for (int l = n->L - 1; l >= 0; --l) {
    const hmp_layer_spec& Ls = S.layers[l];
    LayerLayout& Y = n->lay[l];
    auto g_of = [&](int t, int& ldg) -> const float* {
      if (l == n->L - 1 && t == rt) { ldg = ld_gtop; return gtop; }
      ldg = n->ld[l + 1][t];
      return n->G[l + 1][t];
    };
    bool dx_fused = false;
    bool rootless = false;  // this layer's GEMMs take the root block of dZ from the output gradient
    dz16 = false;
    if (Y.kind == HMP_CONV_GAT) {
      HMP_TRY(chain_flush(n, st));
      Scope sc(n, KC_GAT_BWD, st);
      GatDyn dyn = make_gat_dyn(n, b);
      dyn.g_top = gtop; dyn.ld_gtop = ld_gtop;
      HMP_TRY(gat_bwd1_launch(Y.d_gat, Y.h_gat, dyn, st));
      HMP_TRY(gat_bwd2_launch(Y.d_gat, Y.h_gat, dyn, st));
    } else {  // transposed aggregation: gradient of the projected rows
      Scope sc(n, KC_AGG_BWD, st);
      TAggArgs a;
      memset(&a, 0, sizeof(a));
      a.mean = 1;
      if (n->fin_loss) {  // first backward kernel of the step: one extra block finalises {loss_sum, count}
        a.fin_row_lv = n->d_row_lv; a.fin_rows = b->n_out; a.fin_out2 = d_grads + n->spec.n_active_params; a.fin_state = n->d_state;
        n->fin_loss = false;
        fin_early = true;
      }
      for (int s = 0; s < n->T; ++s) {
        if (Y.ncols[s] == 0 || b->n_nodes[s] == 0) continue;
        TAggSrc& T = a.s[a.n++];
        T.n_rows = b->n_nodes[s];
        T.type = s;
        T.dz = n->dZ[l][s]; T.lddz = Y.ncols[s]; T.ncols = Y.ncols[s];
        if (Y.roff[s] >= 0) {
          int ldg;
          T.groot = g_of(s, ldg);
          T.ldgr = ldg; T.roff = Y.roff[s]; T.Froot = fpad(Ls.out_dim[s]);
        }
        for (int i = 0; i < Y.n_live; ++i) {
          const int c = Y.live[i];
          const hmp_conv_spec& C = Ls.convs[c];
          if (Y.conv[c].agg_first) {  // its block of dZ[l][dst] = the destination's own output gradient (identity lists, weight 1)
            if (C.dst != s || b->n_nodes[C.src] == 0 || b->n_edges[C.edge_type] == 0) continue;
            TAggOut& O = T.out[T.n_out++];
            O.t_rowptr = n->d_iota; O.t_col = n->d_iota; O.rowptr = n->d_iota; O.degf = n->d_ones;
            int ldg;
            O.g = g_of(s, ldg);
            O.ldg = ldg; O.coff = Y.conv[c].aoff; O.F = fpad(C.f_out);
            O.same_type = 0; O.n_dst = b->n_nodes[s]; O.t_ell = nullptr;
            continue;
          }
          if (C.src != s) continue;
          TAggOut& O = T.out[T.n_out++];
          const hmp_plan& P = n->plan[C.edge_type];
          O.t_rowptr = P.d_t_rowptr; O.t_col = P.d_t_col; O.rowptr = P.d_rowptr; O.degf = n->degf[C.edge_type];
          int ldg;
          O.g = g_of(C.dst, ldg);
          O.ldg = ldg; O.coff = Y.conv[c].coff; O.F = fpad(C.f_out);
          O.same_type = C.src == C.dst ? 1 : 0;
          O.n_dst = b->n_nodes[C.dst];
          O.t_ell = ell_use(n) ? n->t_ell[C.edge_type] : nullptr;
          if (n->fuse_now && b->n_edges[C.edge_type] > (int64_t)8 * b->n_nodes[C.src]) T.tile_rows = 8;  // average out-degree > 8
        }
      }
      // input gradient of layer l inside the same kernel (row-local GEMM on 16-row tiles) for small batches
      dx_fused = n->fuse_now && l > 0;
      int fmax = 0;
      for (int i = 0; i < a.n && dx_fused; ++i) {
        if (a.s[i].ncols > 896) dx_fused = false;
        for (int o = 0; o < a.s[i].n_out; ++o) fmax = a.s[i].out[o].F > fmax ? a.s[i].out[o].F : fmax;
        if (a.s[i].groot) fmax = a.s[i].Froot > fmax ? a.s[i].Froot : fmax;
      }
      if (fmax > 256) dx_fused = false;
      if (dx_fused) {
        const hmp_layer_spec& Lp = S.layers[l - 1];
        int ai = 0;
        for (int s = 0; s < n->T; ++s) {
          if (Y.ncols[s] == 0 || b->n_nodes[s] == 0) continue;
          TAggSrc& T = a.s[ai++];
          if (!n->G[l][s]) continue;  // passthrough input of layer 1: no gradient wanted
          T.xw = n->d_packed + Y.wp_off[s]; T.xldw = Y.ldw[s]; T.xN = n->dim[l][s];
          T.xg = n->G[l][s]; T.xldg = n->ld[l][s];
          T.xdrop_on = (n->training && Lp.dropout > 0.f) ? 1 : 0;
          T.xact = Lp.act;
          T.xscale = T.xdrop_on ? 1.f / (1.f - Lp.dropout) : 1.f;
          T.xh = (T.xact != HMP_ACT_NONE || T.xdrop_on) ? n->H[l][s] : nullptr;
          T.xldh = n->ld[l][s];
        }
        if (n->chain_try) {
          hmp_net::Deferred d;
          d.kind = 2; d.ba = a;
          n->deferred.push_back(d);
          sc.cancel();
        } else {
          HMP_TRY(agg_bwd_dx_launch(a, st));
        }
      } else {
        a.gb16 = g16[l + 1] ? 1 : 0;
        // dZ as bf16 too (it is only read back as the A operand of the two backward GEMMs): needs the bf16-reading kernel
        // above and whole 256-row tiles of every dZ^T (the bf16 operand loader has no edge path)
        dz16 = a.gb16 && n->compute_bf16 != 0;
        for (int s = 0; s < n->T && dz16; ++s)
          if (Y.ncols[s] > 0 && b->n_nodes[s] > 0 && (Y.ncols[s] % 256) != 0) dz16 = false;
        if (!n->env.z16) dz16 = false;
        a.dzb16 = dz16 ? 1 : 0;
        // the root block of dZ is a copy of the output gradient (d out / d z_root = 1): with both stored as bf16 and the block
        // on a 256-column boundary the two backward GEMMs read it where it is (GemmProblem::A2) and the copy -- 1 GB per layer
        // at 10^6 rows -- is not made
        rootless = dz16;
        for (int i = 0; i < a.n && rootless; ++i)
          if (a.s[i].groot && ((a.s[i].roff & 255) != 0 || a.s[i].Froot != 256 || (a.s[i].ldgr & 3) != 0)) rootless = false;
        if (n->env.rootcopy) rootless = false;
        if (rootless)
          for (int i = 0; i < a.n; ++i) a.s[i].groot = nullptr;
        if (n->chain_try && n->fuse_now && l == 0 && !a.gb16) {
          hmp_net::Deferred d;
          d.kind = 3; d.ba = a;
          n->deferred.push_back(d);
          sc.cancel();
        } else {
          HMP_TRY(chain_flush(n, st));
          HMP_TRY(agg_bwd_launch(a, st));
        }
      }
    }
    // dZ[l] is complete.  Weight-gradient GEMMs: either ALL layers in one grouped split-K launch after the loop
    // (default: one launch with ~1k workgroups instead of L launches, and no cross-queue fork per layer -- each fork
    // costs ~5 us of dependency latency under graph replay), or per layer on a side stream (HMP_DW_BRANCH=1).
    hipStream_t wst = st;
    if (n->use_branches && n->dw_branch) {
      wst = n->side[1];
      HMP_TRY(fork_to(n, st, wst));
    }
    const bool need_dx = !dx_fused && ((l > 0) || (d_gx != nullptr));
    // aggregate-first convs: dM = dZ_block * W_l (destination-sized), ahead of the launch whose epilogue scatters it
    auto dz_at = [&](int t, int col) -> const float* {
      return dz16 ? reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(n->dZ[l][t]) + col) : n->dZ[l][t] + col;
    };
    auto aggf_live = [&](int c) {
      const hmp_conv_spec& C = Ls.convs[c];
      return Y.conv[c].agg_first && b->n_nodes[C.dst] > 0 && b->n_nodes[C.src] > 0 && b->n_edges[C.edge_type] > 0;
    };
    if (need_dx && n->any_agg_first) {
      std::vector<GemmProblem> pp;
      for (int i = 0; i < Y.n_live; ++i) {
        const int c = Y.live[i];
        if (!aggf_live(c)) continue;
        const hmp_conv_spec& C = Ls.convs[c];
        float* want = l > 0 ? n->G[l][C.src] : d_gx[C.src];
        if (!want) continue;
        const ConvLayout& Q = Y.conv[c];
        GemmProblem p;
        memset(&p, 0, sizeof(p));
        p.A = dz_at(C.dst, Q.aoff); p.lda = Y.ncols[C.dst]; p.trans_a = 0; p.a_bf16 = dz16 ? 1 : 0;
        p.B = n->d_packed + Q.wc_off; p.ldb = Y.ldw[C.src]; p.trans_b = 0;
        p.C = Q.dmrows; p.ldc = Q.ld_m;
        p.M = b->n_nodes[C.dst]; p.N = n->dim[l][C.src]; p.K = fpad(C.f_out);
        p.n_real = p.N;
        p.epi = EPI_NONE;
        pp.push_back(p);
      }
      if (!pp.empty()) {
        HMP_TRY(chain_flush(n, st));
        Scope sp(n, KC_GEMM_BWD, st);
        HMP_TRY(gemm_many(pp, false, st, nullptr, n->compute_bf16 != 0));
      }
    }
    // the gradient the source rows of aggregate-first conv c receive: the segment mean's transpose of dM
    struct SrcTerm { int c; const hmp_plan* P; const float* degf; };
    auto src_term = [&](int s, SrcTerm& T) -> bool {
      const int c = Y.aggf_of_src[s];
      if (c < 0 || !aggf_live(c)) return false;
      T.c = c; T.P = &n->plan[Ls.convs[c].edge_type]; T.degf = n->degf[Ls.convs[c].edge_type];
      return true;
    };
    if (need_dx) {  // input gradient, masked by the previous layer's activation/dropout derivative
      HMP_TRY(chain_flush(n, st));
      Scope sc(n, KC_GEMM_BWD, st);
      std::vector<GemmProblem> ps;
      std::vector<int> ps_type;
      for (int s = 0; s < n->T; ++s) {
        if (Y.ncols[s] == 0 || b->n_nodes[s] == 0) continue;
        float* dst = l > 0 ? n->G[l][s] : d_gx[s];
        if (!dst) continue;
        GemmProblem p;
        memset(&p, 0, sizeof(p));
        p.A = n->dZ[l][s]; p.lda = Y.ncols[s]; p.trans_a = 0;
        p.a_bf16 = dz16 ? 1 : 0;
        if (rootless && Y.roff[s] >= 0) { int ldg; p.A2 = g_of(s, ldg); p.lda2 = ldg; p.a_split = Y.roff[s]; }
        p.B = n->d_packed + Y.wp_off[s]; p.ldb = Y.ldw[s]; p.trans_b = 0;
        p.C = dst; p.ldc = l > 0 ? n->ld[l][s] : b->ldx[s];
        p.M = b->n_nodes[s]; p.N = n->dim[l][s]; p.K = Y.ncols[s];
        p.n_real = p.N;
        if (l > 0) {
          const hmp_layer_spec& Lp = S.layers[l - 1];
          p.epi = EPI_ACTMASK;
          p.H = n->H[l][s]; p.ldh = n->ld[l][s]; p.act = Lp.act;
          p.h_bf16 = n->h16[l][s] ? 1 : 0;
          p.drop_on = (n->training && Lp.dropout > 0.f) ? 1 : 0;
          if (p.drop_on) p.drop = make_drop(n, Lp.dropout, (uint32_t)((l - 1) * HMP_MAX_NODE_TYPES + s));
          if (p.act == HMP_ACT_NONE && !p.drop_on) p.epi = EPI_NONE;
        }
        ps.push_back(p);
        ps_type.push_back(s);
      }
      // bf16 compute mode, 10^6-row regime: G[l] is only ever gathered by layer l-1's transposed aggregation -> stored as bf16
      // when this GEMM runs on the bf16 kernel and that aggregation takes its one-wavefront-per-row shape (see forward_impl)
      // (a large source type whose gradient is the masked transpose alone -- no GEMM in this layer -- counts like the GEMM it replaces)
      bool big_alone = false;
      for (int s = 0; s < n->T; ++s) {
        SrcTerm T;
        if (Y.ncols[s] == 0 && b->n_nodes[s] >= 32768 && src_term(s, T)) big_alone = true;
      }
      if (l > 0 && !n->fuse_now && n->lay[l - 1].kind != HMP_CONV_GAT &&
          (gemm_takes_bf16(ps, n->compute_bf16 != 0) || (n->compute_bf16 != 0 && big_alone))) {
        const hmp_layer_spec& Lp = S.layers[l - 1];
        int fmin = 1 << 30, fmax = 0;
        for (int t = 0; t < n->T; ++t) {
          if (n->lay[l - 1].roff[t] < 0 || b->n_nodes[t] == 0) continue;
          const int f = fpad(Lp.out_dim[t]);
          fmin = f < fmin ? f : fmin;
          fmax = f > fmax ? f : fmax;
        }
        bool ok = fmax <= 256 && fmin > 128;
        for (GemmProblem& p : ps) ok = ok && (p.ldc & 3) == 0;
        if (!n->env.z16) ok = false;
        if (ok) {
          for (GemmProblem& p : ps) p.c_bf16 = 1;
          g16[l] = true;
        }
      }
      // source types of aggregate-first convs: + transpose of the segment mean, before the mask.  Fused into the epilogue of the
      // weight-stationary kernel where that one runs the problem; else as an fp32 matrix the tiled kernels start their sums from
      int64_t sadd_used = 0;
      for (size_t i = 0; i < ps.size(); ++i) {
        SrcTerm T;
        if (!src_term(ps_type[i], T)) continue;
        const ConvLayout& Q = Y.conv[T.c];
        GemmProblem& p = ps[i];
        p.g_rowptr = T.P->d_t_rowptr; p.g_col = T.P->d_t_col; p.g_deg = T.degf; p.g_rows = Q.dmrows; p.g_ld = Q.ld_m;
        if (n->compute_bf16 != 0 && gemm_bf16_dx_takes(p, false)) continue;
        p.g_rowptr = nullptr;
        HMP_CHECK_ARG(sadd_used + (int64_t)p.M * Q.ld_m <= n->sadd_floats, "net: gather-add scratch too small");
        float* sm = n->d_sadd + sadd_used;
        sadd_used += (int64_t)p.M * Q.ld_m;
        HMP_TRY(seg_mean_rows_t_launch(Q.dmrows, Q.ld_m, Q.ld_m, T.P->d_t_rowptr, T.P->d_t_col, T.degf, p.M, nullptr, 0, 0, 0, 0, 1.f, sm, Q.ld_m, 0, st));
        p.Cadd = sm; p.ldadd = Q.ld_m;
      }
      if (!ps.empty()) HMP_TRY(gemm_many(ps, false, st, nullptr, n->compute_bf16 != 0));
      // source types with NO stacked columns in this layer (their only live conv is the aggregate-first one): no GEMM -- the
      // masked transpose directly
      for (int s = 0; s < n->T; ++s) {
        SrcTerm T;
        if (Y.ncols[s] != 0 || b->n_nodes[s] == 0 || !src_term(s, T)) continue;
        float* dst = l > 0 ? n->G[l][s] : d_gx[s];
        if (!dst) continue;
        const ConvLayout& Q = Y.conv[T.c];
        const hmp_layer_spec* Lp = l > 0 ? &S.layers[l - 1] : nullptr;
        const bool drop_on = Lp && n->training && Lp->dropout > 0.f;
        const int act = Lp ? Lp->act : HMP_ACT_NONE;
        const bool masked = Lp && (act != HMP_ACT_NONE || drop_on);
        bool gb = false;
        if (l > 0 && ps.empty()) {  // no GEMM decided the storage of G[l]: the same rule (see above)
          const hmp_layer_spec& Lq = S.layers[l - 1];
          int fmin = 1 << 30, fmax = 0;
          for (int t = 0; t < n->T; ++t) {
            if (n->lay[l - 1].roff[t] < 0 || b->n_nodes[t] == 0) continue;
            const int f = fpad(Lq.out_dim[t]);
            fmin = f < fmin ? f : fmin;
            fmax = f > fmax ? f : fmax;
          }
          g16[l] = n->compute_bf16 != 0 && !n->fuse_now && n->lay[l - 1].kind != HMP_CONV_GAT && fmax <= 256 && fmin > 128 &&
                   (n->ld[l][s] & 3) == 0 && n->env.z16 && (b->n_nodes[s] >= 32768 || bf16_all());
        }
        gb = l > 0 && g16[l];
        HMP_TRY(seg_mean_rows_t_launch(Q.dmrows, Q.ld_m, fpad(n->dim[l][s]), T.P->d_t_rowptr, T.P->d_t_col, T.degf, b->n_nodes[s],
                                       masked ? (const void*)n->H[l][s] : nullptr, n->ld[l][s], n->h16[l][s] ? 1 : 0, act, drop_on ? 1 : 0,
                                       drop_on ? 1.f / (1.f - Lp->dropout) : 1.f, dst, l > 0 ? n->ld[l][s] : b->ldx[s], gb ? 1 : 0, st));
      }
    }
    {  // weight + bias gradient: dWp = dZ^T * [H | 1], split over node chunks (+ GAT: bias column sums, d V_edge)
      std::vector<GemmProblem> local_ps;
      std::vector<int> local_ids;
      std::vector<GemmProblem>& ps = n->dw_branch ? local_ps : wps;
      std::vector<int>& ids = n->dw_branch ? local_ids : wids;
      auto add = [&](int sid, const GemmProblem& p) {
        n->dyn.slab_stride[sid] = (int)p.slab_stride;
        ids.push_back(sid);
        ps.push_back(p);
      };
      for (int s = 0; s < n->T; ++s) {
        if (Y.ncols[s] == 0 || b->n_nodes[s] == 0) continue;  // no nodes: zero gradient (n_slabs stays 0)
        GemmProblem p;
        memset(&p, 0, sizeof(p));
        p.A = n->dZ[l][s]; p.lda = Y.ncols[s]; p.trans_a = 1;
        p.a_bf16 = dz16 ? 1 : 0;
        if (rootless && Y.roff[s] >= 0) { int ldg; p.A2 = g_of(s, ldg); p.lda2 = ldg; p.a_split = Y.roff[s]; }
        p.B = h_ptr(n, l, s); p.ldb = h_ld(n, l, s); p.trans_b = 0;
        p.b_bf16 = n->h16[l][s] ? 1 : 0;
        p.C = n->d_slabs + Y.slab_off[s]; p.ldc = Y.lddw[s];
        p.slab_stride = (int64_t)Y.ncols[s] * Y.lddw[s];
        p.M = Y.ncols[s]; p.N = n->dim[l][s] + 1; p.K = b->n_nodes[s];
        p.n_real = n->dim[l][s]; p.aug_ones = 1;
        add(slab_id_w(l, s), p);
      }
      for (int i = 0; i < Y.n_live; ++i) {  // aggregate-first convs: dW_l = dZ_block^T * M (destination-sized)
        const int c = Y.live[i];
        if (!aggf_live(c)) continue;
        const hmp_conv_spec& C = Ls.convs[c];
        const ConvLayout& Q = Y.conv[c];
        GemmProblem p;
        memset(&p, 0, sizeof(p));
        p.A = dz_at(C.dst, Q.aoff); p.lda = Y.ncols[C.dst]; p.trans_a = 1; p.a_bf16 = dz16 ? 1 : 0;
        p.B = Q.mrows; p.ldb = Q.ld_m; p.trans_b = 0;
        p.C = n->d_slabs + Q.cslab_off; p.ldc = Y.lddw[C.src];
        p.slab_stride = (int64_t)fpad(C.f_out) * Y.lddw[C.src];
        p.M = fpad(C.f_out); p.N = n->dim[l][C.src]; p.K = b->n_nodes[C.dst];
        p.n_real = p.N; p.aug_ones = 0;
        add(slab_id_v(l, c), p);
      }
      if (Y.kind == HMP_CONV_GAT) {
        for (int t = 0; t < n->T; ++t) {  // bias: column sums of the output gradient
          if (Y.n_in[t] == 0 || b->n_nodes[t] == 0) continue;
          int ldg;
          const float* g = g_of(t, ldg);
          GemmProblem p;
          memset(&p, 0, sizeof(p));
          p.A = g; p.lda = ldg; p.trans_a = 1;
          p.B = g; p.ldb = ldg; p.trans_b = 0;  // no real columns are read (n_real = 0): only the ones column
          p.C = n->d_slabs + Y.bslab_off[t]; p.ldc = 4;
          p.slab_stride = (int64_t)fpad(Ls.out_dim[t]) * 4;
          p.M = Ls.out_dim[t]; p.N = 1; p.K = b->n_nodes[t];
          p.n_real = 0; p.aug_ones = 1;
          add(slab_id_b(l, t), p);
        }
        for (int i = 0; i < Y.n_live; ++i) {  // d V_edge = edge_attr^T * d logit (original edge order)
          const int c = Y.live[i];
          const hmp_conv_spec& C = Ls.convs[c];
          if (C.edge_dim == 0 || b->n_edges[C.edge_type] == 0) continue;
          GemmProblem p;
          memset(&p, 0, sizeof(p));
          p.A = b->d_edge_attr[C.edge_type]; p.lda = C.edge_dim; p.trans_a = 1;
          p.B = Y.conv[c].dlogit_orig; p.ldb = GAT_HMAX; p.trans_b = 0;
          p.C = n->d_slabs + Y.conv[c].vslab_off; p.ldc = GAT_HMAX;
          p.slab_stride = GAT_MAX_EDIM * GAT_HMAX;
          p.M = C.edge_dim; p.N = C.heads; p.K = (int)b->n_edges[C.edge_type];
          p.n_real = C.heads;
          add(slab_id_v(l, c), p);
        }
      }
      if (n->dw_branch) {
        HMP_TRY(chain_flush(n, st));
        Scope sc(n, KC_GEMM_BWD, wst);
        std::vector<int> ks;
        HMP_TRY(gemm_many(ps, true, wst, &ks, n->compute_bf16 != 0));
        for (size_t i = 0; i < ids.size(); ++i) n->dyn.n_slabs[ids[i]] = (unsigned char)ks[i];
      }
    }
  }
  HMP_TRY(chain_run(n, st));  // the deferred aggregation launches: as one graph-local launch, or in order
  if (n->dw_branch) {
    if (n->use_branches) HMP_TRY(fork_to(n, n->side[1], st));  // join the weight-gradient branch
  } else {
    Scope sc(n, KC_GEMM_BWD, st);
    std::vector<int> ks;
    bool direct = n->fuse_now;
    if (direct) {  // register-direct TN kernel (one memory round trip per <= 192-node chunk); all-or-nothing per call
      if (!n->env.tn_direct) direct = false;
      // weight gradients of >= 10^9 multiply-adds (GAT at its batch size): split-K on the bf16 matrix pipe by the three-way split
      if (direct && !n->compute_bf16 && !wps.empty() && gemm_x3_split_takes(wps.data(), (int)wps.size())) direct = false;
      for (size_t i = 0; i < wps.size() && direct; ++i)
        if (wps[i].K > TN_DIRECT_SLABS * 384 || wps[i].b_bf16 || wps[i].a_bf16) direct = false;
      for (size_t base = 0; base < wps.size() && direct; base += GEMM_MAX_PROB) {
        TnBatch tb;
        memset(&tb, 0, sizeof(tb));
        const size_t cnt = (wps.size() - base) < (size_t)GEMM_MAX_PROB ? (wps.size() - base) : (size_t)GEMM_MAX_PROB;
        for (size_t i = 0; i < cnt; ++i) {
          const GemmProblem& g = wps[base + i];
          TnProblem& P = tb.p[tb.n++];
          P.A = g.A; P.B = g.B; P.C = g.C; P.slab_stride = g.slab_stride;
          P.M = g.M; P.N = g.N; P.K = g.K; P.lda = g.lda; P.ldb = g.ldb; P.ldc = g.ldc;
          P.n_real = g.n_real; P.aug_ones = g.aug_ones;
        }
        int ok = 0;
        HMP_TRY(gemm_tn_direct_launch(tb, TN_DIRECT_SLABS, &ok, st));
        HMP_CHECK_ARG(ok, "net: weight-gradient problem too deep for the direct kernel after the size check");
        for (size_t i = 0; i < cnt; ++i) ks.push_back(tb.p[i].ksplit);
      }
    }
    if (!direct) HMP_TRY(gemm_many(wps, true, st, &ks, n->compute_bf16 != 0));
    for (size_t i = 0; i < wids.size(); ++i) n->dyn.n_slabs[wids[i]] = (unsigned char)ks[i];
  }
  {
    Scope sc(n, KC_GRAD_REDUCE, st);
    const int64_t na = n->spec.n_active_params;
    // Adam inside the un-pack: needs the valid-label count before the kernel starts (finalised by the first backward
    // kernel) and gradient terms that read no parameters (no GAT layer)
    AdamFuse af = n->adam_fuse;
    af.on = (af.on && fin_early && !n->any_gat) ? 1 : 0;
    n->adam_done = af.on != 0;
    HMP_TRY(grad_reduce_launch(n->d_grad_segs, n->grad_sb, n->dyn, n->d_slabs, d_params, d_grads,
                               n->fin_loss ? n->d_row_lv : nullptr, b->n_out, d_grads + na, n->d_state, af, st));
    n->fin_loss = false;
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
  if (r == HMP_OK) {
    // step counter + status bits live outside the workspace: a re-bind (a larger batch arrived) must not restart Adam's t or
    // the dropout sequence, nor lose status bits
    if (hipMalloc(&n->d_state, 256) != hipSuccess || hipMemset(n->d_state, 0, 256) != hipSuccess) {
      snprintf(err_buf(), 512, "hmp_net_create: device allocation of the net state failed");
      r = HMP_E_HIP;
    }
    n->d_step = n->d_state ? &n->d_state->step : nullptr;
  }
  if (r == HMP_OK) {
    // side-stream branches (pack + layer-0 projection next to the plan, weight gradients next to the backward chain) only
    // on request: measured on MI355X, every fork/join costs ~10 us of dependency latency, more than the overlap buys once
    // the plan is a single launch
    const char* nb = getenv("HMP_BRANCH");
    n->branch_mask = nb ? atoi(nb) : 0;
    bool ok = n->branch_mask != 0;
    for (int i = 0; i < 2 && ok; ++i) ok = hipStreamCreateWithFlags(&n->side[i], hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 32 && ok; ++i) {
      ok = hipEventCreateWithFlags(&n->evs[i], hipEventDisableTiming) == hipSuccess;
      if (ok) n->n_evs = i + 1;
    }
    n->use_branches = ok && n->n_evs == 32;
    const char* db = getenv("HMP_DW_BRANCH");
    n->dw_mode = db ? (db[0] == '1' ? 1 : 0) : -1;  // -1: decide per batch (see backward_impl)
    const char* fz = getenv("HMP_FUSE");
    n->fuse_mode = fz ? (fz[0] == '1' ? 1 : 0) : -1;
#ifdef HMP_EXPERIMENTS
    const char* cz = getenv("HMP_CHAIN");  // graph-local chain launch: measured 5x slower (profiles/r02_c_graph_local_chain.md)
    n->chain_mode = cz ? (cz[0] == '1' ? 1 : 0) : -1;
#else
    n->chain_mode = 0;
#endif
  }
  if (r != HMP_OK) {
    hmp_net_destroy(n);
    return r;
  }
  *out = n;
  return HMP_OK;
}

extern "C" void hmp_net_destroy(hmp_net* n) {
  if (!n) return;
  for (auto& r : n->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (int i = 0; i < n->n_evs; ++i) (void)hipEventDestroy(n->evs[i]);
  for (int i = 0; i < 2; ++i)
    if (n->side[i]) (void)hipStreamDestroy(n->side[i]);
  if (n->d_state) (void)hipFree(n->d_state);
  if (n->chain_h) (void)hipHostFree(n->chain_h);
  if (n->chain_stage) (void)hipHostFree(n->chain_stage);
  if (n->d_chain) (void)hipFree(n->d_chain);
  if (n->chain_copied) (void)hipEventDestroy(n->chain_copied);
  if (n->d_pack_segs) (void)hipFree(n->d_pack_segs);
  if (n->d_pack_map) (void)hipFree(n->d_pack_map);
  if (n->d_grad_segs) (void)hipFree(n->d_grad_segs);
  for (int l = 0; l < HMP_MAX_LAYERS; ++l)
    if (l < n->L && n->lay[l].d_gat) (void)hipFree(n->lay[l].d_gat);
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
  n->ws_base = (char*)d_workspace;
  n->ws_bytes = need;
  // zero once: padding columns of every buffer stay zero for the lifetime of the binding
  HMP_HIP(hipMemset(d_workspace, 0, need));
  HMP_TRY(build_gat_tables(n));
  if (n->any_agg_first) {  // identity lists of the aggregate-first blocks
    std::vector<int> io((size_t)n->iota_cap + 1);
    for (size_t i = 0; i < io.size(); ++i) io[i] = (int)i;
    std::vector<float> on((size_t)n->iota_cap, 1.0f);
    HMP_HIP(hipMemcpy(n->d_iota, io.data(), io.size() * sizeof(int), hipMemcpyHostToDevice));
    HMP_HIP(hipMemcpy(n->d_ones, on.data(), on.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  n->bound = true;
  n->have_fwd = false;
  n->plan_ok = false;
  n->ell_ok = false;
  return HMP_OK;
}

extern "C" int hmp_net_forward(hmp_net* n, const hmp_batch* batch, const float* d_params, int32_t training, uint64_t seed,
                               uint32_t rng_step, const float** d_out, int32_t* ld_out, void* stream) {
  HMP_CHECK_ARG(n && batch && d_params && d_out && ld_out, "hmp_net_forward: null argument");
  read_env(n);
  g_bf16_all = n->env.bf16_all;
  n->training = training; n->seed = seed; n->rng_step = rng_step; n->step_dev = false;
  n->chain_try = false;
  n->deferred.clear();
  HMP_TRY(forward_impl(n, batch, d_params, (hipStream_t)stream));
  *d_out = out_ptr(n);
  *ld_out = n->out_ld;
  return HMP_OK;
}

extern "C" int hmp_net_backward(hmp_net* n, const float* d_gout, int32_t ld_gout, const float* d_params, float* d_grads,
                                float* const* d_gx, void* stream) {
  HMP_CHECK_ARG(n && d_gout && d_grads && d_params, "hmp_net_backward: null argument");
  g_bf16_all = n->env.bf16_all;  // (the switches of the forward this backward belongs to)
  return backward_impl(n, d_gout, ld_gout, d_grads, d_params, d_gx, (hipStream_t)stream);
}

extern "C" int hmp_net_aux_output(hmp_net* n, const float** d_out, int32_t* ld_out, int32_t* n_rows) {
  HMP_CHECK_ARG(n && d_out && ld_out && n_rows, "hmp_net_aux_output: null argument");
  HMP_CHECK_ARG(n->spec.aux_readout_type >= 0 && n->have_fwd, "hmp_net_aux_output: needs aux_readout_type and a forward");
  const int at = n->spec.aux_readout_type;
  *d_out = n->H[n->L][at];
  *ld_out = n->ld[n->L][at];
  *n_rows = n->batch.n_nodes[at];
  return HMP_OK;
}

extern "C" int hmp_net_hidden(hmp_net* n, int32_t layer, int32_t node_type, const void** d_h, int32_t* ld, int32_t* n_rows,
                              int32_t* width, int32_t* is_bf16) {
  HMP_CHECK_ARG(n && d_h && ld && n_rows && width && is_bf16, "hmp_net_hidden: null argument");
  HMP_CHECK_ARG(n->have_fwd, "hmp_net_hidden: needs a forward");
  HMP_CHECK_ARG(layer >= 1 && layer <= n->L && node_type >= 0 && node_type < n->T && n->H[layer][node_type] != nullptr,
                "hmp_net_hidden: layer %d / node type %d has no stored output", layer, node_type);
  *d_h = n->H[layer][node_type];
  *ld = n->ld[layer][node_type];
  *n_rows = n->batch.n_nodes[node_type];
  *width = n->dim[layer][node_type];
  *is_bf16 = n->h16[layer][node_type] ? 1 : 0;
  return HMP_OK;
}

extern "C" int hmp_net_backward2(hmp_net* n, const float* d_gout, int32_t ld_gout, const float* d_gout_aux, int32_t ld_aux,
                                 const float* d_params, float* d_grads, float* const* d_gx, void* stream) {
  HMP_CHECK_ARG(n && d_grads && d_params && (d_gout || d_gout_aux), "hmp_net_backward2: null argument");
  HMP_CHECK_ARG(n->spec.aux_readout_type >= 0, "hmp_net_backward2: the net has one output (use hmp_net_backward)");
  g_bf16_all = n->env.bf16_all;
  if (!d_gout) {  // no gradient for the first output: a zero block of its shape (G[L][readout] is free in the last layer)
    HMP_CHECK_ARG(n->have_fwd, "net: backward without a forward");
    const int rt = n->spec.readout_type, rows = n->batch.n_nodes[rt];
    if (rows > 0) HMP_HIP(hipMemsetAsync(n->G[n->L][rt], 0, (size_t)rows * n->ld[n->L][rt] * 4, (hipStream_t)stream));
    d_gout = n->G[n->L][rt];
    ld_gout = n->ld[n->L][rt];
  }
  return backward_impl(n, d_gout, ld_gout, d_grads, d_params, d_gx, (hipStream_t)stream, d_gout_aux, ld_aux);
}

extern "C" int hmp_net_step_fwd_bwd(hmp_net* n, const hmp_batch* batch, const float* d_params, float* d_grads,
                                    const hmp_train_args* args, void* stream) {
  HMP_CHECK_ARG(n && batch && d_params && d_grads && args, "hmp_net_step_fwd_bwd: null argument");
  HMP_CHECK_ARG(n->spec.aux_readout_type < 0, "hmp_net_step_fwd_bwd: the fused step computes one cross entropy (single-output nets)");
  HMP_CHECK_ARG(batch->d_labels != nullptr, "hmp_net_step_fwd_bwd: labels required");
  hipStream_t st = (hipStream_t)stream;
  read_env(n);
  g_bf16_all = n->env.bf16_all;
  n->training = args->training; n->seed = args->seed; n->rng_step = 0; n->step_dev = true;
  n->d_step = args->d_step ? args->d_step : &n->d_state->step;
  n->ce_labels = batch->d_labels; n->ce_ignored = args->ignored_label; n->ce_done = false;
  // graph-local chain: a batch that says where its graphs begin (Batch.ptr), SAGE stacks, no pooled readout.
  n->deferred.clear();
  // MEASURED SLOWER than the multi-launch sequence (profiles/r02_c_graph_local_chain.md: 0.239 ms against 0.048 ms for the five launches
  // it replaces on config 2): a graph's 7 tiles take 4 rounds of 2 per phase and a round is one full dependent-load chain, whereas the
  // multi-launch sequence runs every tile of every graph at once.  Opt-in only (HMP_CHAIN=1), kept for its test and as the record.
  n->chain_try = n->chain_mode == 1 && !n->any_gat && n->spec.pool_edge_type < 0 && n->L <= CHAIN_MAX_LAYERS && batch->n_graphs > 0 &&
                 batch->max_graph_nodes > 0 && batch->max_graph_nodes <= 4096 && !n->use_branches;
  const int rf = forward_impl(n, batch, d_params, st);
  n->ce_labels = nullptr;
  if (rf != HMP_OK) { n->chain_try = false; n->deferred.clear(); }
  HMP_TRY(rf);
  if (!n->ce_done) {
    HMP_TRY(chain_flush(n, st));
    Scope sc(n, KC_LOSS, st);
    HMP_TRY(masked_ce_rows_launch(out_ptr(n), n->out_ld, batch->n_out, n->out_dim, batch->d_labels, args->ignored_label, n->d_gout,
                                  n->out_ld, n->d_row_lv, n->d_state, st));
  }
  // {loss_sum, count} -> d_grads[na], d_grads[na + 1]: by the first transposed aggregation, else by the gradient un-pack
  n->fin_loss = true;
  const int rb = backward_impl(n, n->d_gout, n->out_ld, d_grads, d_params, nullptr, st);
  if (rb != HMP_OK) { n->chain_try = false; n->deferred.clear(); }
  n->d_step = &n->d_state->step;  // the optimiser's counter belongs to the caller: not kept beyond this call
  n->step_dev = false;
  return rb;
}

extern "C" int hmp_net_step_fused(hmp_net* n, const hmp_batch* batch, float* d_params, float* d_grads, float* d_m, float* d_v,
                                  const hmp_train_args* args, void* stream) {
  HMP_CHECK_ARG(n && batch && d_params && d_grads && d_m && d_v && args, "hmp_net_step_fused: null argument");
  HMP_CHECK_ARG(n->spec.aux_readout_type < 0, "hmp_net_step_fused: the fused step computes one cross entropy (single-output nets)");
  AdamFuse& af = n->adam_fuse;
  af.on = n->fuse_mode == 0 ? 0 : 1;
  af.p = d_params; af.m = d_m; af.v = d_v;
  af.lr = args->lr; af.b1 = args->beta1; af.b2 = args->beta2; af.eps = args->eps; af.wd = args->weight_decay;
  af.step_dev = args->d_step ? args->d_step : &n->d_state->step;
  af.count = d_grads + n->spec.n_active_params + 1;
  n->adam_done = false;
  const int r = hmp_net_step_fwd_bwd(n, batch, d_params, d_grads, args, stream);
  af.on = 0;
  HMP_TRY(r);
  if (n->adam_done) return HMP_OK;
  return hmp_net_step_adam(n, d_params, d_grads, d_m, d_v, args, stream);
}

extern "C" int hmp_net_step_adam(hmp_net* n, float* d_params, const float* d_grads, float* d_m, float* d_v,
                                 const hmp_train_args* args, void* stream) {
  HMP_CHECK_ARG(n && d_params && d_grads && d_m && d_v && args, "hmp_net_step_adam: null argument");
  HMP_CHECK_ARG(n->bound, "hmp_net_step_adam: workspace not bound");
  hipStream_t st = (hipStream_t)stream;
  const int64_t na = n->spec.n_active_params;
  Scope sc(n, KC_ADAM, st);
  // t = the step counter the pack kernel bumped at the head of this step
  return adam_launch(d_params, d_grads, d_m, d_v, na, args->lr, args->beta1, args->beta2, args->eps, args->weight_decay, 0,
                     args->d_step ? args->d_step : &n->d_state->step, d_grads + na + 1, st);
}

extern "C" int hmp_net_read_state(hmp_net* n, int32_t* step, int32_t* status, void* stream) {
  HMP_CHECK_ARG(n && n->d_state, "hmp_net_read_state: null net");
  NetState h;
  HMP_HIP(hipStreamSynchronize((hipStream_t)stream));
  HMP_HIP(hipMemcpy(&h, n->d_state, sizeof(h), hipMemcpyDeviceToHost));
  // the counter of the last step (its optimiser's, hmp_train_args::d_step, or the net's own) was mirrored into the net's state by
  // the kernel that bumped it: the caller's pointer is not kept beyond the call that passed it
  if (step) *step = h.last_step;
  if (status) *status = h.status;
  return HMP_OK;
}

extern "C" int hmp_net_set_compute(hmp_net* n, int32_t bf16) {
  HMP_CHECK_ARG(n, "hmp_net_set_compute: null net");
  HMP_CHECK_ARG(bf16 == 0 || bf16 == 1, "hmp_net_set_compute: mode %d (0 = fp32, 1 = bf16 MFMA for large GEMMs)", bf16);
  n->compute_bf16 = bf16;
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
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  n->recs.clear();
  return HMP_OK;
}
