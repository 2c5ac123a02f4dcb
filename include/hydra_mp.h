/*
 * hydra_mp.h -- C ABI of libhydra_mp.so, the MI355X (gfx950) message-passing engine that replaces the
 * torch_geometric layer stack on Hydra-GNN's room-classification hot path.
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer (HBM) unless stated; sizes are element counts.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises.
 *   - return value: 0 = OK, otherwise an HMP_E_* code; hmp_last_error() gives the message
 *     (thread local).  No entry point falls back to a CPU path.
 *   - features are fp32 row-major with an explicit leading dimension (floats); graph indices are
 *     int32 in the plan, int64 `edge_index[2][E]` on input (the PyG contract: row 0 = source,
 *     row 1 = destination, aggregation at the destination).
 *
 * Each group cites the reference interface it replaces (paths relative to the reference repo;
 * "[PyG]" = torch_geometric 2.3.1 operator the reference calls, restated in SURVEY.md Appendix A).
 */
#ifndef HYDRA_MP_H
#define HYDRA_MP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: round 2 -- hmp_batch grew (plan_valid, d_node_ptr, n_graphs, max_graph_nodes, d_edge_ptr), hmp_train_args::d_step; sections 10-12
 * 3: round 3 -- hmp_comm_query; hmp_net_read_state reports the counter of the last step from the net's own state;
 *    hmp_conv_spec::agg_first; hmp_gemm_bf16_dx / hmp_gemm_bf16_dw */
#define HMP_ABI_VERSION 3

#define HMP_OK 0
#define HMP_E_ARG 1      /* bad argument (shape / alignment / capacity) */
#define HMP_E_HIP 2      /* a HIP runtime call failed */
#define HMP_E_STATE 3    /* call order violated (e.g. backward without forward) */
#define HMP_E_NODEVICE 4 /* no gfx950 device visible */
#define HMP_E_UNSUPPORTED 5 /* an optional system library (RCCL) is not present on this machine */

#define HMP_MAX_NODE_TYPES 8
#define HMP_MAX_EDGE_TYPES 16
#define HMP_MAX_LAYERS 8
#define HMP_MAX_CONVS 16 /* per layer */

/* ---------------------------------------------------------------------------------------------
 * 0. library
 * ------------------------------------------------------------------------------------------- */
int hmp_abi_version(void);
const char* hmp_last_error(void);
/* sizeof() of the ABI structs: 0 hmp_plan, 1 hmp_gat_args, 2 hmp_conv_spec, 3 hmp_layer_spec, 4 hmp_net_spec,
 * 5 hmp_batch, 6 hmp_train_args (lets a foreign-language binding verify its struct mirror) */
size_t hmp_sizeof(int which);
/* number of visible devices whose gcnArchName starts with "gfx950"; never raises */
int hmp_device_count(void);

/* ---------------------------------------------------------------------------------------------
 * 1. graph plan: CSR by destination + CSC (transpose) by source, stable in edge order.
 *    Replaces the per-call `index_select` / `scatter` index handling of
 *    [PyG] MessagePassing.propagate (called from models/utils.py:14,49 via HeteroConv).
 *    Bit-exact contract: col[rowptr[i] .. rowptr[i+1]) are the sources of the edges whose
 *    destination is i, in ascending original edge id (== torch.sort(dst, stable=True)).
 * ------------------------------------------------------------------------------------------- */
typedef struct hmp_plan {
  int32_t n_src, n_dst;
  int64_t n_edges;
  int32_t* d_rowptr;   /* [n_dst+1]  CSR by destination */
  int32_t* d_col;      /* [E] source node of the k-th CSR entry */
  int32_t* d_eid;      /* [E] original edge id of the k-th CSR entry */
  int32_t* d_t_rowptr; /* [n_src+1]  CSC (edges grouped by source) */
  int32_t* d_t_col;    /* [E] destination node of the k-th CSC entry */
  int32_t* d_t_pos;    /* [E] CSR position of the k-th CSC entry (eid = d_eid[d_t_pos[k]]) */
} hmp_plan;

/* bytes of scratch hmp_plan_build needs for E edges between n_src and n_dst nodes */
size_t hmp_plan_scratch_bytes(int64_t n_edges, int32_t n_src, int32_t n_dst);
/* d_edge_index: int64 [2][E] (row stride = E).  Out-of-range endpoints set bit 0 of *d_status
 * (int32, device, may be NULL) and the edge is dropped from the lists (its slot keeps eid = -1). */
int hmp_plan_build(const int64_t* d_edge_index, hmp_plan plan, void* d_scratch, int32_t* d_status, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 2. K1 -- segment mean over CSR rows (SAGE neighbour mean, LeafPool).
 *    Replaces [PyG] SAGEConv.propagate(aggr="mean") = index_select + scatter_add x2 + clamp + div
 *     and LeafPool (models/heterogeneous_neural_tree_network.py:18-31).
 *    out[i, 0:F] = (1/max(deg_i,1)) * sum_{k in row i} x[col[k], 0:F]     (zeros for empty rows)
 *    bwd: g_x[j, 0:F] = sum_{k in CSC row j} g_out[t_col[k], 0:F] / max(deg(t_col[k]),1)
 * ------------------------------------------------------------------------------------------- */
int hmp_segment_mean_fwd(const float* d_x, int32_t ldx, int32_t F, hmp_plan plan, float* d_out, int32_t ldo, void* stream);
int hmp_segment_mean_bwd(const float* d_gout, int32_t ldg, int32_t F, hmp_plan plan, float* d_gx, int32_t ldgx, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 3. K2 -- dense fp32 projection on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 FMA chain).
 *    Replaces F.linear inside [PyG] SAGEConv.lin_l / lin_r and GATConv.lin_src / lin_dst.
 *    C[M,N] = op(A)[M,K] * op(B)[K,N];  trans_a: A stored [K][M];  trans_b: B stored [N][K]
 *    (trans_b = 1 is nn.Linear's weight layout).
 * ------------------------------------------------------------------------------------------- */
/* same problem on the bf16 matrix pipe (operands rounded to bf16, fp32 accumulate); unit-test entry of gemm_bf16.hip */
int hmp_gemm_bf16(const float* d_a, int32_t lda, int32_t trans_a, const float* d_b, int32_t ldb, int32_t trans_b, float* d_c,
                  int32_t ldc, int32_t M, int32_t N, int32_t K, void* stream);
/* C[M, N] = A[M, K] * W[N, K]^T with A STORED as bf16 (the hidden activations of bf16 compute mode, lda in elements, rows 8-byte
 * aligned), W fp32 (rounded to bf16 on the way in), fp32 accumulation, C written as bf16 (c_bf16 != 0, ldc in elements) or fp32.
 * Tall problems (M >= 32768, K <= 256, 16-byte aligned rows) run on the weight-stationary kernel (csrc/gemm_bf16.hip); this is
 * the projection of the reference's SAGEConv.lin_l / lin_r at 10^6 nodes ([PyG] sage_conv.py: `self.lin_l(out)`). */
int hmp_gemm_bf16_a16(const uint16_t* d_a, int32_t lda, const float* d_w, int32_t ldw, void* d_c, int32_t ldc, int32_t c_bf16,
                      int32_t M, int32_t N, int32_t K, void* stream);
int hmp_gemm_f32(const float* d_a, int32_t lda, int32_t trans_a, const float* d_b, int32_t ldb, int32_t trans_b,
                 float* d_c, int32_t ldc, int32_t M, int32_t N, int32_t K, void* stream);
/* The two backward products of that projection in the 10^6-node regime (csrc/gemm_bf16_bwd.hip; unit-test entries of what the
 * executor calls; reference: autograd of F.linear inside [PyG] SAGEConv, models/utils.py:14).  dZ is bf16 [M][lddz]; its columns
 * from `split` on may live in a second bf16 matrix d_dz2 [M][lddz2] (the root block of dZ is the output gradient itself; split = 0:
 * one matrix; otherwise a multiple of 256).
 *   dx: G[M, N] = mask . (dZ[M, K] * W[K, N]), W fp32 [K][ldw] rounded to bf16, fp32 accumulation, G written as bf16; mask from the
 *       stored activations d_h (bf16 [M][ldh], NULL = none): act' of HMP_ACT_*, and with drop_on an element stored as -0 is a dropped
 *       one (factor 0), kept ones are scaled by drop_scale.  Tall problems (M >= 32768, K in {256, 512, 768}, N % 128 == 0) run on
 *       the weight-stationary kernel.
 *   dw: slabs of dW[Mw, F + 1] = dZ[nodes, Mw]^T * [H | 1][nodes, F + 1]: slab z (fp32 [Mw][ldc] at d_slabs + z * slab_stride) holds
 *       the sum over one node range, *n_slabs of them are written (<= max_slabs); H is bf16 or fp32 [nodes][ldh].  Mw % 256 == 0,
 *       F == 256 and nodes >= 65536 run on the output-stationary kernel. */
int hmp_gemm_bf16_dx(const uint16_t* d_dz, int32_t lddz, const uint16_t* d_dz2, int32_t lddz2, int32_t split, const float* d_w,
                     int32_t ldw, const uint16_t* d_h, int32_t ldh, int32_t act, int32_t drop_on, float drop_scale, uint16_t* d_g,
                     int32_t ldg, int32_t M, int32_t N, int32_t K, void* stream);
int hmp_gemm_bf16_dw(const uint16_t* d_dz, int32_t lddz, const uint16_t* d_dz2, int32_t lddz2, int32_t split, const void* d_h,
                     int32_t ldh, int32_t h_bf16, float* d_slabs, int32_t ldc, int64_t slab_stride, int32_t max_slabs,
                     int32_t* n_slabs, int32_t Mw, int32_t F, int32_t nodes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 4. K3 -- GAT edge softmax + weighted aggregation, one row group of lanes per destination row.
 *    Replaces [PyG] GATConv.edge_update + softmax + message/aggregate for ONE conv.
 *    Inputs: h_src [n_src, H*Cp] projected source rows, head h at columns h*Cp (Cp = channels rounded up to 4);
 *    a_src [n_src, >=H] (ld lda_src), a_dst [n_dst, >=H] (ld lda_dst): attention logits halves;
 *    optional edge term <edge_attr[e, 0:edge_dim], v_edge[0:edge_dim, h]> with v_edge [edge_dim][8]
 *    (edge_attr in ORIGINAL edge order; NULL / edge_dim 0 = none).
 *    e_k = leaky_relu(a_src[col_k,h] + a_dst[i,h] + edge term, 0.2); alpha = softmax over the row
 *    (max-subtracted, denominator + 1e-16); out[i, h*C + c] = sum_k dropout(alpha_k,h) * h_src[col_k, h*Cp + c].
 *    With self_loops != 0 entries with col == row are skipped and one loop i->i (edge term 0) is appended per
 *    row i < min(n_src, n_dst).  The softmax is computed online; the per-row running max and denominator are
 *    written to smax / sden [n_dst, 8] for the backward (alpha is recomputed there, never stored in forward).
 *    These two unit entry points upload a small descriptor and SYNCHRONISE the stream (test / integration use;
 *    the network executor below drives the same kernels without synchronising).
 * ------------------------------------------------------------------------------------------- */
typedef struct hmp_gat_args {
  int32_t heads, channels;   /* H <= 8, C <= 256 */
  int32_t self_loops;
  int32_t edge_dim;          /* 0..4 */
  float dropout_p;           /* attention dropout; 0 = off */
  uint64_t seed;             /* dropout RNG (counter hash, csrc/common.h): element (pos, h) of an [E + n_loop, 8] tensor, */
  uint32_t rng_stream, rng_step; /* pos = CSR position of the edge, loops at E + i (see hmp_dropout_mask) */
} hmp_gat_args;

int hmp_gat_fwd(const float* d_h_src, int32_t ldh, const float* d_a_src, int32_t lda_src, const float* d_a_dst,
                int32_t lda_dst, const float* d_edge_attr, const float* d_v_edge, hmp_plan plan, hmp_gat_args args,
                float* d_smax, float* d_sden, float* d_out, int32_t ldo, void* stream);
/* d_gout [n_dst, H*C].  Outputs: g_h_src [n_src, H*Cp] (ld ldgh), g_a_src [n_src, >=H] (ld ldgas), g_a_dst
 * [n_dst, >=H] (ld ldgad); d_alpha_drop / d_dlogit [E + n_loop, 8] receive alpha after dropout and d loss / d raw
 * logit per CSR position; d_dlogit_orig [E, 8] (may be NULL) the same in original edge order, so that
 * d v_edge = edge_attr^T * dlogit_orig. */
int hmp_gat_bwd(const float* d_gout, int32_t ldg, const float* d_h_src, int32_t ldh, const float* d_a_src, int32_t lda_src,
                const float* d_a_dst, int32_t lda_dst, const float* d_edge_attr, const float* d_v_edge, hmp_plan plan,
                hmp_gat_args args, const float* d_smax, const float* d_sden, float* d_alpha_drop, float* d_dlogit,
                float* d_dlogit_orig, float* d_g_h_src, int32_t ldgh, float* d_g_a_src, int32_t ldgas, float* d_g_a_dst,
                int32_t ldgad, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 5. loss + optimiser
 *    masked cross entropy: models/utils.py:143-148 with mask = (label != ignored_label)
 *    (base_training_job.py:213-214).  Writes d_out2 = {sum of -log p[label] over valid rows, number of
 *    valid rows}; d_grad [n, ldg] = (softmax - onehot) for valid rows, 0 otherwise, i.e. the gradient
 *    of the SUM loss -- callers scale by 1/count (so that N-GPU averaging is count-weighted).
 *    Adam: torch.optim.Adam(lr, weight_decay) with coupled L2 (base_training_job.py:181-185):
 *    g = grad*grad_scale + wd*p; m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g;
 *    p -= (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps).
 * ------------------------------------------------------------------------------------------- */
int hmp_masked_ce(const float* d_logits, int32_t ldl, int32_t n_rows, int32_t n_classes, const int64_t* d_labels,
                  int64_t ignored_label, float* d_grad, int32_t ldg, float* d_out2, void* stream);
/* out[r] = argmax_c x[r, c] (first maximum): the `.argmax(dim=1)` of the inference loop
 * (bin/room_classification_server:286) on the executor's output, so only the labels cross PCIe */
int hmp_argmax_rows(const float* d_x, int32_t ldx, int32_t n_rows, int32_t n_cols, int64_t* d_out, void* stream);
/* d_count: device float holding the valid-label count (grad_scale = 1/max(count,1)); NULL => grad_scale = 1 */
int hmp_adam_flat(float* d_p, const float* d_g, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int32_t step, const float* d_count, void* stream);

/* keep-mask generator of the engine's feature dropout, exposed so tests can replay it in the oracle:
 * element (row, col) of a [n_rows, F] tensor is kept iff mask[row*F+col] != 0. */
int hmp_dropout_mask(uint64_t seed, uint32_t rng_step, uint32_t rng_stream, float p, int32_t n_rows, int32_t F,
                     uint8_t* d_mask, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 6. network executor -- the whole HeteroConv layer stack as one native program.
 *    Replaces HeterogeneousNetwork.forward (models/heterogeneous_network.py:99-122),
 *    HeterogeneousNeuralTreeNetwork.forward (models/heterogeneous_neural_tree_network.py:154-185),
 *    HomogeneousNetwork.forward SAGE/GAT branches (models/homogeneous_network.py:122-139) and, for the
 *    fused step, the loop body of BaseTrainingJob.train (base_training_job.py:202-216).
 * ------------------------------------------------------------------------------------------- */
#define HMP_CONV_SAGE 0
#define HMP_CONV_GAT 1
#define HMP_ACT_NONE 0
#define HMP_ACT_RELU 1
#define HMP_ACT_ELU 2

typedef struct hmp_conv_spec {
  int32_t kind;       /* HMP_CONV_* */
  int32_t edge_type;  /* index into hmp_batch edge arrays */
  int32_t src, dst;   /* node type indices */
  int32_t f_out;      /* SAGE: out width; GAT: channels C per head */
  int32_t heads, concat, self_loops, edge_dim; /* GAT only */
  int32_t fill_mean;  /* GAT_edge self-loop attr: 1 = per-destination mean, 0 = zeros */
  int32_t shared_lin; /* GAT built from an int in_channels: lin_dst IS lin_src */
  int32_t active;     /* 0 = output never reaches the loss: skipped in forward and backward */
  int32_t agg_first;  /* SAGE, src != dst: 1 = aggregate the source rows first, then project the (few) destination rows -- the order
                       * [PyG] SAGEConv itself uses (lin_l(mean_j x_j)); exact algebra, chosen by the caller for
                       * convs whose destination type is much smaller than the source type (objects -> rooms at 10^6 objects).
                       * At most one such conv per source type and layer.  0: project every source row, gather the projected rows */
  float att_dropout;  /* GAT: dropout on the attention coefficients (GATConv(dropout=p)), training only */
  /* offsets (floats) into the flat parameter buffer, -1 = absent.
   * SAGE: w0 = lin_l.weight [f_out, f_src], b0 = lin_l.bias [f_out], w1 = lin_r.weight [f_out, f_dst]
   * GAT : w0 = lin_src.weight [H*C, f_src], w1 = lin_dst.weight [H*C, f_dst], a0 = att_src [H*C],
   *       a1 = att_dst [H*C], w2 = lin_edge.weight [H*C, edge_dim], a2 = att_edge [H*C], b0 = bias */
  int64_t w0, w1, w2, a0, a1, a2, b0;
} hmp_conv_spec;

typedef struct hmp_layer_spec {
  int32_t n_convs;
  int32_t act;         /* activation applied to this layer's output (HMP_ACT_NONE on the last) */
  float dropout;       /* feature dropout after the activation (training only) */
  int32_t group_mean;  /* HeteroConv aggr: 0 = sum, 1 = mean over the convs reaching a node type */
  int32_t out_dim[HMP_MAX_NODE_TYPES]; /* output width per node type (0 = type receives nothing) */
  /* layer 0 only: node types the layer does not produce but whose INPUT features stay visible to layer 1
   * (`x_dict.update(pre_mp(...))`, heterogeneous_neural_tree_network.py:158-159); out_dim must equal in_dim */
  int32_t passthrough[HMP_MAX_NODE_TYPES];
  hmp_conv_spec convs[HMP_MAX_CONVS];
} hmp_layer_spec;

typedef struct hmp_net_spec {
  int32_t n_node_types, n_edge_types, n_layers;
  int32_t in_dim[HMP_MAX_NODE_TYPES];
  int32_t edge_src[HMP_MAX_EDGE_TYPES], edge_dst[HMP_MAX_EDGE_TYPES];
  int32_t readout_type;    /* node type whose final state is the output */
  int32_t pool_edge_type;  /* -1, or LeafPool edge type: output = segment mean over it, first n_out rows */
  int32_t aux_readout_type; /* -1, or a SECOND node type whose final state is an output too: the two-headed task
                             * (`return x_dict["rooms"], x_dict["objects"]`, heterogeneous_network.py:123-135); read with
                             * hmp_net_aux_output, its gradient enters through hmp_net_backward2 */
  int64_t n_params;        /* total floats in the flat parameter buffer */
  int64_t n_active_params; /* parameters [0, n_active) receive gradients; the tail is dead weights */
  hmp_layer_spec layers[HMP_MAX_LAYERS];
} hmp_net_spec;

typedef struct hmp_batch {
  int32_t n_nodes[HMP_MAX_NODE_TYPES];
  const float* d_x[HMP_MAX_NODE_TYPES];   /* [n_nodes, in_dim] */
  int32_t ldx[HMP_MAX_NODE_TYPES];
  int64_t n_edges[HMP_MAX_EDGE_TYPES];
  const int64_t* d_edge_index[HMP_MAX_EDGE_TYPES]; /* [2][E] int64 */
  const float* d_edge_attr[HMP_MAX_EDGE_TYPES];    /* [E, edge_dim] or NULL */
  int32_t n_out;             /* rows of the output (== n_nodes[readout] unless pooled) */
  const int64_t* d_labels;   /* [n_out] int64, or NULL (forward only) */
  int32_t plan_valid;        /* != 0: the caller vouches that every edge list equals the previous call's (same topology, e.g.
                              * consecutive frames of the inference server, bin/room_classification_server:273-299): the CSR /
                              * CSC plan in the workspace is reused instead of rebuilt.  Counts must match the previous call. */
  /* optional: the batch as a disjoint union of graphs ([PyG] Batch.ptr per node type; base_training_job.py:164-168 collates
   * that way).  With it the small-batch training step runs everything between the first aggregation and the weight gradients
   * as ONE launch, one workgroup per graph.  n_graphs = 0: unknown (any edge structure is accepted, multi-launch sequence). */
  const int64_t* d_node_ptr[HMP_MAX_NODE_TYPES]; /* [n_graphs + 1] int64 row offsets of the node type, or NULL */
  int32_t n_graphs;
  int32_t max_graph_nodes;   /* largest per-graph node count over all types (host knowledge of the collation), 0 = unknown */
  /* optional, with d_node_ptr of both endpoint types: [n_graphs + 1] int64 edge offsets of the edge type -- the caller vouches
   * that the edges of graph g are entries [ptr[g], ptr[g+1]) of the edge list and join nodes of graph g only (what every
   * collation of a list of graphs produces: [PyG] Batch.from_data_list, base_training_job.py:164-168).  The single-launch plan
   * build then reads, per block of rows, only the edges of the graphs that own those rows instead of the whole list.  An edge
   * found outside its graph's rows sets status bit 4.  NULL: no assumption about the edge order. */
  const int64_t* d_edge_ptr[HMP_MAX_EDGE_TYPES];
} hmp_batch;

typedef struct hmp_train_args {
  float lr, beta1, beta2, eps, weight_decay;
  int64_t ignored_label;
  uint64_t seed;
  int32_t training;          /* dropout on */
  int32_t* d_step;           /* device int32 owned by the optimiser state (next to Adam's m / v): the step counter t that
                              * Adam's bias correction and the dropout stream read; bumped once at the head of every
                              * step.  NULL: the net's own counter (one optimiser per net).  torch.optim.Adam keeps
                              * `state['step']` per optimiser in the same way (base_training_job.py:181-185) */
} hmp_train_args;

typedef struct hmp_net hmp_net; /* opaque */

int hmp_net_create(const hmp_net_spec* spec, hmp_net** out);
void hmp_net_destroy(hmp_net* net);
/* device bytes the executor needs for batches up to these capacities */
size_t hmp_net_workspace_bytes(const hmp_net* net, const int32_t* cap_nodes, const int64_t* cap_edges);
int hmp_net_bind_workspace(hmp_net* net, void* d_workspace, size_t bytes, const int32_t* cap_nodes, const int64_t* cap_edges);

/* forward: builds the plan for `batch`, runs every layer; *d_out -> [n_out, out_dim] inside the
 * workspace (ld = *ld_out).  rng_step selects the dropout stream of this call. */
int hmp_net_forward(hmp_net* net, const hmp_batch* batch, const float* d_params, int32_t training, uint64_t seed,
                    uint32_t rng_step, const float** d_out, int32_t* ld_out, void* stream);
/* backward of the last forward: d_gout [n_out, ld_gout] -> flat gradient d_grads[0 : n_active_params)
 * (overwritten, not accumulated).  d_gx[t] (may be NULL) receives d loss / d x[t], ld = ldx[t]. */
int hmp_net_backward(hmp_net* net, const float* d_gout, int32_t ld_gout, const float* d_params, float* d_grads,
                     float* const* d_gx, void* stream);
/* two-headed nets (spec.aux_readout_type >= 0): the final state of the second node type after the last forward
 * ([*n_rows, *ld_out] inside the workspace), and the backward that takes a gradient for both outputs (either may be NULL = zero;
 * d_gout_aux is [n_nodes[aux], ld_aux], any ld >= the type's output width). */
int hmp_net_aux_output(hmp_net* net, const float** d_out, int32_t* ld_out, int32_t* n_rows);
int hmp_net_backward2(hmp_net* net, const float* d_gout, int32_t ld_gout, const float* d_gout_aux, int32_t ld_aux,
                      const float* d_params, float* d_grads, float* const* d_gx, void* stream);

/* fused training step, phase A: plan + forward + masked CE + backward.  Leaves the SUM-loss gradient in
 * d_grads[0 : n_active) and {loss_sum, count} in d_grads[n_active], d_grads[n_active+1] so that ONE
 * all-reduce(sum) over n_active+2 floats averages count-weighted across ranks.  Phase B: Adam on
 * [0, n_active) with grad scale 1/count read from that tail.  The step counter lives on the device so
 * both phases can be replayed from a captured hipGraph. */
int hmp_net_step_fwd_bwd(hmp_net* net, const hmp_batch* batch, const float* d_params, float* d_grads,
                         const hmp_train_args* args, void* stream);
int hmp_net_step_adam(hmp_net* net, float* d_params, const float* d_grads, float* d_m, float* d_v,
                      const hmp_train_args* args, void* stream);
/* single-rank step: phase A + phase B in one call (no collective in between).  Lets the executor fold Adam into the
 * gradient un-pack kernel where the network allows it (SAGE stacks); the result equals A followed by B. */
int hmp_net_step_fused(hmp_net* net, const hmp_batch* batch, float* d_params, float* d_grads, float* d_m, float* d_v,
                       const hmp_train_args* args, void* stream);
/* diagnosis / tests: where the last forward left the output of layer `layer` (1 .. n_layers) for `node_type`: rows [n_rows, width]
 * at pitch *ld elements, fp32 or (*is_bf16) bfloat16.  A dropped element (training-mode dropout) is stored as -0: its sign bit
 * is the keep-mask the backward reads.  Valid until the next forward / step / workspace re-bind. */
int hmp_net_hidden(hmp_net* net, int32_t layer, int32_t node_type, const void** d_h, int32_t* ld, int32_t* n_rows,
                   int32_t* width, int32_t* is_bf16);
/* compute mode of the dense projections: 0 (default) exact fp32 MFMA everywhere; 1 = GEMM calls in the throughput-bound regime
 * (>= 1024 64x64 output tiles: BASELINE config 5, "hidden=256 bf16") round their operands to bf16 and run on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation; in that regime the intermediates that are only ever gathered or fed to those
 * GEMMs (projected rows Z, their gradient dZ, hidden activations H, input gradients G of 256-wide layers) are also STORED as bf16
 * -- features, logits, parameters, gradients and optimiser state stay fp32.  Not within the 1e-5 parity bar: an explicit
 * precision choice of the caller, checked against oracle/bf16_emul.py (tests/test_gpu_config5.py). */
int hmp_net_set_compute(hmp_net* net, int32_t bf16);
/* host copy of {step counter, status bits}; synchronises the stream */
int hmp_net_read_state(hmp_net* net, int32_t* step, int32_t* status, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 7. hipGraph capture of a launch sequence (the step is ~20 short kernels: launch-bound).
 * ------------------------------------------------------------------------------------------- */
typedef struct hmp_graph hmp_graph;
int hmp_graph_begin(void* stream);                   /* stream must not be the null stream */
int hmp_graph_end(void* stream, hmp_graph** out);
int hmp_graph_launch(hmp_graph* g, void* stream);
void hmp_graph_destroy(hmp_graph* g);

/* HIP event timing on the caller's stream (bench.py roofline leg) */
typedef struct hmp_timer hmp_timer;
int hmp_timer_create(hmp_timer** out);
int hmp_timer_start(hmp_timer* t, void* stream);
int hmp_timer_stop(hmp_timer* t, void* stream);
int hmp_timer_elapsed_ms(hmp_timer* t, float* ms); /* synchronises on the stop event */
void hmp_timer_destroy(hmp_timer* t);
/* per-kernel-class device time accumulated by the executor when profiling is on (HIP events around
 * every launch of that class on the executor's stream).  classes: 0 plan, 1 pack, 2 gemm_fwd,
 * 3 aggregate_fwd, 4 loss, 5 aggregate_bwd, 6 gemm_bwd, 7 grad_reduce, 8 adam, 9 gat_fwd, 10 gat_bwd, 11 pool,
 * 12 front (layer-0 projection + plan + pack in one launch, small batches), 13 chain (graph-local launch: every aggregation
 * phase of the step, one workgroup per graph) */
#define HMP_N_KCLASS 14
int hmp_net_profile(hmp_net* net, int32_t enable);
int hmp_net_profile_read(hmp_net* net, float* ms_sum /*[HMP_N_KCLASS]*/, int32_t* launches /*[HMP_N_KCLASS]*/);

/* ---------------------------------------------------------------------------------------------
 * 8. Device-side collation: replaces [PyG] DataLoader -> Batch.from_data_list + the per-step H2D copy
 *    (base_training_job.py:164-178, :205).  The dataset lives in HBM as packed arrays; a batch = the graphs sel[0..B).
 *    rows:  d_dst[d_dst_off[b] + i] = d_src[d_src_ptr[sel[b]] + i]            (row_bytes per row, multiple of 4)
 *    edges: d_dst[r][d_dst_off[b] + j] = d_src[r][d_edge_ptr[sel[b]] + j] + (r ? d_off_dst : d_off_src)[b]
 *           (d_src is [2][e_total] int64 with graph-local indices, d_dst is [2][e_out])
 * ------------------------------------------------------------------------------------------- */
int hmp_collate_rows(const void* d_src, int64_t row_bytes, const int64_t* d_src_ptr, const int32_t* d_sel,
                     const int64_t* d_dst_off, int32_t B, int64_t n_out_rows, void* d_dst, void* stream);
int hmp_collate_edges(const int64_t* d_src, int64_t e_total, const int64_t* d_edge_ptr, const int32_t* d_sel,
                      const int64_t* d_dst_off, const int64_t* d_off_src, const int64_t* d_off_dst, int32_t B,
                      int64_t e_out, int64_t* d_dst, void* stream);

/* One-launch form: the dataset description is fixed at creation, a batch costs ONE host call and ONE kernel.
 *   slots   the distinct [G + 1] per-graph offset vectors (one per node type and per edge type; host copies are kept);
 *   items   the output arrays: rows (row_bytes > 0: x / y / pos / edge_attr, positioned by `slot`) or an edge_index
 *           (row_bytes == 0: int64 [2][src_total], positioned by `slot`, endpoints shifted by `slot_src` / `slot_dst`).
 * hmp_collator_run: h_sel [B] graph ids (host), d_dst[i] / dst_capacity[i] per item (rows resp. edges), h_totals[slot] receives
 * the batch's node / edge totals (the shapes of the outputs).  No allocation or synchronisation in the steady state.
 * (<= 356 words of offsets + selection travel in the kernel's argument block, larger batches through a pinned ring.) */
typedef struct hmp_collate_item {
  const void* d_src;
  const int64_t* d_ptr;      /* device copy of the item's [G + 1] offsets */
  int64_t src_total;         /* edges: E_total of the packed edge_index */
  int64_t row_bytes;         /* rows: bytes per row (multiple of 4); edges: 0 */
  int32_t slot, slot_src, slot_dst;
} hmp_collate_item;
typedef struct hmp_collator hmp_collator; /* opaque */
int hmp_collator_create(int32_t n_slots, const int64_t* const* h_slot_ptr, int64_t n_graphs, int32_t n_items,
                        const hmp_collate_item* items, hmp_collator** out);
int hmp_collator_run(hmp_collator* c, const int32_t* h_sel, int32_t B, void* const* d_dst, const int64_t* dst_capacity,
                     int64_t* h_totals, int64_t* d_offsets_out /* NULL or [n_slots][offsets_stride]: Batch.ptr of every slot */,
                     int32_t offsets_stride, void* stream);
void hmp_collator_destroy(hmp_collator* c);

/* ---------------------------------------------------------------------------------------------
 * 9. Homogeneous GCN / GIN operators (csrc/homog.hip).  Replace [PyG] GCNConv.propagate + gcn_norm
 *    (models/utils.py:15-16), GINConv.propagate (models/utils.py:17-26) and BatchNorm (models/homogeneous_network.py:93-97).
 *    All take a square plan (n_src == n_dst); projections go through hmp_gemm_f32 BEFORE the neighbourhood sum.
 *    gcn_norm:     dinv[i] = (1 + #{j->i, j != i})^-1/2         (add_remaining_self_loops: one loop of weight 1 per node)
 *    segment_wsum: d_w != NULL (GCN): out_i = w_i ( sum_{j->i, j != i} w_j x_j + w_i x_i )
 *                  d_w == NULL (GIN): out_i = sum_{j->i} x_j + (1 + *d_eps) x_i      (d_eps NULL = 0)
 *                  transpose != 0 runs the same sum over the CSC lists (= the gradient w.r.t. x).
 *    bias_act_drop: y = dropout_p(act(x + bias)), act = HMP_ACT_NONE / RELU / ELU; keep-mask = hmp_dropout_mask(seed, rng_step,
 *                  rng_stream, p, n, F); a dropped element is stored as -0.0f, so the backward needs y only; p > 0 requires an
 *                  activation.
 *    colsum / rowdot_sum: out[c] = sum_r g[r,c];  *out = sum_{r,c} a[r,c] b[r,c]   (fixed summation order)
 *    batchnorm:    torch.nn.BatchNorm1d semantics (training: batch statistics, running stats updated in place with the
 *                  unbiased variance; eval: running statistics).  d_save [2][F] = {mean, 1/sqrt(var+eps)} for the backward.
 * ------------------------------------------------------------------------------------------- */
int hmp_gcn_norm(hmp_plan plan, float* d_dinv, void* stream);
int hmp_segment_wsum(const float* d_x, int32_t ldx, int32_t F, hmp_plan plan, int32_t transpose, const float* d_w,
                     const float* d_eps, float* d_out, int32_t ldo, void* stream);
int hmp_bias_act_drop_fwd(const float* d_x, int32_t ldx, int32_t n_rows, int32_t F, const float* d_bias, int32_t act,
                          float p, uint64_t seed, uint32_t rng_step, uint32_t rng_stream, float* d_y, int32_t ldy,
                          void* stream);
int hmp_bias_act_drop_bwd(const float* d_g, int32_t ldg, const float* d_y, int32_t ldy, int32_t n_rows, int32_t F,
                          int32_t act, float p, float* d_gx, int32_t ldgx, void* stream);
int hmp_colsum(const float* d_g, int32_t ldg, int32_t n_rows, int32_t F, float* d_out, void* stream);
int hmp_rowdot_sum(const float* d_a, int32_t lda, const float* d_b, int32_t ldb, int32_t n_rows, int32_t F, float* d_out,
                   void* stream);
int hmp_batchnorm_fwd(const float* d_x, int32_t ldx, int32_t n_rows, int32_t F, const float* d_gamma, const float* d_beta,
                      float* d_running_mean, float* d_running_var, float momentum, float eps, int32_t training, float* d_y,
                      int32_t ldy, float* d_save, void* stream);
int hmp_batchnorm_bwd(const float* d_g, int32_t ldg, const float* d_x, int32_t ldx, int32_t n_rows, int32_t F,
                      const float* d_gamma, const float* d_save, int32_t training, float* d_gx, int32_t ldgx, float* d_ggamma,
                      float* d_gbeta, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 10. Object connectivity of a room-object scene graph (csrc/dsg.hip).  Replaces the per-frame Python loop
 *     add_object_connectivity (src/hydra_gnn/preprocess_dsgs.py:191-225) with its predicates _is_on / _is_under / _is_near
 *     (:89-180), float64 like numpy.  Objects in the reference's visiting order (ascending node id); d_room[i] = room index or
 *     < 0.  count: d_count[i] = #{j < i in i's room : on | under | near}, d_offset = its exclusive scan (d_offset[n] = total).
 *     fill: d_edges [2][total] = (i, j) pairs ordered by i then j -- the reference's insertion order.
 * ------------------------------------------------------------------------------------------- */
int hmp_object_edges_count(const double* d_pos, const double* d_size, const int32_t* d_room, int32_t n, double threshold_near,
                           double max_near, double max_on, int32_t* d_count, int32_t* d_offset, void* stream);
int hmp_object_edges_fill(const double* d_pos, const double* d_size, const int32_t* d_room, int32_t n, double threshold_near,
                          double max_near, double max_on, const int32_t* d_offset, int32_t* d_edges, int32_t total, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 11. Data-parallel gradient exchange (csrc/comm.hip).  The reference has no multi-GPU code; a batch is a
 *     disjoint union of scene graphs (base_training_job.py:164-168), so ranks exchange nothing but ONE all-reduce(sum) per
 *     step over the flat [gradient sums | loss_sum | count] buffer.  These entry points put that collective on the stream
 *     the step's kernels run on (between hmp_net_step_fwd_bwd and hmp_net_step_adam): RCCL over xGMI, one process per GPU.
 *     Rendezvous: rank 0 calls hmp_comm_unique_id and hands the 128 bytes to the other ranks by any channel the host has
 *     (torch.distributed broadcast, a file, MPI); every rank then calls hmp_comm_create with the device it owns current.
 *     RCCL is resolved at the first call (dlopen): HMP_E_UNSUPPORTED when the machine has none.
 * ------------------------------------------------------------------------------------------- */
#define HMP_COMM_ID_BYTES 128
typedef struct hmp_comm hmp_comm; /* opaque */
int hmp_comm_unique_id(void* id128);
int hmp_comm_create(const void* id128, int32_t rank, int32_t world, hmp_comm** out);
void hmp_comm_destroy(hmp_comm* comm);
/* what RCCL itself says about the communicator (ncclCommCount / ncclCommUserRank): the rank count a benchmark line may quote */
int hmp_comm_query(hmp_comm* comm, int32_t* n_ranks, int32_t* rank);
int hmp_comm_allreduce_sum_f32(hmp_comm* comm, float* d_buf, int64_t n, void* stream);
int hmp_comm_broadcast_f32(hmp_comm* comm, float* d_buf, int64_t n, int32_t root, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 12. H-tree (Neural-Tree) construction on the host (csrc/htree.cpp).  Replaces generate_htree +
 *     add_virtual_nodes_to_htree + the typed extraction of nx_htree_to_torch (src/hydra_gnn/neural_tree/construct.py:241-371,
 *     450-468) and generate_jth / networkx.junction_tree underneath (generate_junction_tree_hierarchies.py:26-116).
 *     Input: a room-object scene graph -- object-object, room-room and room->object edge lists [2][E] int64 (undirected: either
 *     or both directions may be listed).  Output (hmp_htree_sizes, then hmp_htree_fill into caller arrays, all int32):
 *       counts[4]       nodes per type: object, room, object-room, room-room (leaves are COPIES of scene-graph nodes)
 *       object_orig / room_orig   original index (inside its type) of every object / room leaf  == the pool edges o_to_ov / r_to_rv
 *       edges[10]       [2][n] local indices for HTREE_EDGE_TYPES in the reference's order (construct.py:15-26), both directions
 *       init[3]         [2][n] ov_to_or, rv_to_or, rv_to_rr: (original index of the member, clique)
 *     Ties that networkx leaves to Python set order go to the smallest index (see the file header).  Host memory only.
 * ------------------------------------------------------------------------------------------- */
typedef struct hmp_htree hmp_htree; /* opaque */
int hmp_htree_build(int32_t n_objects, int32_t n_rooms, const int64_t* oo_edges, int64_t n_oo, const int64_t* rr_edges,
                    int64_t n_rr, const int64_t* ro_edges, int64_t n_ro, hmp_htree** out);
int hmp_htree_sizes(const hmp_htree* t, int32_t* counts4, int64_t* n_edges10, int64_t* n_init3);
int hmp_htree_fill(const hmp_htree* t, int32_t* object_orig, int32_t* room_orig, int32_t* const* edges10, int32_t* const* init3);
void hmp_htree_destroy(hmp_htree* t);

#ifdef __cplusplus
}
#endif
#endif /* HYDRA_MP_H */
