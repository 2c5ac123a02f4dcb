// Launch-level interfaces between the translation units (device pointers everywhere).
#pragma once
#include "common.h"

namespace hmp {

// ---------------------------------------------------------------------------------------------
// K2: grouped fp32 MFMA GEMM.  C[M,N] = op(A)[M,K] * op(B)[K,N]
// ---------------------------------------------------------------------------------------------
enum { EPI_NONE = 0, EPI_ACTMASK = 1 };

struct GemmProblem {
  const float* A;
  const float* B;
  float* C;
  const float* H;        // EPI_ACTMASK: activations whose derivative masks C (same shape as C)
  int c_bf16;            // gemm_bf16_kernel: C holds bf16 elements (ldc counts elements)
  int a_bf16;            // gemm_bf16_kernel: A holds bf16 elements (lda counts elements); every tile must take the fast loader
  int b_bf16;            // gemm_bf16_kernel: B holds bf16 elements (activations H as the weight gradient's operand); n_real must be
                         // whole tiles (a multiple of 256), the virtual ones column is the kernel's ONES product
  int h_bf16;            // gemm_bf16_kernel, EPI_ACTMASK: H holds bf16 elements (ldh counts elements)
  const float* A2;       // gemm_bf16_kernel with a_bf16: the slice of A from index a_split on (along its row-contiguous / K
  int lda2, a_split;     // axis: columns of dZ) lives in a second bf16 matrix -- the root block of dZ IS the output gradient,
                         // so nobody copies it.  a_split > 0 and a multiple of 256; 0: A is one matrix
  int64_t slab_stride;   // split-K: slab z is written at C + z*slab_stride
  int M, N, K;
  int lda, ldb, ldc, ldh;
  int trans_a;           // 0: A[m*lda+k]   1: A[k*lda+m]
  int trans_b;           // 0: B[k*ldb+n]   1: B[n*ldb+k]
  int n_real;            // columns of B that exist in memory (N or N-1 when aug_ones)
  int aug_ones;          // B gets a virtual last column of ones (bias gradient = column sums of A)
  int ksplit, kchunk;    // filled by gemm_launch
  int tiles_m, tiles_n, tile_start;
  int epi, act;
  int drop_on;
  DropCfg drop;          // dropout site of H (quad index = row*(ldh/4) + col/4)
  // NN form only: gather-add before the mask -- C[m, :] += sum over the entries k in [g_rowptr[m], g_rowptr[m+1]) of
  // g_rows[g_col[k], 0:N] * g_deg[g_col[k]]: the gradient of an AGGREGATE-FIRST conv's segment mean (net.hip), whose source
  // rows are this problem's output rows; g_rows is fp32 [.][g_ld].  g_rowptr == null: none
  // (gemm_bf16_dx_kernel only; the tiled kernels take the same term as a materialised matrix:)
  const int* g_rowptr;
  const int* g_col;
  const float* g_deg;
  const float* g_rows;
  int g_ld;
  const float* Cadd;     // NN form: fp32 [M][ldadd] added to the product before the mask (null: none)
  int ldadd;
};

// sum over the CSC entries of `row` of g_rows[g_col[k], col] * g_deg[g_col[k]] (row < 0: nothing) -- the tiled kernels' form of
// GemmProblem::g_rowptr (one element at a time: the regime that uses it in earnest runs gemm_bf16_dx_kernel)
constexpr int GEMM_MAX_PROB = 8;
constexpr int GEMM_TALL_SLABS = 192;  // split-K slabs the tall weight-gradient kernel may ask for (== net.hip: MAX_SLABS, the slab buffer's size)
struct GemmBatch {
  int n;
  int total_tiles;
  GemmProblem p[GEMM_MAX_PROB];
};

// chooses tile shape + split-K (only when want_split: C is then a stack of slabs) and launches
int gemm_launch(GemmBatch& gb, bool want_split, int max_slabs, hipStream_t st);
// same contract on the bf16 matrix pipe (operands rounded to bf16 on their way into LDS, fp32 accumulate): gemm_bf16.hip
int gemm_bf16_launch(GemmBatch& gb, bool want_split, int max_slabs, hipStream_t st);
// fp32 accuracy on the bf16 matrix pipe (three exact bf16 pieces per operand element, six piece products): gemm_x3.hip; takes =
// every problem of the launch is plain fp32, the launch is big enough, and split-K launches only over < 32 768 nodes (HMP_GEMM_X3=0: never; =2: every split-K launch)
bool gemm_x3_takes(const GemmBatch& gb, bool want_split);
bool gemm_x3_split_takes(const GemmProblem* ps, int n);  // (split-K weight gradients of a step, before the direct TN kernel is chosen)
int gemm_x3_launch(GemmBatch& gb, bool want_split, int max_slabs, hipStream_t st);
// operand-stationary backward kernels of the 10^6-node regime (gemm_bf16_bwd.hip); *_takes: the problem fits the kernel's limits
bool gemm_bf16_dx_takes(const GemmProblem& p, bool want_split);
int gemm_bf16_dx_launch(const GemmProblem& p, hipStream_t st);
bool gemm_bf16_dw_takes(const GemmProblem& p, bool want_split);
int gemm_bf16_dw_launch(const GemmProblem& p, int max_slabs, int* n_slabs, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// front kernel of the small-batch step (front.hip): layer-0 projection tiles + plan parts + pack blocks in one launch
// ---------------------------------------------------------------------------------------------
constexpr int FR_MAX_PROB = 4;
constexpr int FR_MAX_SEG = 6;
constexpr int FR_MAX_SRC = 6;  // == AGG_MAX_IN
struct FrontSeg {    // rows [col0, col0 + rows) of the stacked B operand = sum of nsrc parameter blocks [rows][ld]
  int col0, rows, nsrc, ld;
  uint32_t off[FR_MAX_SRC];  // float offsets into the flat parameter buffer
};
struct FrontProb {   // C[M, N] = A[M, K] * B^T
  const float* A;
  float* C;
  int lda, ldc, M, N, K;
  int n_seg, max_nsrc, vec;
  int blk_start, tiles_m, tiles_n;
  FrontSeg seg[FR_MAX_SEG];
};
struct FrontJob {    // one edge type of the plan; arrays as 4-byte-word offsets from the workspace base
  const int64_t* ei;
  int E, n_src, n_dst;
  uint32_t rowptr, col, eid, t_rowptr, t_col, t_eid, tmp_in, tmp_out, pos_of_eid, degf;
  uint32_t ell, t_ell;  // 0: none
  const int64_t *gp_dst, *gp_src, *gp_edge;  // as PlanJob
  int n_graphs;
};
struct PackSeg;
struct NetState;
struct FrontArgs {
  // roles by block index: [0, gemm_blocks) projection tiles, then plan parts, then pack blocks
  int gemm_blocks, plan_blocks, pack_blocks;
  int n_prob, n_jobs, need_tpos, plan_rc;
  const float* params;
  char* ws;
  int* status;
  const PackSeg* segs;
  const int2* pack_map;  // per pack block: {segment, first item}; 16 items (packed row, 64-column chunk) per block
  float* packed;
  int* step_ctr;         // non-null: the pack role's first block bumps this device step counter
  int* step_mirror;      // ... and leaves the new value here (NetState::last_step: the net never keeps the caller's pointer)
  int prob_start[FR_MAX_PROB];  // prob[i].blk_start again, next to the header fields (front_launch fills it): see karg_warm (common.h)
  int part_start[2 * HMP_MAX_EDGE_TYPES + 1];
  int rows_per_part[2 * HMP_MAX_EDGE_TYPES];
  FrontJob job[HMP_MAX_EDGE_TYPES];
  FrontProb prob[FR_MAX_PROB];
};
int front_launch(FrontArgs& a, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// register-direct TN GEMM for small batches (gemm_direct.hip)
// ---------------------------------------------------------------------------------------------
struct TnProblem {   // slab z of C[M, N] = sum_{k in chunk z} A[k, m] * [B | 1][k, n]
  const float* A;
  const float* B;
  float* C;
  int64_t slab_stride;
  int M, N, K, lda, ldb, ldc;
  int n_real, aug_ones;
  int ksplit, kchunk, tile_start, tiles_m, tiles_n;
};
struct TnBatch {
  int n, total_tiles;
  TnProblem p[GEMM_MAX_PROB];
};
// *ok = 0 (nothing launched) when a problem would need more than max_slabs slabs
int gemm_tn_direct_launch(TnBatch& tb, int max_slabs, int* ok, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------------
struct PlanJob {
  const int64_t* ei;  // [2][E]
  int64_t E;
  int n_src, n_dst;
  int *rowptr, *col, *eid, *t_rowptr, *t_col, *t_pos;
  float* degf;  // optional [n_dst]: 1 / max(in-degree, 1) as float
  int *ell, *t_ell;  // optional [n_dst][ELL_W] / [n_src][ELL_W]: first ids of every row; written by the single-launch build only
  // optional (all three or none; single-launch build only): the batch is a union of n_graphs graphs whose edges are listed graph
  // by graph -- int64 [n_graphs + 1] row offsets of the destination / source node type and edge offsets (hmp_batch::d_edge_ptr)
  const int64_t *gp_dst, *gp_src, *gp_edge;
  int n_graphs;
  // scratch
  int *cnt_in, *cnt_out, *cur_in, *cur_out;  // must be zero on entry (one contiguous block); left zero on exit
  int* flags;  // multi-launch build: [0] / [1] = PlanBatch::build_id of the last build that found the list NOT ordered by destination /
               // source (or holding an invalid edge); part of the zeroed block.  An edge list that arrives ordered by destination --
               // what graph builders emit -- IS its CSR (position = edge id): no scatter, no rank pass for that direction
  int *tmp_in, *tmp_out, *t_eid, *pos_of_eid;
  int *tmpc_in, *tmpc_out;  // multi-launch build: the other endpoint of every scratch slot
};
struct PlanBatch {
  int n;
  int need_tpos;   // build t_pos (only GAT's source-major backward reads it)
  int clear_first; // memset the counters before the launch (caller-provided scratch of unknown content)
  int built_small; // out: plan_launch took the single-launch build (the one that also writes PlanJob::ell / t_ell)
  int build_id;    // multi-launch build: stamp of this build (see PlanJob::flags)
  int* flags_all;  // [2 * n] = job 0's flag block used for ALL jobs: entry 2 j + dir (one pointer in the batch header: reading a
                   // per-job pointer through a per-lane job index made the compiler copy the 4 KB argument block to scratch)
  int64_t edge_start[HMP_MAX_EDGE_TYPES + 1];
  int64_t row_start[2 * HMP_MAX_EDGE_TYPES + 1];  // rows of (job, dir): dir 0 = by dst, 1 = by src
  PlanJob j[HMP_MAX_EDGE_TYPES];
};
size_t plan_scratch_ints(int64_t E, int n_src, int n_dst);
// carve scratch for job (all int32); zero_begin/zero_ints return the region to memset
void plan_carve(PlanJob& job, int* scratch);
int plan_launch(PlanBatch& pb, int* d_status, hipStream_t st);
int plan_link_launch(PlanBatch& pb, hipStream_t st);  // t_pos only, after a single-launch build that ran elsewhere (front kernel)

// ---------------------------------------------------------------------------------------------
// K1 + fused SAGE aggregation
// ---------------------------------------------------------------------------------------------
constexpr int AGG_MAX_IN = 6;
// Neighbour-id table in ELL form next to the CSR arrays: row r's first ELL_W ids at ell[r * ELL_W ..] (slots past the row's degree
// are NOT written).  Its address depends on the row alone, so the small-batch aggregation kernels request it together with the
// row extent -- extents+ids, rows: two dependent round trips where the CSR form needs three (extents, ids, rows).
constexpr int ELL_W = 16;
struct AggIn {
  const int* rowptr;
  const int* col;
  const float* z;  // projected source rows
  int ldz, coff;
  int same_type;   // source node type == destination node type (one index space): candidate for the LDS-windowed gather
  int n_src;       // rows of z
  const int* ell;  // [n_dst][ELL_W] first neighbour ids of every row (single-launch plan by-product, see plan_small.h), or null
};
struct AggDst {
  int n_rows, F;
  int type;  // node type index (graph-local chain: which row-offset vector positions this entry)
  float* out;
  int ldo;
  const float* zroot;  // may be null
  int ldzr, roff;
  const float* bias;   // may be null
  int n_in;
  int act;
  int drop_on;
  DropCfg drop;
  int block_start;
  // fused projection of the NEXT layer (agg_proj_fwd_launch): Z[l+1][t] = out * pw^T, pw = Wp[l+1][t] [pncols][pldw]
  const float* pw;  // null: this node type is not read by the next layer
  float* pz;
  int pldw, pldz, pncols, pK;
  // fused masked cross entropy on the rows of this entry (last layer of the fused training step): gradient of the SUM loss
  // to ce_grad, per-row {loss, valid} to ce_row_lv
  const int64_t* ce_labels;  // null: off
  int64_t ce_ignored;
  int ce_classes, ce_ldg;
  float* ce_grad;
  float* ce_row_lv;
  int win_in, win_src_rows;  // filled by agg_fwd_launch: in-conv served from the LDS window (-1: none), rows of its source
  int tile_rows;             // agg_proj_fwd_launch: rows per workgroup, 16 or 8.  A launch is as long as its slowest tile and a tile of
                             // high-degree rows (rooms: ~12 objects each) requests twice the lines of the others (tools/ktime_blocks.py:
                             // 7.3 us against 5.8 us): such entries are cut into twice as many tiles of 8 rows (there are idle CUs)
  AggIn in[AGG_MAX_IN];
};
struct AggArgs {
  int n;
  int total_blocks;
  int mean;  // divide every gather by max(deg,1)
  int xcd;   // large launches: consecutive row ranges stay on one XCD (block counts padded to 8 per entry), see agg_fwd_launch
  int zb16;  // the projected rows (AggIn::z, AggDst::zroot) hold bf16 elements (bf16 compute mode, 256-wide rows)
  int hb16;  // with zb16: the outputs (AggDst::out) are WRITTEN as bf16 elements too (activations of a hidden layer, read only by GEMMs)
  int win_R, win_W;  // LDS-windowed launch (agg_fwd_win_kernel): destination rows per workgroup / source rows staged
  NetState* state;  // status bits (fused cross entropy: label out of range)
  int bstart[HMP_MAX_NODE_TYPES];  // d[i].block_start again, in the struct's first lines (filled by the launchers): a workgroup
                                   // finds its entry without touching one argument line per entry, see karg_warm (common.h)
  AggDst d[HMP_MAX_NODE_TYPES];
};
int agg_fwd_launch(AggArgs& a, hipStream_t st);
// same + next-layer projection per 16-row tile (requires F <= 256, pK % 16 == 0, pK <= 256 for every entry with pw)
int agg_proj_fwd_launch(AggArgs& a, hipStream_t st);

struct TAggOut {
  const int* t_rowptr;
  const int* t_col;
  const int* rowptr;  // forward CSR rowptr of the same edge type (for 1/deg of the destination)
  const float* degf;  // 1 / max(deg,1) per destination (plan by-product), null: derive from rowptr
  const float* g;     // gradient rows of the destination type
  int ldg, coff, F;
  int same_type;      // destination node type == source node type: candidate for the LDS-windowed gather
  int n_dst;          // rows of g
  const int* t_ell;   // [n_src][ELL_W] first out-neighbour ids of every row, or null (see AggIn::ell)
};
struct TAggSrc {
  int n_rows;
  int type;  // node type index
  float* dz;
  int lddz, ncols;     // ncols (padded): columns not covered by a segment are zeroed
  const float* groot;  // may be null
  int ldgr, roff, Froot;
  int n_out;
  int block_start;
  // fused input gradient (agg_bwd_dx_launch): xg = (dz * xw) . act'(xh), xw = Wp[l][s] [ncols][xldw]
  const float* xw;  // null: no input gradient for this node type
  float* xg;
  const float* xh;  // activations whose derivative masks xg (null: none)
  int xldw, xN, xldg, xldh, xact, xdrop_on;
  float xscale;
  int win_out, win_dst_rows;  // filled by agg_bwd_launch (see AggDst::win_in)
  int tile_rows;              // agg_bwd_dx_launch: rows per workgroup, 16 or 8 (see AggDst::tile_rows)
  TAggOut out[AGG_MAX_IN];
};
struct TAggArgs {
  int n;
  int total_blocks;
  int mean;
  int xcd;  // as AggArgs::xcd
  int gb16; // the gradient rows (TAggOut::g, TAggSrc::groot) hold bf16 elements
  int dzb16; // dz is written as bf16 (requires gb16)
  int win_R, win_W;  // LDS-windowed launch (agg_bwd_win_kernel)
  // one extra block sums the per-row {loss, valid} pairs of the loss in fixed order -> fin_out2 / fin_state (null: off)
  const float* fin_row_lv;
  int fin_rows;
  float* fin_out2;
  NetState* fin_state;
  int bstart[HMP_MAX_NODE_TYPES];  // s[i].block_start again (as AggArgs::bstart)
  TAggSrc s[HMP_MAX_NODE_TYPES];
};
int agg_bwd_launch(TAggArgs& a, hipStream_t st);
// same + the row-local input-gradient GEMM per 16-row tile (requires every segment width <= 256, ncols <= 896)
int agg_bwd_dx_launch(TAggArgs& a, hipStream_t st);

// aggregate-first convs (aggfirst.hip): segment mean of source rows (fp32 or bf16) into fp32 rows, and its transpose with the
// activation / dropout mask of the source rows for source types that have no input-gradient GEMM in the layer
int seg_mean_rows_launch(const void* x, int ldx, int x_bf16, int F, const int* rowptr, const int* col, int n_rows, float* m, int ldm,
                         hipStream_t st);
int seg_mean_rows_t_launch(const float* dm, int lddm, int F, const int* t_rowptr, const int* t_col, const float* degf, int n_rows,
                           const void* h, int ldh, int h_bf16, int act, int drop_on, float dscale, void* g, int ldg, int g_bf16, hipStream_t st);

// graph-local chain (aggregate.hip: chain_kernel): all launches between the front kernel and the weight-gradient GEMM as phases
// of ONE launch, one workgroup per graph.  The argument block lives in device memory (several KB).
constexpr int CHAIN_MAX_LAYERS = 4;
struct ChainArgs {
  int L, lds_stride;
  int gs_last, gs_first;  // row-group widths of the last forward layer / the layer-0 transposed aggregation
  const int64_t* ptr[HMP_MAX_NODE_TYPES];  // [n_graphs + 1] row offsets of every node type (null: type without rows here)
  AggArgs fwd[CHAIN_MAX_LAYERS];
  int fwd_type[CHAIN_MAX_LAYERS][HMP_MAX_NODE_TYPES];  // node type of fwd[l].d[i]
  TAggArgs bwd[CHAIN_MAX_LAYERS];                       // bwd[l] = layer l
  int bwd_type[CHAIN_MAX_LAYERS][HMP_MAX_NODE_TYPES];
  unsigned* ticket;
  const float* fin_row_lv;
  float* fin_out2;
  NetState* fin_state;
};
int chain_launch(const ChainArgs* d_args, int n_graphs, int fin_rows, int gs, size_t lds_bytes, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// parameter packing / gradient un-packing (tables live in device memory, built at bind time)
// ---------------------------------------------------------------------------------------------
// kinds of packed rows
//  PACK_SUM     row r            = sum_q P[src[q]][r, :]                       (SAGE weights, root sums, bias sums)
//  PACK_HEADS   row h*Cp + c     = P[src[0]][h*C + c, :] (c < C), else 0       (GAT lin_src with heads padded to 4)
//  PACK_ATTDOT  row h            = sum_c P[att][h*C + c] * P[src[0]][h*C + c, :]   (V^T = fold of att into lin: a = x * V)
//  PACK_ATTDOT_T element (d, h)  = sum_c P[att][h*C + c] * P[src[0]][(h*C + c), d] (V_edge [edge_dim][H])
enum { PACK_SUM = 0, PACK_HEADS = 1, PACK_ATTDOT = 2, PACK_ATTDOT_T = 3 };
struct PackSeg {
  int64_t dst;      // float offset into the packed buffer
  int rows, rows_pad, cols, ld_dst;
  int ld_src;       // == cols (parameters are dense)
  int nsrc;
  int kind, H, C, Cp;
  int64_t att;      // PACK_ATTDOT*: float offset of the attention vector [H*C]
  int64_t src[AGG_MAX_IN];  // float offsets into the flat parameter buffer (summed)
};
// block -> segment map of the table-driven kernels: segment i owns blocks [start[i], start[i+1]); passed by value so the
// lookup is a scan of scalar (kernarg) loads instead of a dependent chain of global loads
constexpr int SEG_MAX = 400;
struct SegBlocks {
  int n;
  int start[SEG_MAX + 1];
};
// step_ctr != null: block 0 increments *step_ctr (the fused step starts with the pack)
int pack_launch(const PackSeg* d_segs, const SegBlocks& sb, const float* d_params, float* d_packed, int* step_ctr, int* step_mirror,
                hipStream_t st);

// A parameter gradient element (r, c) is the sum of up to 3 terms read from split-K slabs S (summed over slabs):
//  GT_COPY       S[(r/C*Cp + r%C) * ld + c]                         (rows of the stacked operand; C = Cp: identity)
//  GT_ATT_OUTER  P[att + r] * S[(r/C) * ld + c]                     (d lin from a = x * fold(att, lin))
//  GT_ATT_DOT    sum_f S[(r/C) * ld + f] * P[w + r * ldw + f]       (d att; parameter element r = h*C + c, cols = 1)
//  GT_ATT_OUTER_T  P[att + r] * S[c * ld + r/C]                     (d lin_edge from V_edge [D][H])
//  GT_ATT_DOT_T    sum_d S[d * ld + r/C] * P[w + r * ldw + d]       (d att_edge)
enum { GT_COPY = 0, GT_ATT_OUTER = 1, GT_ATT_DOT = 2, GT_ATT_OUTER_T = 3, GT_ATT_DOT_T = 4 };
struct GradTerm {
  int kind;
  int slab_id;      // index into the per-call n_slabs / slab_stride arrays
  int64_t src;      // float offset of the term's origin inside slab 0
  int ld;
  int C, Cp;
  int inner;        // GT_ATT_DOT*: length of the dot product
  int64_t att, w;   // parameter offsets
  int ldw;
  float scale;
};
struct GradSeg {
  int64_t dst;       // float offset into the flat gradient buffer
  int rows, cols;    // parameter shape (vectors: cols = 1)
  int n_terms;
  GradTerm t[3];
};
// slab ids of layer l: l*SLAB_IDS_PER_LAYER + {t: stacked weights of node type t | 8 + t: bias column sums of
// node type t (GAT) | 16 + c: V_edge of conv c (GAT_edge)}
constexpr int SLAB_IDS_PER_LAYER = 2 * HMP_MAX_NODE_TYPES + HMP_MAX_CONVS;
constexpr int GRAD_MAX_SLAB_IDS = HMP_MAX_LAYERS * SLAB_IDS_PER_LAYER;
struct GradReduceDyn {  // passed by value: keep it small
  unsigned char n_slabs[GRAD_MAX_SLAB_IDS];
  int slab_stride[GRAD_MAX_SLAB_IDS];
};

// ---------------------------------------------------------------------------------------------
// K3: GAT edge softmax + aggregation
// ---------------------------------------------------------------------------------------------
constexpr int GAT_HMAX = 8;
constexpr int GAT_MAX_EDIM = 4;

// Static (per bound network) description of one GAT layer, resident in device memory: the by-value kernel
// argument budget (4 KB) cannot hold it.  Everything that changes with the batch travels in GatDyn.
struct GatInS {
  int et, src_t;
  const int *rowptr, *col, *eid, *t_rowptr, *t_col, *t_pos;
  const float* z;        // projected source rows: h_s (H*Cp) at column hoff
  int ldz, hoff;
  const float* za;       // rows holding a_s (H) at column asoff (the executor: za == z)
  int ldza, asoff;
  const float* zd;       // projected destination rows: a_d (H) at column adoff
  int ldzd, adoff;
  const float* vedge;    // [edim][GAT_HMAX] fold of att_edge into lin_edge, or null
  int edim;
  int self_loops;
  float *smax, *sden;    // [cap_dst, GAT_HMAX] per-row running max / softmax denominator (+1e-16)
  float adrop_p;
  uint32_t adrop_stream;
  float *alpha_drop, *dlogit;  // [cap_E + cap_loop, GAT_HMAX]   (backward)
  float* dlogit_orig;    // [cap_E, GAT_HMAX] original edge order (edge_attr convs only)
  float* dza_src;        // rows receiving d a_s at column asoff (the executor: the source dZ)
  int lddza_src;
  float* dz_dst;         // destination dZ (d a_d at adoff)
  int lddz_dst;
};
struct GatDstS {
  int t;
  int H, C, Cp, concat;
  float* out;
  int ldo;
  const float* bias;     // sum of the convs' biases, width of the output
  int act;
  float drop_p;
  uint32_t drop_stream;
  float group_scale;     // 1 (HeteroConv sum) or 1/n_convs (mean)
  const float* g;        // gradient of this layer's output for type t (null: taken from GatDyn.g_top)
  int ldg;
  int n_in;
  GatInS in[AGG_MAX_IN];
};
struct GatSrcS {
  int t;
  int n_out;
  struct { int d, i; } out[AGG_MAX_IN];  // (destination entry, in-conv index) pairs leaving this source type
  float* dz;
  int lddz;
};
struct GatLayerS {
  int n_dst, n_src;
  GatDstS d[HMP_MAX_NODE_TYPES];
  GatSrcS s[HMP_MAX_NODE_TYPES];
};
struct GatDyn {
  int n_nodes[HMP_MAX_NODE_TYPES];
  long long n_edges[HMP_MAX_EDGE_TYPES];
  const float* edge_attr[HMP_MAX_EDGE_TYPES];
  int block_start[HMP_MAX_NODE_TYPES + 1];
  const float* g_top;    // external output gradient (last layer / autograd path)
  int ld_gtop;
  int training;
  uint32_t k0, k1, step;
  const int* step_dev;
};
int gat_fwd_launch(const GatLayerS* d_tab, const GatLayerS& h_tab, GatDyn& dyn, hipStream_t st);
int gat_bwd1_launch(const GatLayerS* d_tab, const GatLayerS& h_tab, GatDyn& dyn, hipStream_t st);
int gat_bwd2_launch(const GatLayerS* d_tab, const GatLayerS& h_tab, GatDyn& dyn, hipStream_t st);

// Adam riding in the gradient un-pack (single-rank step of networks whose gradient terms read no parameters, i.e. SAGE):
// the thread that produced gradient element i updates parameter i right away.  count = valid labels (device), step = t.
struct AdamFuse {
  int on;
  float* p;
  float* m;
  float* v;
  float lr, b1, b2, eps, wd;
  const int* step_dev;
  const float* count;
};
// row_lv != null: block (0,0) also sums the per-row {loss, valid} pairs (masked_ce_rows_launch) into out2 / state
int grad_reduce_launch(const GradSeg* d_segs, const SegBlocks& sb, const GradReduceDyn& dyn, const float* d_slabs,
                       const float* d_params, float* d_grads, const float* row_lv, int n_lv_rows, float* out2, NetState* state,
                       const AdamFuse& adam, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// loss / adam
// ---------------------------------------------------------------------------------------------
struct NetState {  // device resident
  int step;        // fused steps started so far (bumped by the pack kernel at the head of every step) -- the net's OWN counter;
                   // an optimiser that keeps Adam moments passes its own counter (hmp_train_args::d_step)
  int status;      // bit 0: edge endpoint out of range, bit 1: label out of range
  float loss_sum, count;
  int last_step;   // value of the counter the last started step ran on (its optimiser's, or `step`): what hmp_net_read_state reports
};
int masked_ce_launch(const float* logits, int ldl, int n_rows, int n_classes, const int64_t* labels, int64_t ignored,
                     float* grad, int ldg, float* out2, NetState* state_or_null, hipStream_t st);
int masked_ce_rows_launch(const float* logits, int ldl, int n_rows, int n_classes, const int64_t* labels, int64_t ignored,
                          float* grad, int ldg, float* row_lv, NetState* state, hipStream_t st);
// step_dev != null: t = *step_dev is read on the device (graph replay); else t = step_host
int adam_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                int step_host, const int* step_dev, const float* d_count, hipStream_t st);

}  // namespace hmp
