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
};

constexpr int GEMM_MAX_PROB = 8;
struct GemmBatch {
  int n;
  int total_tiles;
  GemmProblem p[GEMM_MAX_PROB];
};

// chooses tile shape + split-K (only when want_split: C is then a stack of slabs) and launches
int gemm_launch(GemmBatch& gb, bool want_split, int max_slabs, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------------
struct PlanJob {
  const int64_t* ei;  // [2][E]
  int64_t E;
  int n_src, n_dst;
  int *rowptr, *col, *eid, *t_rowptr, *t_col, *t_pos;
  // scratch
  int *cnt_in, *cnt_out, *cur_in, *cur_out;  // zeroed before the launch (one contiguous block)
  int *tmp_in, *tmp_out, *t_eid, *pos_of_eid;
};
struct PlanBatch {
  int n;
  int64_t edge_start[HMP_MAX_EDGE_TYPES + 1];
  int64_t row_start[2 * HMP_MAX_EDGE_TYPES + 1];  // rows of (job, dir): dir 0 = by dst, 1 = by src
  PlanJob j[HMP_MAX_EDGE_TYPES];
};
size_t plan_scratch_ints(int64_t E, int n_src, int n_dst);
// carve scratch for job (all int32); zero_begin/zero_ints return the region to memset
void plan_carve(PlanJob& job, int* scratch);
int plan_launch(PlanBatch& pb, int* d_status, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// K1 + fused SAGE aggregation
// ---------------------------------------------------------------------------------------------
constexpr int AGG_MAX_IN = 6;
struct AggIn {
  const int* rowptr;
  const int* col;
  const float* z;  // projected source rows
  int ldz, coff;
};
struct AggDst {
  int n_rows, F;
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
  AggIn in[AGG_MAX_IN];
};
struct AggArgs {
  int n;
  int total_blocks;
  int mean;  // divide every gather by max(deg,1)
  AggDst d[HMP_MAX_NODE_TYPES];
};
int agg_fwd_launch(AggArgs& a, hipStream_t st);

struct TAggOut {
  const int* t_rowptr;
  const int* t_col;
  const int* rowptr;  // forward CSR rowptr of the same edge type (for 1/deg of the destination)
  const float* g;     // gradient rows of the destination type
  int ldg, coff, F;
};
struct TAggSrc {
  int n_rows;
  float* dz;
  int lddz, ncols;     // ncols (padded): columns not covered by a segment are zeroed
  const float* groot;  // may be null
  int ldgr, roff, Froot;
  int n_out;
  int block_start;
  TAggOut out[AGG_MAX_IN];
};
struct TAggArgs {
  int n;
  int total_blocks;
  int mean;
  TAggSrc s[HMP_MAX_NODE_TYPES];
};
int agg_bwd_launch(TAggArgs& a, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// parameter packing / gradient un-packing (tables live in device memory, built at bind time)
// ---------------------------------------------------------------------------------------------
struct PackSeg {
  int64_t dst;      // float offset into the packed buffer
  int rows, rows_pad, cols, ld_dst;
  int ld_src;       // == cols (parameters are dense)
  int nsrc;
  int64_t src[AGG_MAX_IN];  // float offsets into the flat parameter buffer (summed)
};
int pack_launch(const PackSeg* d_segs, int n_segs, int64_t total_rows, const int64_t* d_row_start, const float* d_params,
                float* d_packed, hipStream_t st);

struct GradSeg {
  int64_t dst;       // float offset into the flat gradient buffer
  int rows, cols;    // parameter shape (bias: cols = 1)
  int64_t src;       // float offset of element (0,0) inside slab 0 of the packed-gradient buffer
  int ld_src;
  int slab_id;       // index into the per-call n_slabs / slab_stride arrays
};
struct GradReduceDyn {
  int n_slabs[HMP_MAX_LAYERS * HMP_MAX_NODE_TYPES];
  int64_t slab_stride[HMP_MAX_LAYERS * HMP_MAX_NODE_TYPES];
};
int grad_reduce_launch(const GradSeg* d_segs, int n_segs, int64_t total_elems, const int64_t* d_elem_start,
                       const GradReduceDyn& dyn, const float* d_slabs, float* d_grads, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// loss / adam
// ---------------------------------------------------------------------------------------------
struct NetState {  // device resident
  int step;        // completed optimiser steps
  int status;      // bit 0: edge endpoint out of range, bit 1: label out of range
  float loss_sum, count;
};
int masked_ce_launch(const float* logits, int ldl, int n_rows, int n_classes, const int64_t* labels, int64_t ignored,
                     float* grad, int ldg, float* out2, NetState* state_or_null, hipStream_t st);
// step_dev != null: t = *step_dev + 1 is read on the device (graph replay); else t = step_host
int adam_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                int step_host, const int* step_dev, const float* d_count, hipStream_t st);
int step_increment_launch(NetState* state, hipStream_t st);

}  // namespace hmp
