// Operators of the homogeneous GCN / GIN family: the Stanford3DSG configurations
// baseline_GCN / baseline_GIN / htree_GCN / htree_GIN of the reference build
//   GCNConv(in, out, add_self_loops=True)                                   (models/utils.py:15-16)
//   GINConv(Sequential(Linear, ReLU, Linear), eps=0, train_eps=True)        (models/utils.py:17-26)
//   BatchNorm(hidden_dim) between a GIN conv and its ReLU                    (models/homogeneous_network.py:93-97,133-134)
// Their graphs are tiny (2..27 nodes each, 128 per batch), so these are plain one-pass kernels composed op by op from the host
// (hydra_gnn_amd/ops.py); the dense products go through hmp_gemm_f32.
//
// Both convolutions are linear in x before their first Linear, so the host projects first (Z = X W^T, the narrow side) and
// the neighbourhood sum below runs at the OUTPUT width:
//   GCN:  out_i = d_i^-1/2 ( sum_{j->i, j != i} d_j^-1/2 z_j  +  d_i^-1/2 z_i ),   d_i = 1 + #{j->i, j != i}
//         ([PyG] gcn_norm with add_remaining_self_loops: existing loops are replaced by exactly one loop of weight 1)
//   GIN:  out_i = sum_{j->i} z_j + (1 + eps) z_i
// The transposed operator (gradient w.r.t. z) has the same form on the CSC lists, so one kernel serves both directions.
#include "common.h"

namespace hmp {

constexpr int WS_MAXC = 4;  // columns per lane and pass

template <int GS>
__global__ __launch_bounds__(256) void segment_wsum_kernel(const float* __restrict__ x, int ldx, int F, const int* __restrict__ rowptr,
                                                           const int* __restrict__ col, int n, const float* __restrict__ w,
                                                           const float* __restrict__ eps_dev, int skip_self, float* __restrict__ out,
                                                           int ldo) {
  const int row = blockIdx.x * (256 / GS) + (int)threadIdx.x / GS;
  const int lane = (int)threadIdx.x % GS;
  if (row >= n) return;
  const int beg = rowptr[row], end = rowptr[row + 1];
  const float wi = w ? w[row] : 1.f;
  const float self_c = w ? wi : 1.f + (eps_dev ? *eps_dev : 0.f);
  for (int c0 = 0; c0 < F; c0 += GS * WS_MAXC) {
    float acc[WS_MAXC];
#pragma unroll
    for (int i = 0; i < WS_MAXC; ++i) acc[i] = 0.f;
    for (int k = beg; k < end; ++k) {
      const int j = col[k];
      if (j < 0 || (skip_self && j == row)) continue;  // dropped edge (plan: out-of-range endpoint) / replaced self loop
      const float wj = w ? w[j] : 1.f;
      const float* xr = x + (int64_t)j * ldx;
#pragma unroll
      for (int i = 0; i < WS_MAXC; ++i) {
        const int c = c0 + lane + i * GS;
        if (c < F) acc[i] += wj * xr[c];
      }
    }
    const float* xs = x + (int64_t)row * ldx;
#pragma unroll
    for (int i = 0; i < WS_MAXC; ++i) {
      const int c = c0 + lane + i * GS;
      if (c < F) out[(int64_t)row * ldo + c] = wi * (acc[i] + self_c * xs[c]);
    }
  }
}

__global__ __launch_bounds__(256) void gcn_norm_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, int n,
                                                       float* __restrict__ dinv) {
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (row >= n) return;
  int deg = 1;
  for (int k = rowptr[row]; k < rowptr[row + 1]; ++k) {
    const int j = col[k];
    deg += (j >= 0 && j != row) ? 1 : 0;
  }
  dinv[row] = 1.f / sqrtf((float)deg);
}

// y = dropout(act(x + bias)), act = HMP_ACT_NONE / RELU / ELU; the keep-mask is the engine's (hmp_dropout_mask replays it: element
// (row, col) of an [n, F] tensor).  A dropped element is stored as -0.0f, a kept zero as +0.0f (the engine's convention), so the
// backward reads everything it needs off y.
__global__ __launch_bounds__(256) void bias_act_drop_kernel(const float* __restrict__ x, int ldx, int n, int F,
                                                            const float* __restrict__ bias, int act, int drop_on, DropCfg cfg,
                                                            float* __restrict__ y, int ldy) {
  const int qpr = (F + 3) >> 2;
  const int64_t total = (int64_t)n * qpr;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(q / qpr), c = (int)(q % qpr) * 4;
    bool keep[4] = {true, true, true, true};
    if (drop_on) drop_keep4(cfg, (uint32_t)q, keep);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (c + i >= F) continue;
      float v = x[(int64_t)row * ldx + c + i] + (bias ? bias[c + i] : 0.f);
      if (act == HMP_ACT_RELU) v = fmaxf(v, 0.f);
      else if (act == HMP_ACT_ELU) v = v > 0.f ? v : expm1f(v);
      if (drop_on) v = keep[i] ? (v * cfg.scale + 0.0f) : -0.0f;
      else if (act != HMP_ACT_NONE) v = v + 0.0f;  // a kept -0.0 becomes +0.0: the sign of zero is reserved for "dropped"
      y[(int64_t)row * ldy + c + i] = v;
    }
  }
}

// gx = g * d y / d x from the forward OUTPUT y (scale = 1/(1-p), 1 without dropout):
//   dropped (y is -0.0f) -> 0;  relu: y > 0 ? g*scale : 0;  elu: y > 0 ? g*scale : g*(y + scale)   [y = scale*elu(x), elu' = elu + 1]
__global__ __launch_bounds__(256) void bias_act_drop_bwd_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ y, int ldy,
                                                                int n, int F, int act, float scale, float* __restrict__ gx, int ldgx) {
  const int64_t total = (int64_t)n * F;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(e / F), c = (int)(e % F);
    const float gv = g[(int64_t)row * ldg + c];
    float r = gv;
    if (act != HMP_ACT_NONE) {
      const float yv = y[(int64_t)row * ldy + c];
      if (__float_as_uint(yv) == 0x80000000u) r = 0.f;
      else if (yv > 0.f) r = gv * scale;
      else r = act == HMP_ACT_ELU ? gv * (yv + scale) : 0.f;
    }
    gx[(int64_t)row * ldgx + c] = r;
  }
}

// ---- column reductions: a block owns 32 columns; 8 row groups stride the rows, partials meet in LDS in a fixed order ------------
__device__ __forceinline__ float col_reduce8(float v, float (*red)[32], int rg, int cl) {
  red[rg][cl] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += red[i][cl];
  __syncthreads();
  return s;
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ g, int ldg, int n, int F, float* __restrict__ out) {
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < F)
    for (int r = rg; r < n; r += 8) s += g[(int64_t)r * ldg + c];
  s = col_reduce8(s, red, rg, cl);
  if (rg == 0 && c < F) out[c] = s;
}

// sum_{r,c} a[r,c] * b[r,c]  (gradient of GIN's eps); one block, fixed order
__global__ __launch_bounds__(1024) void rowdot_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, int n,
                                                      int F, float* __restrict__ out) {
  __shared__ float red[1024];
  const int64_t total = (int64_t)n * F;
  float s = 0.f;
  for (int64_t e = threadIdx.x; e < total; e += 1024) {
    const int r = (int)(e / F), c = (int)(e % F);
    s += a[(int64_t)r * lda + c] * b[(int64_t)r * ldb + c];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int h = 512; h > 0; h >>= 1) {
    if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = red[0];
}

// torch.nn.BatchNorm1d (wrapped by [PyG] BatchNorm): batch statistics when training (biased variance normalises, the unbiased
// one updates running_var), running statistics otherwise.  save[0][c] = mean used, save[1][c] = 1/sqrt(var + eps).
__global__ __launch_bounds__(256) void batchnorm_fwd_kernel(const float* __restrict__ x, int ldx, int n, int F,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ run_mean, float* __restrict__ run_var, float momentum,
                                                            float eps, int training, float* __restrict__ y, int ldy,
                                                            float* __restrict__ save) {
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const bool live = c < F;
  float mean, var;
  if (training) {
    float s = 0.f;
    if (live)
      for (int r = rg; r < n; r += 8) s += x[(int64_t)r * ldx + c];
    mean = col_reduce8(s, red, rg, cl) / (float)n;
    float q = 0.f;
    if (live)
      for (int r = rg; r < n; r += 8) {
        const float d = x[(int64_t)r * ldx + c] - mean;
        q += d * d;
      }
    const float ss = col_reduce8(q, red, rg, cl);
    var = ss / (float)n;
    if (live && rg == 0) {
      run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
      run_var[c] = (1.f - momentum) * run_var[c] + momentum * (n > 1 ? ss / (float)(n - 1) : var);
    }
  } else {
    mean = live ? run_mean[c] : 0.f;
    var = live ? run_var[c] : 1.f;
  }
  const float invstd = 1.f / sqrtf(var + eps);
  if (live) {
    if (rg == 0) {
      save[c] = mean;
      save[F + c] = invstd;
    }
    const float ga = gamma[c], be = beta[c];
    for (int r = rg; r < n; r += 8) y[(int64_t)r * ldy + c] = (x[(int64_t)r * ldx + c] - mean) * invstd * ga + be;
  }
}

__global__ __launch_bounds__(256) void batchnorm_bwd_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ x, int ldx,
                                                            int n, int F, const float* __restrict__ gamma, const float* __restrict__ save,
                                                            int training, float* __restrict__ gx, int ldgx, float* __restrict__ ggamma,
                                                            float* __restrict__ gbeta) {
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const bool live = c < F;
  const float mean = live ? save[c] : 0.f, invstd = live ? save[F + c] : 0.f;
  float sg = 0.f, sgx = 0.f;
  if (live)
    for (int r = rg; r < n; r += 8) {
      const float gv = g[(int64_t)r * ldg + c];
      sg += gv;
      sgx += gv * (x[(int64_t)r * ldx + c] - mean) * invstd;
    }
  sg = col_reduce8(sg, red, rg, cl);
  sgx = col_reduce8(sgx, red, rg, cl);
  if (!live) return;
  if (rg == 0) {
    ggamma[c] = sgx;
    gbeta[c] = sg;
  }
  const float ga = gamma[c] * invstd;
  const float mg = training ? sg / (float)n : 0.f, mgx = training ? sgx / (float)n : 0.f;
  for (int r = rg; r < n; r += 8) {
    const float xh = (x[(int64_t)r * ldx + c] - mean) * invstd;
    gx[(int64_t)r * ldgx + c] = ga * (g[(int64_t)r * ldg + c] - mg - xh * mgx);
  }
}

static int grid_for(int64_t total) {
  const int64_t want = cdiv(total, 256);
  return (int)(want > 4096 ? 4096 : (want < 1 ? 1 : want));
}

}  // namespace hmp

using namespace hmp;

extern "C" int hmp_gcn_norm(hmp_plan plan, float* d_dinv, void* stream) {
  HMP_CHECK_ARG(d_dinv && plan.n_src == plan.n_dst, "hmp_gcn_norm: needs a square graph (n_src == n_dst) and an output");
  if (plan.n_dst == 0) return HMP_OK;
  hipLaunchKernelGGL(gcn_norm_kernel, dim3(cdiv(plan.n_dst, 256)), dim3(256), 0, (hipStream_t)stream, plan.d_rowptr, plan.d_col,
                     plan.n_dst, d_dinv);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

extern "C" int hmp_segment_wsum(const float* d_x, int32_t ldx, int32_t F, hmp_plan plan, int32_t transpose, const float* d_w,
                                const float* d_eps, float* d_out, int32_t ldo, void* stream) {
  HMP_CHECK_ARG(d_x && d_out && F > 0 && ldx >= F && ldo >= F, "hmp_segment_wsum: bad argument");
  HMP_CHECK_ARG(plan.n_src == plan.n_dst, "hmp_segment_wsum: needs a square graph (the self term reads x[i])");
  HMP_CHECK_ARG(d_x != d_out, "hmp_segment_wsum: in-place is not supported");
  const int n = plan.n_dst;
  if (n == 0) return HMP_OK;
  const int* rp = transpose ? plan.d_t_rowptr : plan.d_rowptr;
  const int* cl = transpose ? plan.d_t_col : plan.d_col;
  const int skip = d_w ? 1 : 0;
  hipStream_t st = (hipStream_t)stream;
#define WS_LAUNCH(GS)                                                                                                           \
  hipLaunchKernelGGL(segment_wsum_kernel<GS>, dim3(cdiv(n, 256 / GS)), dim3(256), 0, st, d_x, ldx, F, rp, cl, n, d_w, d_eps, skip, \
                     d_out, ldo)
  if (F <= 8) WS_LAUNCH(8);
  else if (F <= 16) WS_LAUNCH(16);
  else if (F <= 32) WS_LAUNCH(32);
  else WS_LAUNCH(64);
#undef WS_LAUNCH
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

extern "C" int hmp_bias_act_drop_fwd(const float* d_x, int32_t ldx, int32_t n_rows, int32_t F, const float* d_bias, int32_t act,
                                     float p, uint64_t seed, uint32_t rng_step, uint32_t rng_stream, float* d_y, int32_t ldy,
                                     void* stream) {
  HMP_CHECK_ARG(d_x && d_y && n_rows >= 0 && F > 0 && ldx >= F && ldy >= F && p >= 0.f && p < 1.f, "hmp_bias_act_drop_fwd: bad argument");
  HMP_CHECK_ARG(act == HMP_ACT_NONE || act == HMP_ACT_RELU || act == HMP_ACT_ELU, "hmp_bias_act_drop_fwd: unknown activation %d", act);
  HMP_CHECK_ARG(p == 0.f || act != HMP_ACT_NONE, "hmp_bias_act_drop_fwd: dropout is only defined after an activation (the reference never drops a raw sum)");
  if (n_rows == 0) return HMP_OK;
  DropCfg cfg;
  cfg.k0 = (uint32_t)seed; cfg.k1 = (uint32_t)(seed >> 32);
  cfg.step = rng_step; cfg.stream = rng_stream;
  cfg.thresh = drop_thresh(p); cfg.scale = 1.f / (1.f - p);
  cfg.step_dev = nullptr;
  hipLaunchKernelGGL(bias_act_drop_kernel, dim3(grid_for((int64_t)n_rows * ((F + 3) >> 2))), dim3(256), 0, (hipStream_t)stream, d_x, ldx,
                     n_rows, F, d_bias, act, p > 0.f ? 1 : 0, cfg, d_y, ldy);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

extern "C" int hmp_bias_act_drop_bwd(const float* d_g, int32_t ldg, const float* d_y, int32_t ldy, int32_t n_rows, int32_t F,
                                     int32_t act, float p, float* d_gx, int32_t ldgx, void* stream) {
  HMP_CHECK_ARG(d_g && d_gx && (d_y || act == HMP_ACT_NONE) && n_rows >= 0 && F > 0 && ldg >= F && ldgx >= F && p >= 0.f && p < 1.f,
                "hmp_bias_act_drop_bwd: bad argument");
  if (n_rows == 0) return HMP_OK;
  hipLaunchKernelGGL(bias_act_drop_bwd_kernel, dim3(grid_for((int64_t)n_rows * F)), dim3(256), 0, (hipStream_t)stream, d_g, ldg, d_y,
                     ldy, n_rows, F, act, 1.f / (1.f - p), d_gx, ldgx);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

extern "C" int hmp_colsum(const float* d_g, int32_t ldg, int32_t n_rows, int32_t F, float* d_out, void* stream) {
  HMP_CHECK_ARG(d_g && d_out && n_rows >= 0 && F > 0 && ldg >= F, "hmp_colsum: bad argument");
  hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(F, 32)), dim3(256), 0, (hipStream_t)stream, d_g, ldg, n_rows, F, d_out);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

extern "C" int hmp_rowdot_sum(const float* d_a, int32_t lda, const float* d_b, int32_t ldb, int32_t n_rows, int32_t F, float* d_out,
                              void* stream) {
  HMP_CHECK_ARG(d_a && d_b && d_out && n_rows >= 0 && F > 0 && lda >= F && ldb >= F, "hmp_rowdot_sum: bad argument");
  hipLaunchKernelGGL(rowdot_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, d_a, lda, d_b, ldb, n_rows, F, d_out);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

extern "C" int hmp_batchnorm_fwd(const float* d_x, int32_t ldx, int32_t n_rows, int32_t F, const float* d_gamma, const float* d_beta,
                                 float* d_running_mean, float* d_running_var, float momentum, float eps, int32_t training, float* d_y,
                                 int32_t ldy, float* d_save, void* stream) {
  HMP_CHECK_ARG(d_x && d_y && d_gamma && d_beta && d_running_mean && d_running_var && d_save && F > 0 && ldx >= F && ldy >= F,
                "hmp_batchnorm_fwd: bad argument");
  HMP_CHECK_ARG(!training || n_rows > 1, "hmp_batchnorm_fwd: training needs more than one row (torch raises for a single value per channel)");
  if (n_rows == 0) return HMP_OK;
  hipLaunchKernelGGL(batchnorm_fwd_kernel, dim3(cdiv(F, 32)), dim3(256), 0, (hipStream_t)stream, d_x, ldx, n_rows, F, d_gamma, d_beta,
                     d_running_mean, d_running_var, momentum, eps, training, d_y, ldy, d_save);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

extern "C" int hmp_batchnorm_bwd(const float* d_g, int32_t ldg, const float* d_x, int32_t ldx, int32_t n_rows, int32_t F,
                                 const float* d_gamma, const float* d_save, int32_t training, float* d_gx, int32_t ldgx, float* d_ggamma,
                                 float* d_gbeta, void* stream) {
  HMP_CHECK_ARG(d_g && d_x && d_gamma && d_save && d_gx && d_ggamma && d_gbeta && F > 0 && ldg >= F && ldx >= F && ldgx >= F,
                "hmp_batchnorm_bwd: bad argument");
  if (n_rows == 0) return HMP_OK;
  hipLaunchKernelGGL(batchnorm_bwd_kernel, dim3(cdiv(F, 32)), dim3(256), 0, (hipStream_t)stream, d_g, ldg, d_x, ldx, n_rows, F, d_gamma,
                     d_save, training, d_gx, ldgx, d_ggamma, d_gbeta);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}
