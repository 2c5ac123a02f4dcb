// Loss, optimiser and the parameter <-> packed-operand shuffles around the layer kernels.
#include "kernels.h"

namespace hmp {

// ---------------------------------------------------------------------------------------------
// masked cross entropy (models/utils.py:143-148 with mask = label != ignored): one block, one
// wavefront per row, fixed-order reductions => deterministic.  grad is the gradient of the SUM loss.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void masked_ce_kernel(const float* __restrict__ logits, int ldl, int n_rows, int n_classes,
                                                         const int64_t* __restrict__ labels, int64_t ignored, float* __restrict__ grad,
                                                         int ldg, float* __restrict__ out2, NetState* state) {
  __shared__ float s_loss[16];
  __shared__ float s_cnt[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float loss = 0.f, cnt = 0.f;
  int bad = 0;
  for (int row = w; row < n_rows; row += 16) {
    const float* lr = logits + (int64_t)row * ldl;
    const int64_t y = labels[row];
    const bool valid = (y != ignored);
    if (valid && (y < 0 || y >= n_classes)) bad = 1;
    float m = -INFINITY;
    for (int c = lane; c < n_classes; c += 64) m = fmaxf(m, lr[c]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
    for (int c = lane; c < n_classes; c += 64) s += expf(lr[c] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float lse = m + logf(s);
    const bool use = valid && y >= 0 && y < n_classes;
    if (grad) {
      for (int c = lane; c < ldg; c += 64) {
        float g = 0.f;
        if (use && c < n_classes) g = expf(lr[c] - lse) - (c == (int)y ? 1.f : 0.f);
        grad[(int64_t)row * ldg + c] = g;
      }
    }
    if (use && lane == 0) {
      loss += lse - lr[y];
      cnt += 1.f;
    }
  }
  if (lane == 0) { s_loss[w] = loss; s_cnt[w] = cnt; }
  if (bad && state) atomicOr(&state->status, 2);
  __syncthreads();
  if (threadIdx.x == 0) {
    float L = 0.f, C = 0.f;
    for (int i = 0; i < 16; ++i) { L += s_loss[i]; C += s_cnt[i]; }
    out2[0] = L;
    out2[1] = C;
    if (state) { state->loss_sum = L; state->count = C; }
  }
}

int masked_ce_launch(const float* logits, int ldl, int n_rows, int n_classes, const int64_t* labels, int64_t ignored, float* grad,
                     int ldg, float* out2, NetState* state, hipStream_t st) {
  hipLaunchKernelGGL(masked_ce_kernel, dim3(1), dim3(1024), 0, st, logits, ldl, n_rows, n_classes, labels, ignored, grad, ldg, out2, state);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

// ---------------------------------------------------------------------------------------------
// Adam with coupled L2 (torch.optim.Adam semantics, base_training_job.py:181-185)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                                                   int step_host, const int* __restrict__ step_dev, const float* __restrict__ d_count) {
  const int t = step_dev ? (*step_dev + 1) : step_host;
  const float bc1 = 1.f - powf(b1, (float)t);
  const float bc2 = 1.f - powf(b2, (float)t);
  const float step_size = lr / bc1;
  const float bc2_sqrt = sqrtf(bc2);
  float gscale = 1.f;
  if (d_count) {
    const float c = *d_count;
    gscale = 1.f / (c > 1.f ? c : 1.f);
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float pi = p[i];
    const float gi = g[i] * gscale + wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - step_size * (mi / denom);
  }
}

int adam_launch(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                int step_host, const int* step_dev, const float* d_count, hipStream_t st) {
  if (n == 0) return HMP_OK;
  const int64_t want = cdiv(n, 256);
  const int grid = (int)(want > 2048 ? 2048 : want);
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd, step_host, step_dev, d_count);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

__global__ void step_increment_kernel(NetState* s) { s->step += 1; }
int step_increment_launch(NetState* state, hipStream_t st) {
  hipLaunchKernelGGL(step_increment_kernel, dim3(1), dim3(1), 0, st, state);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

// ---------------------------------------------------------------------------------------------
// dropout keep-mask export (tests replay the engine's masks in the oracle)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dropout_mask_kernel(DropCfg cfg, int n_rows, int F, uint8_t* __restrict__ mask) {
  const int qpr = (F + 3) >> 2;  // quads per row of the padded row
  const int64_t total = (int64_t)n_rows * qpr;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
    const int row = (int)(q / qpr), c = (int)(q % qpr) * 4;
    bool keep[4];
    drop_keep4(cfg, (uint32_t)q, keep);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (c + i < F) mask[(int64_t)row * F + c + i] = keep[i] ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------------------------
// pack: flat parameters -> per-(layer, source type) stacked weight operand.  One thread block row per
// packed row; a packed row is the SUM of up to AGG_MAX_IN parameter rows (root weights of every conv
// reaching the node type add up: sum_e W_r,e * x == (sum_e W_r,e) * x) or zero padding.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_kernel(const PackSeg* __restrict__ segs, int n_segs, const int64_t* __restrict__ row_start,
                                                   const float* __restrict__ params, float* __restrict__ packed) {
  const int64_t total_rows = row_start[n_segs];
  for (int64_t gr = blockIdx.x; gr < total_rows; gr += gridDim.x) {
    int lo = 0, hi = n_segs - 1;  // last seg with row_start <= gr
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (row_start[mid] <= gr) lo = mid; else hi = mid - 1;
    }
    const PackSeg S = segs[lo];
    const int r = (int)(gr - row_start[lo]);
    float* dst = packed + S.dst + (int64_t)r * S.ld_dst;
    for (int c = threadIdx.x; c < S.ld_dst; c += blockDim.x) {
      float v = 0.f;
      if (S.kind == PACK_SUM) {
        if (r < S.rows && c < S.cols)
          for (int q = 0; q < S.nsrc; ++q) v += params[S.src[q] + (int64_t)r * S.ld_src + c];
      } else if (S.kind == PACK_HEADS) {
        const int h = r / S.Cp, cc = r % S.Cp;
        if (h < S.H && cc < S.C && c < S.cols) v = params[S.src[0] + (int64_t)(h * S.C + cc) * S.ld_src + c];
      } else if (S.kind == PACK_ATTDOT) {  // row r = head
        if (r < S.H && c < S.cols)
          for (int cc = 0; cc < S.C; ++cc)
            v += params[S.att + r * S.C + cc] * params[S.src[0] + (int64_t)(r * S.C + cc) * S.ld_src + c];
      } else {  // PACK_ATTDOT_T: row r = edge-attribute dimension d, column c = head
        if (r < S.rows && c < S.H)
          for (int cc = 0; cc < S.C; ++cc)
            v += params[S.att + c * S.C + cc] * params[S.src[0] + (int64_t)(c * S.C + cc) * S.ld_src + r];
      }
      dst[c] = v;
    }
  }
}

int pack_launch(const PackSeg* d_segs, int n_segs, int64_t total_rows, const int64_t* d_row_start, const float* d_params,
                float* d_packed, hipStream_t st) {
  if (n_segs == 0 || total_rows == 0) return HMP_OK;
  const int grid = (int)(total_rows > 4096 ? 4096 : total_rows);
  hipLaunchKernelGGL(pack_kernel, dim3(grid), dim3(256), 0, st, d_segs, n_segs, d_row_start, d_params, d_packed);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

// ---------------------------------------------------------------------------------------------
// grad reduce: split-K slabs of the packed weight gradients -> flat gradient buffer (fixed slab order).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grad_reduce_kernel(const GradSeg* __restrict__ segs, int n_segs, const int64_t* __restrict__ elem_start,
                                                          const GradReduceDyn dyn, const float* __restrict__ slabs,
                                                          const float* __restrict__ params, float* __restrict__ grads) {
  const int64_t total = elem_start[n_segs];
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
    int lo = 0, hi = n_segs - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (elem_start[mid] <= g) lo = mid; else hi = mid - 1;
    }
    const GradSeg& S = segs[lo];
    const int64_t i = g - elem_start[lo];
    const int r = (int)(i / S.cols), c = (int)(i % S.cols);
    float total = 0.f;
    for (int ti = 0; ti < S.n_terms; ++ti) {
      const GradTerm& T = S.t[ti];
      const int ns = dyn.n_slabs[T.slab_id];
      const int64_t stride = dyn.slab_stride[T.slab_id];
      const float* base = slabs + T.src;
      float v = 0.f;
      if (T.kind == GT_COPY) {
        const float* src = base + (int64_t)((r / T.C) * T.Cp + (r % T.C)) * T.ld + c;
        for (int z = 0; z < ns; ++z) v += src[z * stride];
      } else if (T.kind == GT_ATT_OUTER) {
        const float* src = base + (int64_t)(r / T.C) * T.ld + c;
        float s = 0.f;
        for (int z = 0; z < ns; ++z) s += src[z * stride];
        v = params[T.att + r] * s;
      } else if (T.kind == GT_ATT_DOT) {
        const float* src = base + (int64_t)(r / T.C) * T.ld;
        const float* w = params + T.w + (int64_t)r * T.ldw;
        for (int f = 0; f < T.inner; ++f) {
          float s = 0.f;
          for (int z = 0; z < ns; ++z) s += src[z * stride + f];
          v += s * w[f];
        }
      } else if (T.kind == GT_ATT_OUTER_T) {
        const float* src = base + (int64_t)c * T.ld + (r / T.C);
        float s = 0.f;
        for (int z = 0; z < ns; ++z) s += src[z * stride];
        v = params[T.att + r] * s;
      } else {  // GT_ATT_DOT_T
        const float* w = params + T.w + (int64_t)r * T.ldw;
        for (int d = 0; d < T.inner; ++d) {
          float s = 0.f;
          for (int z = 0; z < ns; ++z) s += base[z * stride + (int64_t)d * T.ld + (r / T.C)];
          v += s * w[d];
        }
      }
      total += v * T.scale;
    }
    grads[S.dst + i] = total;
  }
}

int grad_reduce_launch(const GradSeg* d_segs, int n_segs, int64_t total_elems, const int64_t* d_elem_start, const GradReduceDyn& dyn,
                       const float* d_slabs, const float* d_params, float* d_grads, hipStream_t st) {
  if (n_segs == 0 || total_elems == 0) return HMP_OK;
  const int64_t want = cdiv(total_elems, 256);
  const int grid = (int)(want > 2048 ? 2048 : want);
  hipLaunchKernelGGL(grad_reduce_kernel, dim3(grid), dim3(256), 0, st, d_segs, n_segs, d_elem_start, dyn, d_slabs, d_params, d_grads);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

}  // namespace hmp

extern "C" int hmp_masked_ce(const float* d_logits, int32_t ldl, int32_t n_rows, int32_t n_classes, const int64_t* d_labels,
                             int64_t ignored_label, float* d_grad, int32_t ldg, float* d_out2, void* stream) {
  using namespace hmp;
  HMP_CHECK_ARG(d_logits && d_labels && d_out2, "hmp_masked_ce: null pointer");
  HMP_CHECK_ARG(n_rows >= 0 && n_classes > 0 && ldl >= n_classes && (d_grad == nullptr || ldg >= n_classes), "hmp_masked_ce: bad shape");
  return masked_ce_launch(d_logits, ldl, n_rows, n_classes, d_labels, ignored_label, d_grad, ldg, d_out2, nullptr, (hipStream_t)stream);
}

extern "C" int hmp_adam_flat(float* d_p, const float* d_g, float* d_m, float* d_v, int64_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int32_t step, const float* d_count, void* stream) {
  using namespace hmp;
  HMP_CHECK_ARG(n >= 0 && (n == 0 || (d_p && d_g && d_m && d_v)), "hmp_adam_flat: null pointer");
  HMP_CHECK_ARG(step >= 1, "hmp_adam_flat: step must be >= 1");
  return adam_launch(d_p, d_g, d_m, d_v, n, lr, beta1, beta2, eps, weight_decay, step, nullptr, d_count, (hipStream_t)stream);
}

extern "C" int hmp_dropout_mask(uint64_t seed, uint32_t rng_step, uint32_t rng_stream, float p, int32_t n_rows, int32_t F,
                                uint8_t* d_mask, void* stream) {
  using namespace hmp;
  HMP_CHECK_ARG(d_mask && n_rows >= 0 && F >= 0 && p >= 0.f && p < 1.f, "hmp_dropout_mask: bad argument");
  if (n_rows == 0 || F == 0) return HMP_OK;
  DropCfg cfg;
  cfg.k0 = (uint32_t)seed; cfg.k1 = (uint32_t)(seed >> 32);
  cfg.step = rng_step; cfg.stream = rng_stream;
  cfg.thresh = drop_thresh(p); cfg.scale = 1.f / (1.f - p);
  cfg.step_dev = nullptr;
  const int64_t total = (int64_t)n_rows * ((F + 3) >> 2);
  const int64_t want = cdiv(total, 256);
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((int)(want > 2048 ? 2048 : want)), dim3(256), 0, (hipStream_t)stream, cfg, n_rows, F, d_mask);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}
