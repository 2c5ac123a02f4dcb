// Loss, optimiser and the parameter <-> packed-operand shuffles around the layer kernels.
#include "device_fns.h"

namespace hmp {

// ---------------------------------------------------------------------------------------------
// masked cross entropy (models/utils.py:143-148 with mask = label != ignored).  grad is the gradient of the SUM
// loss.  Unit entry point: one block, one wavefront per row, fixed-order reductions => deterministic.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void ce_row(const float* __restrict__ lr, int n_classes, int64_t y, int64_t ignored, int lane,
                                       float* __restrict__ grow, int ldg, float& loss, float& valid, int& bad) {
  const bool is_valid = (y != ignored);
  if (is_valid && (y < 0 || y >= n_classes)) bad = 1;
  float m = -INFINITY;
  for (int c = lane; c < n_classes; c += 64) m = fmaxf(m, lr[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float s = 0.f;
  for (int c = lane; c < n_classes; c += 64) s += expf(lr[c] - m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float lse = m + logf(s);
  const bool use = is_valid && y >= 0 && y < n_classes;
  if (grow) {
    for (int c = lane; c < ldg; c += 64) {
      float g = 0.f;
      if (use && c < n_classes) g = expf(lr[c] - lse) - (c == (int)y ? 1.f : 0.f);
      grow[c] = g;
    }
  }
  loss = use ? (lse - lr[y]) : 0.f;
  valid = use ? 1.f : 0.f;
}

__global__ __launch_bounds__(1024) void masked_ce_kernel(const float* __restrict__ logits, int ldl, int n_rows, int n_classes,
                                                         const int64_t* __restrict__ labels, int64_t ignored, float* __restrict__ grad,
                                                         int ldg, float* __restrict__ out2, NetState* state) {
  __shared__ float s_loss[16];
  __shared__ float s_cnt[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float loss = 0.f, cnt = 0.f;
  int bad = 0;
  for (int row = w; row < n_rows; row += 16) {
    float l, v;
    ce_row(logits + (int64_t)row * ldl, n_classes, labels[row], ignored, lane, grad ? grad + (int64_t)row * ldg : nullptr, ldg, l, v, bad);
    loss += l;
    cnt += v;
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

// executor version: one wavefront per row over many blocks; per-row {loss, valid} go to row_lv[2*row ..] and are summed in
// fixed order by block (0,0) of the gradient un-pack kernel, which runs later in the same step anyway
__global__ __launch_bounds__(256) void masked_ce_rows_kernel(const float* __restrict__ logits, int ldl, int n_rows, int n_classes,
                                                             const int64_t* __restrict__ labels, int64_t ignored, float* __restrict__ grad,
                                                             int ldg, float* __restrict__ row_lv, NetState* state) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  float l, v;
  int bad = 0;
  ce_row(logits + (int64_t)row * ldl, n_classes, labels[row], ignored, lane, grad + (int64_t)row * ldg, ldg, l, v, bad);
  if (lane == 0) {
    row_lv[2 * row] = l;
    row_lv[2 * row + 1] = v;
    if (bad) atomicOr(&state->status, 2);
  }
}

int masked_ce_rows_launch(const float* logits, int ldl, int n_rows, int n_classes, const int64_t* labels, int64_t ignored, float* grad,
                          int ldg, float* row_lv, NetState* state, hipStream_t st) {
  if (n_rows == 0) return HMP_OK;
  hipLaunchKernelGGL(masked_ce_rows_kernel, dim3(cdiv(n_rows, 4)), dim3(256), 0, st, logits, ldl, n_rows, n_classes, labels, ignored,
                     grad, ldg, row_lv, state);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

// ---------------------------------------------------------------------------------------------
// row argmax: the `.argmax(dim=1)` of the inference server (bin/room_classification_server:286); 16 lanes per row, first
// maximum wins (torch's tie rule for distinct positions), labels as int64 so the buffer can be handed to the caller as is
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int ld, int n_rows, int n_cols,
                                                          int64_t* __restrict__ out) {
  const int row = blockIdx.x * 16 + ((int)threadIdx.x >> 4);
  const int lane = threadIdx.x & 15;
  float best = -INFINITY;
  int arg = 0x7fffffff;
  if (row < n_rows)
    for (int c = lane; c < n_cols; c += 16) {
      const float v = x[(int64_t)row * ld + c];
      if (v > best || arg == 0x7fffffff) { best = v; arg = c; }
    }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 16);
    const int oa = __shfl_xor(arg, o, 16);
    if (oa != 0x7fffffff && (arg == 0x7fffffff || ob > best || (ob == best && oa < arg))) { best = ob; arg = oa; }
  }
  if (row < n_rows && lane == 0) out[row] = arg == 0x7fffffff ? 0 : arg;
}

// ---------------------------------------------------------------------------------------------
// Adam with coupled L2 (torch.optim.Adam semantics, base_training_job.py:181-185)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                                                   int step_host, const int* __restrict__ step_dev, const float* __restrict__ d_count) {
  const int t = step_dev ? *step_dev : step_host;
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
// pack: flat parameters -> per-(layer, source type) stacked weight operand.  grid = (row chunks, segments): a block
// reads its segment descriptor once and writes 4 packed rows (64 lanes per row).  Block (0,0) also starts the step:
// it bumps the device step counter that dropout and Adam read later in the same step.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_kernel(const PackSeg* __restrict__ segs, const SegBlocks sb, const float* __restrict__ params,
                                                   float* __restrict__ packed, int* step_ctr, int* step_mirror) {
  pack_block(segs, sb, params, packed, step_ctr, step_mirror, (int)blockIdx.x);
}

int pack_launch(const PackSeg* d_segs, const SegBlocks& sb, const float* d_params, float* d_packed, int* step_ctr, int* step_mirror,
                hipStream_t st) {
  if (sb.n == 0 || sb.start[sb.n] == 0) return HMP_OK;
  hipLaunchKernelGGL(pack_kernel, dim3(sb.start[sb.n]), dim3(256), 0, st, d_segs, sb, d_params, d_packed, step_ctr, step_mirror);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
}

// ---------------------------------------------------------------------------------------------
// grad reduce: split-K slabs of the packed weight gradients -> flat gradient buffer (fixed slab order).
// grid = (element chunks of 1024, segments).  Block (0,0) additionally sums the per-row {loss, valid} pairs of the
// loss kernel in fixed order into out2 = {loss_sum, count} (the tail of the flat gradient buffer).
// ---------------------------------------------------------------------------------------------
// sum over the split-K slabs of one element, slab 0 first; 8 (clamped) loads in flight per trip instead of one dependent
// load per slab
__device__ __forceinline__ float slab_sum(const float* __restrict__ src, int ns, int64_t stride) {
  float v = 0.f;
  for (int z0 = 0; z0 < ns; z0 += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = src[(int64_t)min(z0 + u, ns - 1) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (z0 + u < ns) v += t[u];
  }
  return v;
}

__global__ __launch_bounds__(256) void grad_reduce_kernel(const GradSeg* __restrict__ segs, const SegBlocks sb, const GradReduceDyn dyn,
                                                          const float* __restrict__ slabs, const float* params,
                                                          float* __restrict__ grads, const float* __restrict__ row_lv, int n_lv_rows,
                                                          float* __restrict__ out2, NetState* state, const AdamFuse adam) {
  int si = 0;
  while (si + 1 < sb.n && (int)blockIdx.x >= sb.start[si + 1]) ++si;
  const GradSeg& S = segs[si];
  const int64_t n_el = (int64_t)S.rows * S.cols;
  if (S.n_terms == 1 && (S.t[0].kind == GT_ATT_DOT || S.t[0].kind == GT_ATT_DOT_T)) {
    // attention-vector gradients are long dot products: one WAVEFRONT per element, lanes over the reduction index
    const GradTerm& T = S.t[0];
    const int64_t r = (int64_t)((int)blockIdx.x - sb.start[si]) * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r < n_el) {
      const int ns = dyn.n_slabs[T.slab_id];
      const int64_t stride = dyn.slab_stride[T.slab_id];
      const float* w = params + T.w + r * T.ldw;
      float part = 0.f;
      if (T.kind == GT_ATT_DOT) {
        const float* src = slabs + T.src + (int64_t)(r / T.C) * T.ld;
        for (int f = lane; f < T.inner; f += 64) part += slab_sum(src + f, ns, stride) * w[f];
      } else {
        const float* base = slabs + T.src + (r / T.C);
        for (int d = lane; d < T.inner; d += 64) part += slab_sum(base + (int64_t)d * T.ld, ns, stride) * w[d];
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
      if (lane == 0) grads[S.dst + r] = part * T.scale;
    }
  } else {
    const int64_t i = (int64_t)((int)blockIdx.x - sb.start[si]) * 256 + threadIdx.x;
    if (i < n_el) {
    const int r = (int)(i / S.cols), c = (int)(i % S.cols);
    // Adam state of this element, requested before the slab sums so that it travels with them (adam.on is kernel-uniform)
    float a_p = 0.f, a_m = 0.f, a_v = 0.f, a_cnt = 1.f;
    int a_t = 1;
    if (adam.on) {
      a_p = adam.p[S.dst + i]; a_m = adam.m[S.dst + i]; a_v = adam.v[S.dst + i];
      a_cnt = *adam.count;
      a_t = *adam.step_dev;
    }
    float total = 0.f;
    for (int ti = 0; ti < S.n_terms; ++ti) {
      const GradTerm& T = S.t[ti];
      const int ns = dyn.n_slabs[T.slab_id];
      const int64_t stride = dyn.slab_stride[T.slab_id];
      const float* base = slabs + T.src;
      float v = 0.f;
      if (T.kind == GT_COPY) {
        v = slab_sum(base + (int64_t)((r / T.C) * T.Cp + (r % T.C)) * T.ld + c, ns, stride);
      } else if (T.kind == GT_ATT_OUTER) {
        v = params[T.att + r] * slab_sum(base + (int64_t)(r / T.C) * T.ld + c, ns, stride);
      } else if (T.kind == GT_ATT_DOT) {
        const float* src = base + (int64_t)(r / T.C) * T.ld;
        const float* w = params + T.w + (int64_t)r * T.ldw;
        for (int f = 0; f < T.inner; ++f) {
          float s = 0.f;
          for (int z = 0; z < ns; ++z) s += src[z * stride + f];
          v += s * w[f];
        }
      } else if (T.kind == GT_ATT_OUTER_T) {
        v = params[T.att + r] * slab_sum(base + (int64_t)c * T.ld + (r / T.C), ns, stride);
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
    if (adam.on) {
      // torch.optim.Adam with coupled L2 on this element (same arithmetic as adam_kernel); only taken when no gradient
      // term of the network reads parameters (SAGE), so updating in place cannot race with another thread's reads
      const int t = a_t;
      const float bc1 = 1.f - powf(adam.b1, (float)t);
      const float bc2 = 1.f - powf(adam.b2, (float)t);
      const float c = a_cnt;
      const float gscale = 1.f / (c > 1.f ? c : 1.f);
      const int64_t e = S.dst + i;
      const float pi = a_p;
      const float gi = total * gscale + adam.wd * pi;
      const float mi = adam.b1 * a_m + (1.f - adam.b1) * gi;
      const float vi = adam.b2 * a_v + (1.f - adam.b2) * gi * gi;
      adam.m[e] = mi;
      adam.v[e] = vi;
      const float denom = sqrtf(vi) / sqrtf(bc2) + adam.eps;
      adam.p[e] = pi - (adam.lr / bc1) * (mi / denom);
    }
    }
  }
  if (row_lv != nullptr && blockIdx.x == 0) {
    __shared__ float sl[256], sv[256];
    float l = 0.f, v = 0.f;
    for (int r = threadIdx.x; r < n_lv_rows; r += 256) { l += row_lv[2 * r]; v += row_lv[2 * r + 1]; }
    sl[threadIdx.x] = l;
    sv[threadIdx.x] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sv[threadIdx.x] += sv[threadIdx.x + o]; }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      out2[0] = sl[0];
      out2[1] = sv[0];
      if (state) { state->loss_sum = sl[0]; state->count = sv[0]; }
    }
  }
}

int grad_reduce_launch(const GradSeg* d_segs, const SegBlocks& sb, const GradReduceDyn& dyn, const float* d_slabs,
                       const float* d_params, float* d_grads, const float* row_lv, int n_lv_rows, float* out2, NetState* state,
                       const AdamFuse& adam, hipStream_t st) {
  if (sb.n == 0 || sb.start[sb.n] == 0) return HMP_OK;
  hipLaunchKernelGGL(grad_reduce_kernel, dim3(sb.start[sb.n]), dim3(256), 0, st, d_segs, sb, dyn, d_slabs, d_params, d_grads, row_lv,
                     n_lv_rows, out2, state, adam);
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

extern "C" int hmp_argmax_rows(const float* d_x, int32_t ldx, int32_t n_rows, int32_t n_cols, int64_t* d_out, void* stream) {
  using namespace hmp;
  HMP_CHECK_ARG(d_x && d_out && n_rows >= 0 && n_cols > 0 && ldx >= n_cols, "hmp_argmax_rows: bad argument");
  if (n_rows == 0) return HMP_OK;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(cdiv(n_rows, 16)), dim3(256), 0, (hipStream_t)stream, d_x, ldx, n_rows, n_cols, d_out);
  HMP_LAUNCH_CHECK();
  return HMP_OK;
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
