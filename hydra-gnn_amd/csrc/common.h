// Internal helpers shared by the translation units of libhydra_mp.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/hydra_mp.h"

namespace hmp {

// ---- error plumbing ---------------------------------------------------------------------------
char* err_buf();  // thread-local, 512 bytes (runtime.hip)

#define HMP_FAIL(code, ...)                          \
  do {                                               \
    snprintf(::hmp::err_buf(), 512, __VA_ARGS__);    \
    return (code);                                   \
  } while (0)

#define HMP_CHECK_ARG(cond, ...)                     \
  do {                                               \
    if (!(cond)) HMP_FAIL(HMP_E_ARG, __VA_ARGS__);   \
  } while (0)

#define HMP_HIP(expr)                                                                           \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess)                                                                       \
      HMP_FAIL(HMP_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

#define HMP_LAUNCH_CHECK()                                                                      \
  do {                                                                                          \
    hipError_t _e = hipGetLastError();                                                          \
    if (_e != hipSuccess)                                                                       \
      HMP_FAIL(HMP_E_HIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

#define HMP_TRY(expr)          \
  do {                         \
    int _r = (expr);           \
    if (_r != HMP_OK) return _r; \
  } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int align4(int x) { return (x + 3) & ~3; }
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---- kernel-argument warm-up ---------------------------------------------------------------------------------------
// Kernel arguments are read through the scalar cache one 64-byte line at a time, and the compiler waits for every field where
// it is first used: a workgroup that walks a 500-byte table entry of a by-value argument struct pays ~8 DEPENDENT misses
// (0.28 us each on an idle MI355X, tools/microbench/kernarg_latency.hip) before its first gather load is issued.
// karg_warm requests one word of every line of [off, off + bytes) of the kernel-argument segment back to back and waits
// once: one round trip, after which the field reads hit the scalar cache.  (The loaded words are kept in registers until the
// wait -- the compiler does not know these are loads, so it must not reuse their destinations earlier.)
template <int LINES>
__device__ __forceinline__ void karg_warm(int off, int bytes) {
  const char* kp = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  int t[LINES];
#pragma unroll
  for (int i = 0; i < LINES; ++i) {
    const int o = (off + min(i * 64, bytes - 4)) & ~3;
    asm volatile("s_load_dword %0, %1, %2" : "=s"(t[i]) : "s"(kp), "s"(o));
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < LINES; ++i) asm volatile("" ::"s"(t[i]));
}

// ---- in-kernel phase timestamps (profiling build only: make KTIME=1) -------------------------------------------
// KT(i) stores the 100 MHz wall clock of thread 0 of block 0 into slot i of the translation unit's buffer;
// hmp_debug_ktime_<tag>(out[64]) copies the buffer to the host.  Compiled out of the product library.
#ifdef HMP_KTIME
#define KT_DEFINE(tag)                                                                                   \
  static __device__ unsigned long long kt_buf[64];                                                       \
  static __device__ int kt_sel = 0; /* the workgroup that stamps (hmp_debug_ktime_<tag>_select) */        \
  extern "C" int hmp_debug_ktime_##tag##_select(int blk) {                                               \
    return hipMemcpyToSymbol(HIP_SYMBOL(kt_sel), &blk, sizeof(int)) == hipSuccess ? 0 : 1;               \
  }                                                                                                      \
  extern "C" int hmp_debug_ktime_##tag(unsigned long long* out) {                                        \
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(kt_buf), sizeof(kt_buf)) == hipSuccess ? 0 : 1;           \
  }                                                                                                      \
  extern "C" int hmp_debug_ktime_##tag##_set(const unsigned long long* in) {                             \
    return hipMemcpyToSymbol(HIP_SYMBOL(kt_buf), in, sizeof(kt_buf)) == hipSuccess ? 0 : 1;              \
  }
#define KT(i)                                                                      \
  do {                                                                             \
    if (threadIdx.x == 0 && (int)blockIdx.x == kt_sel) kt_buf[i] = wall_clock64();  \
  } while (0)
// span of a launch over ALL its workgroups: slot i = earliest start, i + 1 = latest end, i + 2 + sub = longest workgroup of kind sub (thread 0 of
// every workgroup; the host presets the slots: hmp_debug_ktime_<tag>_set)
#define KT_SPAN_BEGIN(i)                                                  \
  unsigned long long kt_span_t0 = 0;                                      \
  do {                                                                    \
    if (threadIdx.x == 0) {                                               \
      kt_span_t0 = wall_clock64();                                        \
      atomicMin(&kt_buf[i], kt_span_t0);                                  \
    }                                                                     \
  } while (0)
#define KT_SPAN_END(i, sub)                                               \
  do {                                                                    \
    if (threadIdx.x == 0) {                                               \
      const unsigned long long kt_now = wall_clock64();                   \
      atomicMax(&kt_buf[(i) + 1], kt_now);                                \
      atomicMax(&kt_buf[(i) + 2 + (sub)], kt_now - kt_span_t0);           \
    }                                                                     \
  } while (0)
// per-workgroup life of one launch: (start, end) of workgroup b into a second, larger buffer (first 1024 workgroups)
#define KT_BLOCKS_DEFINE(tag)                                                                            \
  static __device__ unsigned long long kt_blk[2048];                                                     \
  extern "C" int hmp_debug_ktime_##tag##_blocks(unsigned long long* out) {                               \
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(kt_blk), sizeof(kt_blk)) == hipSuccess ? 0 : 1;           \
  }
#define KT_BLOCK_BEGIN()                                                          \
  do {                                                                            \
    if (threadIdx.x == 0 && blockIdx.x < 1024) kt_blk[2 * blockIdx.x] = wall_clock64(); \
  } while (0)
#define KT_BLOCK_END()                                                            \
  do {                                                                            \
    if (threadIdx.x == 0 && blockIdx.x < 1024) kt_blk[2 * blockIdx.x + 1] = wall_clock64(); \
  } while (0)
// stamp once every vector load issued so far has landed
#define KTW(i)                                   \
  do {                                           \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
    KT(i);                                       \
  } while (0)
// accumulating form: kt_buf[i] += now - t0 (t0 from KT_NOW()); thread 0 of block 0 only
#define KT_NOW() wall_clock64()
#define KT_ADD(i, t0)                                                               \
  do {                                                                              \
    if (threadIdx.x == 0 && (int)blockIdx.x == kt_sel) kt_buf[i] += wall_clock64() - (t0); \
  } while (0)
#define KT_ZERO(i)                                                 \
  do {                                                             \
    if (threadIdx.x == 0 && (int)blockIdx.x == kt_sel) kt_buf[i] = 0; \
  } while (0)
#else
#define KT_DEFINE(tag)
#define KT(i) \
  do {        \
  } while (0)
#define KTW(i) \
  do {         \
  } while (0)
#define KT_BLOCKS_DEFINE(tag)
#define KT_BLOCK_BEGIN() \
  do {                   \
  } while (0)
#define KT_BLOCK_END() \
  do {                 \
  } while (0)
#define KT_SPAN_BEGIN(i) \
  do {                   \
  } while (0)
#define KT_SPAN_END(i, sub) \
  do {                      \
  } while (0)
#define KT_NOW() 0ull
#define KT_ADD(i, t0) \
  do {                \
    (void)(t0);       \
  } while (0)
#define KT_ZERO(i) \
  do {             \
  } while (0)
#endif

// ---- dropout RNG: counter based (the same element of the same tensor at the same step always draws the same number, so a
//      mask can be regenerated instead of stored, and tests replay it through hmp_dropout_mask).  Round 1 used Philox4x32-10:
//      40 32x32 -> 64-bit multiplies (quarter rate on the vector ALU) per 4 decisions made the RNG the largest single cost of
//      the 256-wide aggregation epilogue (rocprofv3, config 5).  A dropout mask needs decorrelated bits, not a crypto-grade
//      stream: two rounds of a 32-bit avalanche hash (xorshift-multiply, "lowbias32" constants) give one word = two 16-bit
//      draws; 4 multiplies per 4 decisions.  p is resolved to 2^-16.
__host__ __device__ inline uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// RNG coordinates of one dropout site: (seed, step) make the key, `stream` numbers the tensor
// (layer, node/edge type), the element index is the counter.
struct DropCfg {
  uint32_t k0, k1;      // seed
  uint32_t step;        // draw number of this call ...
  uint32_t stream;      // tensor id
  uint32_t thresh;      // keep iff draw >= thresh   (thresh = p * 2^32)
  float scale;          // 1/(1-p)
  const int* step_dev;  // ... plus *step_dev when set (device step counter: fresh masks under graph replay)
};

// The device step counter through the VECTOR memory path.  A scalar load of it costs a kernel its first memory round trip on its
// own: scalar loads return out of order, so the next s_waitcnt lgkmcnt(0) -- in front of the first kernel-argument field read
// after it -- waits for the counter too, before any gather load is issued.  The vector load is waited for where the value is
// used (the dropout epilogue, behind the gathers).  A null pointer reads 0 through a descriptor of zero records.
__device__ __forceinline__ uint32_t drop_step_vload(const int* step_dev) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(step_dev), 0, step_dev ? 4 : 0, 0x00020000);
  return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, 0, 0, 0);
}

__device__ inline DropCfg drop_resolve(DropCfg c) {
  if (c.step_dev) c.step += (uint32_t)(*c.step_dev);
  c.step_dev = nullptr;
  return c;
}

__host__ __device__ inline uint32_t drop_thresh(float p) {
  double t = (double)p * 4294967296.0;
  if (t < 0) t = 0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)t;
}

// the two hash words of quad q (elements 4*q .. 4*q+3 of the tensor numbered cfg.stream): element j draws the 16-bit half
// (j & 1) of word (j >> 1) and is kept when the draw >= thresh >> 16
__host__ __device__ inline void drop_pair(const DropCfg& cfg, uint32_t q, uint32_t& a, uint32_t& b) {
  // (stream, step, seed) fold into a wave-uniform key; the quad index is the counter
  const uint32_t key = cfg.k0 ^ (cfg.stream * 0x9E3779B1u) ^ (cfg.step * 0x85EBCA77u);
  a = hash32(q ^ key);
  b = hash32((a + 0x9E3779B9u) ^ cfg.k1 ^ (q * 0xC2B2AE3Du));
}
// keep flags for the 4 consecutive elements 4*q .. 4*q+3 of the tensor numbered cfg.stream
__host__ __device__ inline void drop_keep4(const DropCfg& cfg, uint32_t q, bool keep[4]) {
  uint32_t a, b;
  drop_pair(cfg, q, a, b);
  const uint32_t t16 = cfg.thresh >> 16;
  keep[0] = (a & 0xffffu) >= t16;
  keep[1] = (a >> 16) >= t16;
  keep[2] = (b & 0xffffu) >= t16;
  keep[3] = (b >> 16) >= t16;
}

}  // namespace hmp
