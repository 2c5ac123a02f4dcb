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

// ---- in-kernel phase timestamps (profiling build only: make KTIME=1) -------------------------------------------
// KT(i) stores the 100 MHz wall clock of thread 0 of block 0 into slot i of the translation unit's buffer;
// hmp_debug_ktime_<tag>(out[64]) copies the buffer to the host.  Compiled out of the product library.
#ifdef HMP_KTIME
#define KT_DEFINE(tag)                                                                                   \
  static __device__ unsigned long long kt_buf[64];                                                       \
  extern "C" int hmp_debug_ktime_##tag(unsigned long long* out) {                                        \
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(kt_buf), sizeof(kt_buf)) == hipSuccess ? 0 : 1;           \
  }
#define KT(i)                                                                      \
  do {                                                                             \
    if (threadIdx.x == 0 && blockIdx.x == 0) kt_buf[i] = wall_clock64();           \
  } while (0)
// accumulating form: kt_buf[i] += now - t0 (t0 from KT_NOW()); thread 0 of block 0 only
#define KT_NOW() wall_clock64()
#define KT_ADD(i, t0)                                                               \
  do {                                                                              \
    if (threadIdx.x == 0 && blockIdx.x == 0) kt_buf[i] += wall_clock64() - (t0);    \
  } while (0)
#define KT_ZERO(i)                                                 \
  do {                                                             \
    if (threadIdx.x == 0 && blockIdx.x == 0) kt_buf[i] = 0;        \
  } while (0)
#else
#define KT_DEFINE(tag)
#define KT(i) \
  do {        \
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

// keep flags for the 4 consecutive elements 4*q .. 4*q+3 of the tensor numbered cfg.stream
__host__ __device__ inline void drop_keep4(const DropCfg& cfg, uint32_t q, bool keep[4]) {
  // (stream, step, seed) fold into a wave-uniform key; the quad index is the counter
  const uint32_t key = cfg.k0 ^ (cfg.stream * 0x9E3779B1u) ^ (cfg.step * 0x85EBCA77u);
  const uint32_t a = hash32(q ^ key);
  const uint32_t b = hash32((a + 0x9E3779B9u) ^ cfg.k1 ^ (q * 0xC2B2AE3Du));
  const uint32_t t16 = cfg.thresh >> 16;
  keep[0] = (a & 0xffffu) >= t16;
  keep[1] = (a >> 16) >= t16;
  keep[2] = (b & 0xffffu) >= t16;
  keep[3] = (b >> 16) >= t16;
}

}  // namespace hmp
