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
#else
#define KT_DEFINE(tag)
#define KT(i) \
  do {        \
  } while (0)
#endif

// ---- Philox4x32-10 (counter based; the same element always draws the same number, so the
//      backward pass regenerates the forward's keep-mask instead of storing it) ------------------
struct Philox4 {
  uint32_t v[4];
};

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                 uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += W0; k1 += W1;
  }
  Philox4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
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
__device__ inline void drop_keep4(const DropCfg& cfg, uint32_t q, bool keep[4]) {
  Philox4 r = philox4x32_10(q, cfg.stream, cfg.step, 0x48594452u, cfg.k0, cfg.k1);
#pragma unroll
  for (int i = 0; i < 4; ++i) keep[i] = r.v[i] >= cfg.thresh;
}

}  // namespace hmp
