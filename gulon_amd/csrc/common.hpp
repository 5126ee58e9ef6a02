// Shared host/device helpers for libgulon_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gulon_hip.h"

#define GULON_API extern "C" __attribute__((visibility("default")))

namespace gulon {

void set_error(const char *fmt, ...);

struct DeviceError {
  int32_t code;
};

// Throws after recording the message; caught by the GULON_TRY wrapper.
#define HIP_CHECK(expr)                                                                   \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      ::gulon::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                         __LINE__);                                                       \
      throw ::gulon::DeviceError{_e == hipErrorOutOfMemory ? GULON_ERR_OOM : GULON_ERR_DEVICE}; \
    }                                                                                     \
  } while (0)

#define GULON_REQUIRE(cond, ...)                            \
  do {                                                      \
    if (!(cond)) {                                          \
      ::gulon::set_error(__VA_ARGS__);                      \
      throw ::gulon::DeviceError{GULON_ERR_INVALID_ARGUMENT}; \
    }                                                       \
  } while (0)

#define GULON_UNSUPPORTED(cond, ...)                     \
  do {                                                   \
    if (cond) {                                          \
      ::gulon::set_error(__VA_ARGS__);                   \
      throw ::gulon::DeviceError{GULON_ERR_UNSUPPORTED}; \
    }                                                    \
  } while (0)

template <class F>
static inline int32_t guarded(F &&f) {
  try {
    f();
    return GULON_OK;
  } catch (const DeviceError &e) {
    return e.code;
  } catch (const std::bad_alloc &) {
    set_error("host allocation failed");
    return GULON_ERR_OOM;
  } catch (...) {
    set_error("unknown failure");
    return GULON_ERR_DEVICE;
  }
}

// Owning device buffer.
template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  bool owned = true;   // false: a view of another DevBuf's memory (query contexts share an index's codes)
  DevBuf() = default;
  explicit DevBuf(size_t count) { alloc(count); }
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n), owned(o.owned) { o.p = nullptr; o.n = 0; o.owned = true; }
  DevBuf &operator=(DevBuf &&o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; owned = o.owned; o.p = nullptr; o.n = 0; o.owned = true; }
    return *this;
  }
  void borrow(const DevBuf &o) { release(); p = o.p; n = o.n; owned = false; }
  ~DevBuf() { release(); }
  void alloc(size_t count) {
    release();
    if (count) HIP_CHECK(hipMalloc((void **)&p, count * sizeof(T)));
    n = count;
  }
  void ensure(size_t count) { if (count > n) alloc(count); }
  void release() { if (p && owned) (void)hipFree(p); p = nullptr; n = 0; owned = true; }
  void upload(const T *h, size_t count, hipStream_t st = nullptr) {
    ensure(count);
    if (count) HIP_CHECK(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, st));
  }
  void download(T *h, size_t count, hipStream_t st = nullptr) const {
    if (count) HIP_CHECK(hipMemcpyAsync(h, p, count * sizeof(T), hipMemcpyDeviceToHost, st));
  }
};

// A pointer the device reads out of memory (a descriptor struct, an opaque register copy) is "generic" to the compiler
// and dereferenced with FLAT instructions.  A flat access counts in lgkmcnt as well as vmcnt (it might have hit LDS),
// so every wait for an LDS or scalar result also waits for the global traffic in flight -- phases that should overlap
// run one after the other.  as_global() states the address space: global_load / global_store, vmcnt only.
template <class T> using gptr = T __attribute__((address_space(1))) *;
template <class T> __device__ __forceinline__ gptr<T> as_global(T *p) { return (gptr<T>)p; }
// (HIP's float2 / float4 are classes whose operators expect generic `this`: global pointers use the native vectors)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// Vectors.subvectors (Vectors.scala:84-104)
static inline void subvectors(int d, int m, std::vector<int> &from, std::vector<int> &until) {
  from.resize(m); until.resize(m);
  int ideal = (d + m - 1) / m;
  int shortfall = ideal * m - d;
  int full = m - shortfall;
  for (int i = 0; i < m; i++) {
    if (i < full) { from[i] = i * ideal; until[i] = from[i] + ideal; }
    else { from[i] = full * ideal + (i - full) * (ideal - 1); until[i] = from[i] + ideal - 1; }
  }
}

// java.util.Random (JDK specification) -- host and device.
struct JRandom {
  static constexpr uint64_t MULT = 0x5DEECE66DULL, ADD = 0xBULL, MASK = (1ULL << 48) - 1;
  uint64_t seed;
  __host__ __device__ explicit JRandom(int64_t s) : seed(((uint64_t)s ^ MULT) & MASK) {}
  __host__ __device__ int32_t next(int bits) {
    seed = (seed * MULT + ADD) & MASK;
    return (int32_t)(uint32_t)(seed >> (48 - bits));
  }
  __host__ __device__ bool next_boolean() { return next(1) != 0; }
  __host__ __device__ int32_t next_int(int32_t bound) {
    int32_t r = next(31);
    int32_t m = bound - 1;
    if ((bound & m) == 0) return (int32_t)(((int64_t)bound * (int64_t)r) >> 31);
    int32_t u = r;
    for (;;) {
      r = u % bound;
      int32_t t = (int32_t)((uint32_t)u - (uint32_t)r + (uint32_t)m);
      if (t >= 0) break;
      u = next(31);
    }
    return r;
  }
  // advance the stream by `steps` draws in O(log steps)
  __host__ __device__ void skip(uint64_t steps) {
    uint64_t a = MULT, c = ADD, A = 1, Cc = 0;
    while (steps) {
      if (steps & 1) { A = (A * a) & MASK; Cc = (Cc * a + c) & MASK; }
      c = ((a + 1) * c) & MASK;
      a = (a * a) & MASK;
      steps >>= 1;
    }
    seed = (A * seed + Cc) & MASK;
  }
};

// ---------------------------------------------------------------------------
// Wavefront-distributed sorted list: lane i holds the i-th smallest
// (distance, row) pair; lanes >= keff hold (+inf, INT_MAX).  This is the
// gfx950 replacement for TopKHeap (TopKHeap.scala:3-94): same membership as
// the heap whenever the K-th and (K+1)-th distances differ, deterministic
// (distance, row id) order otherwise -- DESIGN.md "tie rule".
// ---------------------------------------------------------------------------
#ifdef __HIPCC__
// Cross-lane moves that stay off the LDS pipe (the scan saturates it): v_readlane for a
// wave-uniform source lane, DPP wave_shr:1 for the shift-by-one of an insertion.
__device__ inline int readlane_i(int x, int l) { return __builtin_amdgcn_readlane(x, l); }
__device__ inline float readlane_f(float x, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
}
__device__ inline int wave_shr1_i(int x) {   // lane i <- lane i-1 (lane 0 keeps its own value)
  return __builtin_amdgcn_update_dpp(x, x, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ inline float wave_shr1_f(float x) { return __int_as_float(wave_shr1_i(__float_as_int(x))); }

struct WaveList {
  float v;   // this lane's distance
  int i;     // this lane's row id
  float tau; // wave-uniform: distance at lane keff-1 (+inf until full)
  int tau_i; // wave-uniform: row id at lane keff-1
  __device__ void init() { v = INFINITY; i = INT_MAX; tau = INFINITY; tau_i = INT_MAX; }
  // wave-uniform test: would (cv, cr) enter the list?
  __device__ bool accepts(float cv, int cr) const {
    return cv < tau || (cv == tau && cr < tau_i);
  }
  // all 64 lanes must call with wave-uniform (cv, cr)
  __device__ void insert(float cv, int cr, int keff, int lane) {
    bool before = (v < cv) || (v == cv && i < cr);
    int pos = __popcll(__ballot(before));
    float uv = wave_shr1_f(v);
    int ui = wave_shr1_i(i);
    if (lane == pos) { v = cv; i = cr; }
    else if (lane > pos) { v = uv; i = ui; }
    if (lane >= keff) { v = INFINITY; i = INT_MAX; }
    tau = readlane_f(v, keff - 1);
    tau_i = readlane_i(i, keff - 1);
  }
};
#endif

}  // namespace gulon

// Opaque handle types.
struct gulon_dataset {
  gulon::DevBuf<float> x;   // n x d row-major (owned unless borrowed)
  int32_t n = 0, d = 0;
};
