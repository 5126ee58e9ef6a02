// Runtime, Coder, dataset and synthetic-data entry points of libgulon_hip.so.
#include "common.hpp"

namespace gulon {

static thread_local std::string g_last_error;

void set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

// ---- synthetic data: counter-based integer hash; the CPU test generator restates it bit for bit ----
__host__ __device__ static inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__device__ static inline float synth_uniform(uint64_t seed, uint64_t stream, uint64_t idx) {
  uint64_t h = mix64(mix64(seed ^ (stream * 0xD1B54A32D192ED03ULL)) + idx);
  return (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f);
}
__device__ static inline float synth_gauss(uint64_t seed, uint64_t stream, uint64_t idx) {
  uint64_t base = mix64(seed ^ (stream * 0xD1B54A32D192ED03ULL));
  float acc = 0.0f;
  for (int t = 0; t < 6; t++) {
    uint64_t h = mix64(base + idx * 6 + (uint64_t)t);
    acc += (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f);
    acc += (float)(uint32_t)((h >> 16) & 0xFFFFFF) * (1.0f / 16777216.0f);
  }
  return acc - 6.0f;
}
// grid-stride: a dispatch cannot carry 2^32 or more work-items, and n * d can (10 M x 1024)
__global__ void synth_fill(float *__restrict__ X, long long total, int d, int kind, uint64_t seed, int ncentres) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
       t += (long long)gridDim.x * blockDim.x) {
  uint64_t idx = (uint64_t)t;
  uint64_t row = idx / (uint64_t)d;
  uint64_t c = idx - row * (uint64_t)d;
  float v;
  if (kind == 0) v = synth_gauss(seed, 1, idx);
  else if (kind == 2) v = synth_uniform(seed, 1, idx);
  else {
    uint64_t ce = mix64(mix64(seed ^ 0x5851F42D4C957F2DULL) + row) % (uint64_t)ncentres;
    uint64_t cidx = ce * (uint64_t)d + c;
    float centre, scale;
    if (kind == 1) {
      centre = synth_uniform(seed, 2, cidx) * 10.0f - 5.0f;
      scale = synth_uniform(seed, 3, cidx) * 0.9f + 0.1f;
    } else {
      centre = synth_uniform(seed, 2, cidx) * 4.0f - 2.0f;
      scale = synth_uniform(seed, 3, cidx) + 0.5f;
    }
    float g = synth_gauss(seed, 1, idx);
    v = centre + g * scale;
  }
  X[t] = v;
  }
}

__global__ void gather_rows(const float *__restrict__ X, int d, const int *__restrict__ rows, long long total,
                            float *__restrict__ out) {
  long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= total) return;
  long long r = t / d;
  int c = (int)(t - r * d);
  out[t] = X[(size_t)rows[r] * d + c];
}

// MathUtils.distanceSq (MathUtils.scala:85-95): dx = y(i) - x(i) with x = X[row], y = query
__global__ void distance_sq_rows(const float *__restrict__ X, int d, const float *__restrict__ Q, int B, int K,
                                 const int *__restrict__ rows, float *__restrict__ out) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B * K) return;
  int q = t / K;
  int r = rows[t];
  float sum = 0.f;
  if (r >= 0) {
    const float *x = X + (size_t)r * d;
    const float *y = Q + (size_t)q * d;
    for (int i = 0; i < d; i++) {
      float dx = y[i] - x[i];
      sum += dx * dx;
    }
  }
  out[t] = sum;
}

}  // namespace gulon

using namespace gulon;

GULON_API const char *gulon_last_error(void) { return g_last_error.c_str(); }
GULON_API int32_t gulon_abi_version(void) { return GULON_ABI_VERSION; }

GULON_API int32_t gulon_device_count(int32_t *out) {
  return guarded([&] {
    GULON_REQUIRE(out != nullptr, "out is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *out = n;
  });
}
GULON_API int32_t gulon_set_device(int32_t device) {
  return guarded([&] { HIP_CHECK(hipSetDevice(device)); });
}
GULON_API int32_t gulon_device_synchronize(void) {
  return guarded([&] { HIP_CHECK(hipDeviceSynchronize()); });
}
GULON_API int32_t gulon_dev_malloc(void **out, size_t bytes) {
  return guarded([&] {
    GULON_REQUIRE(out != nullptr, "out is null");
    *out = nullptr;
    if (bytes) HIP_CHECK(hipMalloc(out, bytes));
  });
}
GULON_API int32_t gulon_dev_free(void *p) {
  return guarded([&] { if (p) HIP_CHECK(hipFree(p)); });
}
GULON_API int32_t gulon_memcpy_h2d(void *dst, const void *src, size_t bytes) {
  return guarded([&] { if (bytes) HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); });
}
GULON_API int32_t gulon_memcpy_d2h(void *dst, const void *src, size_t bytes) {
  return guarded([&] { if (bytes) HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); });
}

// Vectors.subvectors (Vectors.scala:84-104)
GULON_API int32_t gulon_subvectors(int32_t d, int32_t m, int32_t *from, int32_t *until) {
  return guarded([&] {
    GULON_REQUIRE(d >= 1 && m >= 1 && from && until, "bad subvectors arguments d=%d m=%d", d, m);
    std::vector<int> f, u;
    subvectors(d, m, f, u);
    for (int i = 0; i < m; i++) { from[i] = f[i]; until[i] = u[i]; }
  });
}

// ---- Coder (host integer work; Coder.scala) ---------------------------------
static int round_width(int w) {   // Coder.factoryFor (Coder.scala:35-45)
  if (w < 0) return -1;
  if (w == 0) return 0;
  if (w <= 2) return 2;
  if (w <= 4) return 4;
  if (w <= 8) return 8;
  if (w <= 10) return 10;
  if (w <= 12) return 12;
  if (w <= 16) return 16;
  return -1;
}
static int packed_bytes(int width, int length) { int per = 8 / width; return (length + per - 1) / per; }
static int coder_bytes(int width, int length) {
  switch (width) {
    case 0: return 0;
    case 2: case 4: case 8: return packed_bytes(width, length);
    case 10: return length + packed_bytes(2, length);
    case 12: return length + packed_bytes(4, length);
    case 16: return length + packed_bytes(8, length);
    default: return -1;
  }
}
GULON_API int32_t gulon_coder_width(int32_t num_clusters, int32_t *width_out) {
  return guarded([&] {
    GULON_REQUIRE(width_out != nullptr, "width_out is null");
    uint32_t x = (uint32_t)(num_clusters - 1);             // ProductQuantizer.scala:12
    int w = x == 0 ? 0 : 32 - __builtin_clz(x);
    int r = round_width(w);
    *width_out = r;
    GULON_REQUIRE(r >= 0, "too many clusters: %d", num_clusters);   // ProductQuantizer.scala:13-15
  });
}
GULON_API int32_t gulon_coder_bytes(int32_t width, int32_t length, int32_t *bytes_out) {
  return guarded([&] {
    GULON_REQUIRE(bytes_out != nullptr && length >= 0, "bad arguments");
    int b = coder_bytes(width, length);
    GULON_REQUIRE(b >= 0, "unsupported width: %d", width);           // Coder.scala:57
    *bytes_out = b;
  });
}
static void packed_build(int width, uint8_t *code, const int32_t *idx, int n, int offset) {
  for (int i = 0; i < n; i++) {
    if (width == 2) { int id = idx[i] & 0x3; int j = offset + (i >> 2); code[j] = (uint8_t)(code[j] | (id << ((i & 3) * 2))); }
    else if (width == 4) { int id = idx[i] & 0xF; int j = offset + (i >> 1); code[j] = (uint8_t)(code[j] | (id << ((i & 1) * 4))); }
    else code[offset + i] = (uint8_t)idx[i];
  }
}
static int packed_get(int width, const uint8_t *b, int offset, int i) {
  if (width == 2) return (b[offset + (i >> 2)] >> ((i & 3) * 2)) & 0x3;
  if (width == 4) return (b[offset + (i >> 1)] >> ((i & 1) * 4)) & 0xF;
  return b[offset + i];
}
GULON_API int32_t gulon_coder_build(int32_t width, const int32_t *indices, int32_t length, uint8_t *code_out) {
  return guarded([&] {
    int nb = coder_bytes(width, length);
    GULON_REQUIRE(nb >= 0 && length >= 0, "unsupported width: %d", width);
    if (nb) memset(code_out, 0, (size_t)nb);
    if (width == 0) return;
    if (width <= 8) { packed_build(width, code_out, indices, length, 0); return; }
    int lw = width - 8;
    for (int i = 0; i < length; i++) code_out[i] = (uint8_t)((uint32_t)indices[i] >> lw);
    packed_build(lw, code_out, indices, length, length);
  });
}
GULON_API int32_t gulon_coder_unpack(int32_t width, const uint8_t *code, int32_t length, int32_t *indices_out) {
  return guarded([&] {
    GULON_REQUIRE(coder_bytes(width, length) >= 0 && length >= 0, "unsupported width: %d", width);
    for (int i = 0; i < length; i++) {
      if (width == 0) indices_out[i] = 0;
      else if (width <= 8) indices_out[i] = packed_get(width, code, 0, i);
      else { int lw = width - 8; indices_out[i] = ((code[i] & 0xFF) << lw) | (packed_get(lw, code, length, i) & 0xFF); }
    }
  });
}

// ---- dataset -----------------------------------------------------------------
GULON_API int32_t gulon_dataset_create(const float *x_host, int32_t n, int32_t d, gulon_dataset **out) {
  return guarded([&] {
    GULON_REQUIRE(out != nullptr, "out is null");
    *out = nullptr;
    GULON_REQUIRE(n >= 0 && d >= 1 && (x_host != nullptr || n == 0), "bad dataset shape n=%d d=%d", n, d);
    std::unique_ptr<gulon_dataset> ds(new gulon_dataset());
    ds->n = n; ds->d = d;
    ds->x.alloc(std::max<size_t>((size_t)n * d, 1));
    if (n) HIP_CHECK(hipMemcpy(ds->x.p, x_host, sizeof(float) * (size_t)n * d, hipMemcpyHostToDevice));
    *out = ds.release();
  });
}
GULON_API int32_t gulon_dataset_create_synth(int32_t n, int32_t d, int32_t kind, uint64_t seed, int32_t ncentres,
                                             gulon_dataset **out) {
  return guarded([&] {
    GULON_REQUIRE(out != nullptr, "out is null");
    *out = nullptr;
    GULON_REQUIRE(n >= 0 && d >= 1 && kind >= 0 && kind <= 3 && ((kind != 1 && kind != 3) || ncentres >= 1), "bad synth arguments");
    std::unique_ptr<gulon_dataset> ds(new gulon_dataset());
    ds->n = n; ds->d = d;
    long long total = (long long)n * d;
    ds->x.alloc(std::max<size_t>((size_t)total, 1));
    if (total) {
      hipLaunchKernelGGL(synth_fill, dim3((unsigned)std::min<long long>(ceil_div(total, 256), 1 << 20)), dim3(256), 0, 0,
                         ds->x.p, total, d, kind, seed,
                         ncentres);
      HIP_CHECK(hipGetLastError());
      HIP_CHECK(hipDeviceSynchronize());
    }
    *out = ds.release();
  });
}
GULON_API int32_t gulon_dataset_destroy(gulon_dataset *ds) {
  return guarded([&] { delete ds; });
}
GULON_API int32_t gulon_dataset_shape(const gulon_dataset *ds, int32_t *n, int32_t *d) {
  return guarded([&] {
    GULON_REQUIRE(ds != nullptr, "dataset is null");
    if (n) *n = ds->n;
    if (d) *d = ds->d;
  });
}
GULON_API int32_t gulon_dataset_device_ptr(const gulon_dataset *ds, const float **out) {
  return guarded([&] {
    GULON_REQUIRE(ds != nullptr && out != nullptr, "null argument");
    *out = ds->x.p;
  });
}
GULON_API int32_t gulon_dataset_get_rows(const gulon_dataset *ds, const int32_t *rows, int32_t nrows,
                                         float *out_host) {
  return guarded([&] {
    GULON_REQUIRE(ds != nullptr && nrows >= 0, "bad arguments");
    if (nrows == 0) return;
    for (int i = 0; i < nrows; i++)
      GULON_REQUIRE(rows[i] >= 0 && rows[i] < ds->n, "row %d out of range [0,%d)", rows[i], ds->n);
    DevBuf<int> dr; dr.upload(rows, nrows);
    long long total = (long long)nrows * ds->d;
    DevBuf<float> o((size_t)total);
    hipLaunchKernelGGL(gather_rows, dim3(ceil_div(total, 256)), dim3(256), 0, 0, ds->x.p, ds->d, dr.p, total, o.p);
    HIP_CHECK(hipGetLastError());
    o.download(out_host, (size_t)total);
    HIP_CHECK(hipDeviceSynchronize());
  });
}

GULON_API int32_t gulon_distance_sq_rows(const gulon_dataset *ds, const float *queries, int32_t b,
                                         const int32_t *rows, int32_t k_nn, float *out) {
  return guarded([&] {
    GULON_REQUIRE(ds != nullptr && b >= 0 && k_nn >= 0, "bad arguments");
    size_t bk = (size_t)b * k_nn;
    if (bk == 0) return;
    for (size_t i = 0; i < bk; i++) GULON_REQUIRE(rows[i] < ds->n, "row %d out of range", rows[i]);
    DevBuf<float> dq, dout(bk); DevBuf<int> dr;
    dq.upload(queries, (size_t)b * ds->d);
    dr.upload(rows, bk);
    hipLaunchKernelGGL(distance_sq_rows, dim3(ceil_div((long long)bk, 256)), dim3(256), 0, 0, ds->x.p, ds->d, dq.p, b,
                       k_nn, dr.p, dout.p);
    HIP_CHECK(hipGetLastError());
    dout.download(out, bk);
    HIP_CHECK(hipDeviceSynchronize());
  });
}
