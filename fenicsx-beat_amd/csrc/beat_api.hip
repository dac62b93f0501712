// Context, error text, memory helpers and the small field utilities of libbeat_hip.
#include "beat_common.h"

#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

static thread_local std::string g_last_error;

thread_local bool beat_tls_launched = false;

void beat_set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
}

extern "C" int beat_abi_version(void) { return BEAT_ABI_VERSION; }
extern "C" const char* beat_last_error(void) { return g_last_error.c_str(); }

extern "C" int beat_ctx_create(int device, void* hip_stream, beat_ctx** out) {
  BEAT_REQUIRE(out != nullptr, "null output pointer");
  int count = 0;
  BEAT_HIP_CHECK(hipGetDeviceCount(&count));
  BEAT_REQUIRE(device >= 0 && device < count, "device %d out of range (have %d)", device, count);
  BEAT_HIP_CHECK(hipSetDevice(device));
  beat_ctx* ctx = new beat_ctx();
  ctx->device = device;
  ctx->stream = (hipStream_t)hip_stream;
  BEAT_HIP_CHECK(hipMalloc(&ctx->d_partials, sizeof(double) * BEAT_NRED * BEAT_MAX_PARTIALS));
  BEAT_HIP_CHECK(hipMalloc(&ctx->d_small, sizeof(double) * 64));
  BEAT_HIP_CHECK(hipHostMalloc(&ctx->h_pinned, sizeof(double) * 64));
  BEAT_HIP_CHECK(hipMemset(ctx->d_partials, 0, sizeof(double) * BEAT_NRED * BEAT_MAX_PARTIALS));
  *out = ctx;
  return BEAT_OK;
}

extern "C" int beat_ctx_destroy(beat_ctx* ctx) {
  if (ctx == nullptr) return BEAT_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipFree(ctx->d_partials);
  (void)hipFree(ctx->d_small);
  (void)hipHostFree(ctx->h_pinned);
  delete ctx;
  return BEAT_OK;
}

extern "C" int beat_ctx_set_stream(beat_ctx* ctx, void* hip_stream) {
  BEAT_REQUIRE(ctx != nullptr, "null context");
  ctx->stream = (hipStream_t)hip_stream;
  return BEAT_OK;
}

extern "C" int beat_ctx_synchronize(beat_ctx* ctx) {
  BEAT_REQUIRE(ctx != nullptr, "null context");
  BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return BEAT_OK;
}

extern "C" int beat_malloc(beat_ctx* ctx, size_t bytes, void** dev_out) {
  BEAT_REQUIRE(ctx != nullptr && dev_out != nullptr, "null argument");
  BEAT_HIP_CHECK(hipSetDevice(ctx->device));
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
  if (e != hipSuccess) {
    beat_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return BEAT_ENOMEM;
  }
  BEAT_HIP_CHECK(hipMemsetAsync(p, 0, bytes, ctx->stream));
  *dev_out = p;
  return BEAT_OK;
}

extern "C" int beat_free(beat_ctx* ctx, void* dev_ptr) {
  BEAT_REQUIRE(ctx != nullptr, "null context");
  if (dev_ptr) BEAT_HIP_CHECK(hipFree(dev_ptr));
  return BEAT_OK;
}

extern "C" int beat_memcpy_h2d(beat_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes) {
  BEAT_REQUIRE(ctx != nullptr, "null context");
  BEAT_HIP_CHECK(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return BEAT_OK;
}

extern "C" int beat_memcpy_d2h(beat_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes) {
  BEAT_REQUIRE(ctx != nullptr, "null context");
  BEAT_HIP_CHECK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return BEAT_OK;
}

// ---- streaming field utilities ----------------------------------------------------------------
// 16 B per lane where alignment allows (two doubles per thread), grid capped at 2048 blocks.
__global__ __launch_bounds__(BEAT_BLOCK) void copy_kernel(double* __restrict__ dst,
                                                          const double* __restrict__ src, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * BEAT_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BEAT_BLOCK + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
__global__ __launch_bounds__(BEAT_BLOCK) void copy2_kernel(double2* __restrict__ dst,
                                                           const double2* __restrict__ src, int64_t n2) {
  const int64_t stride = (int64_t)gridDim.x * BEAT_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BEAT_BLOCK + threadIdx.x; i < n2; i += stride) dst[i] = src[i];
}
__global__ __launch_bounds__(BEAT_BLOCK) void fill_kernel(double* __restrict__ dst, double v, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * BEAT_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BEAT_BLOCK + threadIdx.x; i < n; i += stride) dst[i] = v;
}
__global__ __launch_bounds__(BEAT_BLOCK) void gather_kernel(double* __restrict__ dst,
                                                            const double* __restrict__ src,
                                                            const int64_t* __restrict__ idx, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * BEAT_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BEAT_BLOCK + threadIdx.x; i < n; i += stride) dst[i] = src[idx[i]];
}
__global__ __launch_bounds__(BEAT_BLOCK) void scatter_kernel(double* __restrict__ dst,
                                                             const double* __restrict__ src,
                                                             const int64_t* __restrict__ idx, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * BEAT_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BEAT_BLOCK + threadIdx.x; i < n; i += stride) dst[idx[i]] = src[i];
}

static inline unsigned stream_grid(int64_t n) {
  int64_t b = (n + BEAT_BLOCK - 1) / BEAT_BLOCK;
  if (b < 1) b = 1;
  if (b > 2048) b = 2048;
  return (unsigned)b;
}

extern "C" int beat_copy(beat_ctx* ctx, double* dev_dst, const double* dev_src, int64_t n) {
  BEAT_REQUIRE(ctx != nullptr && dev_dst != nullptr && dev_src != nullptr && n >= 0, "bad argument");
  if (n == 0 || dev_dst == dev_src) return BEAT_OK;
  const bool aligned = (((uintptr_t)dev_dst | (uintptr_t)dev_src) & 15) == 0 && (n % 2 == 0);
  if (aligned)
    BEAT_KERNEL(copy2_kernel, dim3(stream_grid(n / 2)), dim3(BEAT_BLOCK), 0, ctx->stream,
                       (double2*)dev_dst, (const double2*)dev_src, n / 2);
  else
    BEAT_KERNEL(copy_kernel, dim3(stream_grid(n)), dim3(BEAT_BLOCK), 0, ctx->stream, dev_dst,
                       dev_src, n);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

extern "C" int beat_fill(beat_ctx* ctx, double* dev_dst, double value, int64_t n) {
  BEAT_REQUIRE(ctx != nullptr && dev_dst != nullptr && n >= 0, "bad argument");
  if (n == 0) return BEAT_OK;
  BEAT_KERNEL(fill_kernel, dim3(stream_grid(n)), dim3(BEAT_BLOCK), 0, ctx->stream, dev_dst, value, n);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

extern "C" int beat_gather(beat_ctx* ctx, double* dev_dst, const double* dev_src,
                           const int64_t* dev_idx, int64_t n) {
  BEAT_REQUIRE(ctx != nullptr && dev_dst && dev_src && dev_idx && n >= 0, "bad argument");
  if (n == 0) return BEAT_OK;
  BEAT_KERNEL(gather_kernel, dim3(stream_grid(n)), dim3(BEAT_BLOCK), 0, ctx->stream, dev_dst,
                     dev_src, dev_idx, n);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

extern "C" int beat_scatter(beat_ctx* ctx, double* dev_dst, const double* dev_src,
                            const int64_t* dev_idx, int64_t n) {
  BEAT_REQUIRE(ctx != nullptr && dev_dst && dev_src && dev_idx && n >= 0, "bad argument");
  if (n == 0) return BEAT_OK;
  BEAT_KERNEL(scatter_kernel, dim3(stream_grid(n)), dim3(BEAT_BLOCK), 0, ctx->stream, dev_dst,
                     dev_src, dev_idx, n);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

// ---- point evaluation and min/max -------------------------------------------------------------
struct ProbeBatch {
  int64_t idx[64];
  double w[64];
  int n;  // points in this batch (<= 16)
};

__global__ void probe_kernel(const double* __restrict__ field, ProbeBatch b, double* __restrict__ out) {
  const int k = threadIdx.x;
  if (k >= b.n) return;
  double s = 0.0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const double w = b.w[4 * k + j];
    if (w != 0.0) s = fma(w, field[b.idx[4 * k + j]], s);
  }
  out[k] = s;
}

extern "C" int beat_field_probe(beat_ctx* ctx, const double* dev_field, const int64_t* host_idx,
                                const double* host_w, int npts, double* host_out) {
  BEAT_REQUIRE(ctx != nullptr && dev_field && host_idx && host_w && host_out && npts >= 0, "bad argument");
  // a handful of points (9 in the Niederer demo): one tiny kernel + one D2H copy per 16 points
  for (int base = 0; base < npts; base += 16) {
    ProbeBatch b;
    b.n = npts - base < 16 ? npts - base : 16;
    for (int k = 0; k < 4 * b.n; ++k) {
      b.idx[k] = host_idx[4 * base + k];
      b.w[k] = host_w[4 * base + k];
    }
    BEAT_KERNEL(probe_kernel, dim3(1), dim3(64), 0, ctx->stream, dev_field, b, ctx->d_small);
    BEAT_LAUNCH_CHECK();
    BEAT_HIP_CHECK(hipMemcpyAsync(ctx->h_pinned, ctx->d_small, sizeof(double) * b.n, hipMemcpyDeviceToHost,
                                  ctx->stream));
    BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < b.n; ++k) host_out[base + k] = ctx->h_pinned[k];
  }
  return BEAT_OK;
}

// The same evaluation into a caller-owned device buffer, without any synchronisation: a time loop records one row of
// probe values per step and reads them back in one piece every so often.
extern "C" int beat_field_probe_record(beat_ctx* ctx, const double* dev_field, const int64_t* host_idx,
                                       const double* host_w, int npts, double* dev_out) {
  BEAT_REQUIRE(ctx != nullptr && dev_field && host_idx && host_w && dev_out && npts >= 0, "bad argument");
  for (int base = 0; base < npts; base += 16) {
    ProbeBatch b;
    b.n = npts - base < 16 ? npts - base : 16;
    for (int k = 0; k < 4 * b.n; ++k) {
      b.idx[k] = host_idx[4 * base + k];
      b.w[k] = host_w[4 * base + k];
    }
    BEAT_KERNEL(probe_kernel, dim3(1), dim3(64), 0, ctx->stream, dev_field, b, dev_out + base);
    BEAT_LAUNCH_CHECK();
  }
  return BEAT_OK;
}

__global__ __launch_bounds__(BEAT_BLOCK) void minmax_partial_kernel(const double* __restrict__ x, int64_t n,
                                                                    double* __restrict__ part) {
  __shared__ double smin[4], smax[4];
  double lo = DBL_MAX, hi = -DBL_MAX;
  int bad = 0;  // fmin / fmax drop NaNs (minNum semantics): track them separately, numpy.min/max propagate them
  const int64_t stride = (int64_t)gridDim.x * BEAT_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BEAT_BLOCK + threadIdx.x; i < n; i += stride) {
    const double v = x[i];
    bad |= (v != v);
    lo = fmin(lo, v);
    hi = fmax(hi, v);
  }
  bad = __syncthreads_or(bad);
  lo = beat_wave_min(lo);
  hi = beat_wave_max(hi);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    smin[wave] = lo;
    smax[wave] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double qnan = __longlong_as_double(0x7ff8000000000000LL);
    part[2 * blockIdx.x] = bad ? qnan : fmin(fmin(smin[0], smin[1]), fmin(smin[2], smin[3]));
    part[2 * blockIdx.x + 1] = bad ? qnan : fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
  }
}

namespace {
__global__ __launch_bounds__(BEAT_BLOCK) void interp2_kernel(double* __restrict__ dst, const double* __restrict__ src,
                                                             const int64_t* __restrict__ idx,
                                                             const double* __restrict__ w, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * BEAT_BLOCK;
  for (int64_t j = (int64_t)blockIdx.x * BEAT_BLOCK + threadIdx.x; j < n; j += stride)
    dst[j] = fma(w[2 * j], src[idx[2 * j]], w[2 * j + 1] * src[idx[2 * j + 1]]);
}
}  // namespace

extern "C" int beat_interp2(beat_ctx* ctx, double* dev_dst, const double* dev_src, const int64_t* dev_idx,
                            const double* dev_w, int64_t n) {
  BEAT_REQUIRE(ctx != nullptr && dev_dst && dev_src && dev_idx && dev_w && n >= 0, "bad argument");
  if (n == 0) return BEAT_OK;
  BEAT_KERNEL(interp2_kernel, dim3(stream_grid(n)), dim3(BEAT_BLOCK), 0, ctx->stream, dev_dst, dev_src, dev_idx,
                     dev_w, n);
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

namespace {
__global__ __launch_bounds__(BEAT_BLOCK) void dot_partial_kernel(const double* __restrict__ x,
                                                                 const double* __restrict__ y, int64_t n,
                                                                 double* __restrict__ partials) {
  __shared__ double red[4];
  double s = 0.0;
  const int64_t stride = (int64_t)gridDim.x * BEAT_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BEAT_BLOCK + threadIdx.x; i < n; i += stride) s = fma(x[i], y[i], s);
  s = beat_block_sum(s, red);
  if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
}  // namespace

extern "C" int beat_field_dot(beat_ctx* ctx, const double* dev_x, const double* dev_y, int64_t n, double* host_out) {
  BEAT_REQUIRE(ctx != nullptr && dev_x && dev_y && host_out && n > 0, "bad argument");
  const unsigned grid = stream_grid(n) > 1024 ? 1024 : stream_grid(n);
  BEAT_KERNEL(dot_partial_kernel, dim3(grid), dim3(BEAT_BLOCK), 0, ctx->stream, dev_x, dev_y, n,
                     ctx->d_partials);
  BEAT_LAUNCH_CHECK();
  std::vector<double> h(grid);
  BEAT_HIP_CHECK(hipMemcpyAsync(h.data(), ctx->d_partials, sizeof(double) * grid, hipMemcpyDeviceToHost, ctx->stream));
  BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  double s = 0.0;
  for (unsigned b = 0; b < grid; ++b) s += h[b];  // fixed order: deterministic
  *host_out = s;
  return BEAT_OK;
}

extern "C" int beat_field_minmax(beat_ctx* ctx, const double* dev_field, int64_t n, double* host_min,
                                 double* host_max) {
  BEAT_REQUIRE(ctx != nullptr && dev_field && host_min && host_max && n > 0, "bad argument");
  const unsigned grid = stream_grid(n) > 1024 ? 1024 : stream_grid(n);
  BEAT_KERNEL(minmax_partial_kernel, dim3(grid), dim3(BEAT_BLOCK), 0, ctx->stream, dev_field, n,
                     ctx->d_partials);
  BEAT_LAUNCH_CHECK();
  std::vector<double> h(2 * grid);
  BEAT_HIP_CHECK(hipMemcpyAsync(h.data(), ctx->d_partials, sizeof(double) * 2 * grid,
                                hipMemcpyDeviceToHost, ctx->stream));
  BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  double lo = DBL_MAX, hi = -DBL_MAX;
  bool bad = false;  // a NaN anywhere in the field makes both extrema NaN (as numpy.min / numpy.max)
  for (unsigned b = 0; b < grid; ++b) {
    bad = bad || h[2 * b] != h[2 * b] || h[2 * b + 1] != h[2 * b + 1];
    lo = h[2 * b] < lo ? h[2 * b] : lo;
    hi = h[2 * b + 1] > hi ? h[2 * b + 1] : hi;
  }
  *host_min = bad ? std::nan("") : lo;
  *host_max = bad ? std::nan("") : hi;
  return BEAT_OK;
}
