// Streaming probes: what does the memory system of THIS device give a kernel that does nothing but move bytes, in the access
// patterns the hot kernels use?  bench.py's `roofline.inplace_stream` is measured with beat_stream_probe on the very state array
// the ionic kernel runs on (round 4 used torch's x.mul_(1.0) -- a second-hand ceiling), tools/stream_probe.py sweeps every variant
// (profiles/r05_streaming.md).  No reference counterpart: the reference publishes no throughput; this is measurement
// infrastructure beside beat_copy / beat_fill.
//
//   mode   0 in place   x[i] = s * x[i]            (s = 1.0 as a run-time argument: one load + one store per element)
//          1 read only  sum of x                   (per-lane sum, stored only if it compares equal to an impossible value)
//          2 write only x[i] = s
//          3 copy       y[i] = x[i]                (y = dev + n: the second half of the buffer)
//          4 in place over R rows of a (R, ld) array, all R loads of an index issued before the R stores -- the ionic
//            kernels' pattern: R row streams per wavefront (R = rows argument)
//   policy bit 0: non-temporal loads, bit 1: non-temporal stores (__builtin_nontemporal_load / _store: `nt` on gfx950);
//          bit 2: loads and stores through raw-buffer instructions (buffer_load_dwordx4) instead of global_load_dwordx4
//   unroll 1, 2 or 4 independent 16-byte accesses in flight per lane and loop trip
//   blocks workgroups of 256 threads in the launch (grid-stride loop); 0 = one workgroup per 256 * unroll * 16 bytes, no loop
#include "beat_common.h"

namespace {

typedef double v2d __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));

template <int POL>
__device__ __forceinline__ v2d ld16(const v2d* base, int64_t i) {
  if constexpr (POL & 4) {
    // a descriptor covers 4 GiB: rebase it per access block (the offset inside stays below 2^31)
    const int64_t blk = i >> 26;  // 2^26 x 16 B = 1 GiB
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (blk << 26)), 0, 0x7ffffff0, 0x00020000);
    const v4i v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)((i - (blk << 26)) * 16), 0, (POL & 1) ? 2 : 0);
    return __builtin_bit_cast(v2d, v);
  } else if constexpr (POL & 1) {
    return __builtin_nontemporal_load(base + i);
  } else {
    return base[i];
  }
}
template <int POL>
__device__ __forceinline__ void st16(v2d* base, int64_t i, v2d v) {
  if constexpr (POL & 4) {
    const int64_t blk = i >> 26;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (blk << 26)), 0, 0x7ffffff0, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, v), r, (int)((i - (blk << 26)) * 16), 0, (POL & 2) ? 2 : 0);
  } else if constexpr (POL & 2) {
    __builtin_nontemporal_store(v, base + i);
  } else {
    base[i] = v;
  }
}

// n2 = number of 16-byte elements; every lane handles U elements BEAT_BLOCK apart per trip (coalesced per access)
template <int MODE, int POL, int U>
__global__ __launch_bounds__(BEAT_BLOCK) void stream_kernel(v2d* __restrict__ x, int64_t n2, double s, double* __restrict__ sink) {
  const int64_t chunk = (int64_t)BEAT_BLOCK * U;
  const int64_t stride = (int64_t)gridDim.x * chunk;
  v2d acc = {0.0, 0.0};
  for (int64_t base = (int64_t)blockIdx.x * chunk + threadIdx.x; base < n2; base += stride) {
    v2d v[U];
    if constexpr (MODE != 2) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int64_t i = base + (int64_t)u * BEAT_BLOCK;
        v[u] = i < n2 ? ld16<POL>(x, i) : v2d{0.0, 0.0};
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = base + (int64_t)u * BEAT_BLOCK;
      if constexpr (MODE == 0) {
        if (i < n2) st16<POL>(x, i, v[u] * s);
      } else if constexpr (MODE == 1) {
        acc += v[u];
      } else if constexpr (MODE == 2) {
        if (i < n2) st16<POL>(x, i, v2d{s, s});
      } else {
        if (i < n2) st16<POL>(x + n2, i, v[u]);
      }
    }
  }
  if constexpr (MODE == 1) {
    if (acc.x + acc.y == 0.12345678912345e300) sink[0] = acc.x;  // keeps the loads alive, never true
  }
}

// R rows of a (R, ld2) array of 16-byte elements: per index all R loads, then all R stores
template <int R, int POL>
__global__ __launch_bounds__(BEAT_BLOCK) void rows_kernel(v2d* __restrict__ x, int64_t n2, int64_t ld2, double s) {
  const int64_t stride = (int64_t)gridDim.x * BEAT_BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BEAT_BLOCK + threadIdx.x; i < n2; i += stride) {
    v2d v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) v[r] = ld16<POL>(x + r * ld2, i);
#pragma unroll
    for (int r = 0; r < R; ++r) st16<POL>(x + r * ld2, i, v[r] * s);
  }
}

template <int MODE, int POL>
int launch_u(beat_ctx* ctx, int unroll, unsigned grid, v2d* x, int64_t n2, double s) {
  switch (unroll) {
    case 1: BEAT_KERNEL((stream_kernel<MODE, POL, 1>), dim3(grid), dim3(BEAT_BLOCK), 0, ctx->stream, x, n2, s, ctx->d_small); break;
    case 2: BEAT_KERNEL((stream_kernel<MODE, POL, 2>), dim3(grid), dim3(BEAT_BLOCK), 0, ctx->stream, x, n2, s, ctx->d_small); break;
    case 4: BEAT_KERNEL((stream_kernel<MODE, POL, 4>), dim3(grid), dim3(BEAT_BLOCK), 0, ctx->stream, x, n2, s, ctx->d_small); break;
    default: beat_set_error("unroll must be 1, 2 or 4"); return BEAT_EINVAL;
  }
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}
template <int MODE>
int launch_p(beat_ctx* ctx, int policy, int unroll, unsigned grid, v2d* x, int64_t n2, double s) {
  switch (policy) {
    case 0: return launch_u<MODE, 0>(ctx, unroll, grid, x, n2, s);
    case 1: return launch_u<MODE, 1>(ctx, unroll, grid, x, n2, s);
    case 2: return launch_u<MODE, 2>(ctx, unroll, grid, x, n2, s);
    case 3: return launch_u<MODE, 3>(ctx, unroll, grid, x, n2, s);
    case 4: return launch_u<MODE, 4>(ctx, unroll, grid, x, n2, s);
    case 5: return launch_u<MODE, 5>(ctx, unroll, grid, x, n2, s);
    case 6: return launch_u<MODE, 6>(ctx, unroll, grid, x, n2, s);
    case 7: return launch_u<MODE, 7>(ctx, unroll, grid, x, n2, s);
  }
  beat_set_error("policy must be 0..7");
  return BEAT_EINVAL;
}
template <int R>
int launch_rows(beat_ctx* ctx, int policy, unsigned grid, v2d* x, int64_t n2, int64_t ld2, double s) {
  switch (policy) {
    case 0: BEAT_KERNEL((rows_kernel<R, 0>), dim3(grid), dim3(BEAT_BLOCK), 0, ctx->stream, x, n2, ld2, s); break;
    case 1: BEAT_KERNEL((rows_kernel<R, 1>), dim3(grid), dim3(BEAT_BLOCK), 0, ctx->stream, x, n2, ld2, s); break;
    case 2: BEAT_KERNEL((rows_kernel<R, 2>), dim3(grid), dim3(BEAT_BLOCK), 0, ctx->stream, x, n2, ld2, s); break;
    case 3: BEAT_KERNEL((rows_kernel<R, 3>), dim3(grid), dim3(BEAT_BLOCK), 0, ctx->stream, x, n2, ld2, s); break;
    default: beat_set_error("rows mode: policy must be 0..3"); return BEAT_EINVAL;
  }
  BEAT_LAUNCH_CHECK();
  return BEAT_OK;
}

}  // namespace

extern "C" int beat_stream_probe(beat_ctx* ctx, double* dev, int64_t n, int mode, int policy, int unroll, int blocks,
                                 int rows, int64_t ld) {
  BEAT_REQUIRE(ctx != nullptr && dev != nullptr && n > 0, "bad argument");
  BEAT_REQUIRE(((uintptr_t)dev & 15) == 0 && (n & 1) == 0, "16-byte aligned buffer and an even element count expected");
  BEAT_REQUIRE(blocks >= 0 && blocks <= (1 << 22), "blocks out of range");
  const int64_t n2 = n / 2;
  v2d* x = (v2d*)dev;
  if (mode == 4) {
    BEAT_REQUIRE((ld & 1) == 0 && ld >= n, "even row stride >= n expected");
    const unsigned grid = blocks > 0 ? (unsigned)blocks : (unsigned)std::min<int64_t>((n2 + BEAT_BLOCK - 1) / BEAT_BLOCK, 1 << 22);
    switch (rows) {
      case 1: return launch_rows<1>(ctx, policy, grid, x, n2, ld / 2, 1.0);
      case 4: return launch_rows<4>(ctx, policy, grid, x, n2, ld / 2, 1.0);
      case 8: return launch_rows<8>(ctx, policy, grid, x, n2, ld / 2, 1.0);
      case 19: return launch_rows<19>(ctx, policy, grid, x, n2, ld / 2, 1.0);
      case 45: return launch_rows<45>(ctx, policy, grid, x, n2, ld / 2, 1.0);
    }
    beat_set_error("rows must be 1, 4, 8, 19 or 45");
    return BEAT_EINVAL;
  }
  BEAT_REQUIRE(unroll == 1 || unroll == 2 || unroll == 4, "unroll must be 1, 2 or 4");
  const int64_t per_block = (int64_t)BEAT_BLOCK * unroll;
  const unsigned grid = blocks > 0 ? (unsigned)blocks : (unsigned)std::min<int64_t>((n2 + per_block - 1) / per_block, 1 << 22);
  switch (mode) {
    case 0: return launch_p<0>(ctx, policy, unroll, grid, x, n2, 1.0);
    case 1: return launch_p<1>(ctx, policy, unroll, grid, x, n2, 1.0);
    case 2: return launch_p<2>(ctx, policy, unroll, grid, x, n2, 0.0);  // (the caller owns the buffer's content)
    case 3: return launch_p<3>(ctx, policy, unroll, grid, x, n2 / 2, 1.0);  // first half -> second half
  }
  beat_set_error("mode must be 0..4");
  return BEAT_EINVAL;
}
