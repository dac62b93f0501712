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

// mode 5: the register-row kernels' ACCESS PATTERN without their arithmetic: a wave owns `segw` consecutive x-nodes of RY rows and
// marches along z, loading RY + 2 rows of the source field per plane (one double per lane, the plane after next in flight while the
// current one is summed) and storing RY rows of the destination.  segw = 62 with shift = -1 is what the kernels do (lanes 0 and 63
// carry the x-halo: every wave's 512-byte row read starts 8 bytes before a multiple of 496); segw = 64 with shift = 0 reads and
// writes whole aligned 512-byte pieces (and would need its x-halo from elsewhere).  What it answers: does the misaligned,
// overlapping segment cost bandwidth by itself?
template <int RY>
__global__ __launch_bounds__(BEAT_BLOCK) void march_kernel(const double* __restrict__ src, double* __restrict__ dst, int nx, int ny, int nz,
                                                           int segw, int shift, int zc, int total_waves, int halo) {
  constexpr int NR = RY + 2;
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= total_waves) return;
  const int nseg = (nx + segw - 1) / segw, nrb = (ny + RY - 1) / RY;
  const int seg = w % nseg, rb = (w / nseg) % nrb, chunk = w / (nseg * nrb);
  const int gx = seg * segw + shift + lane;
  const int cx = min(max(gx, 0), nx - 1);
  const bool out = gx >= 0 && gx < nx && lane + shift >= 0 && lane + shift < segw;
  const int64_t plane = (int64_t)nx * ny;
  int off[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) off[r] = min(max(rb * RY - 1 + r, 0), ny - 1) * nx + cx;
  const int zb = chunk * zc, ze = min(zb + zc, nz);
  double cur[NR], nxt[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) cur[r] = src[(int64_t)zb * plane + off[r]];
  // halo = 1 (with aligned segments): ONE more load per plane in which lanes 0 .. NR-1 fetch the element left of the segment in rows
  // 0 .. NR-1 and lanes 8 .. 8+NR-1 the element right of it; every other lane passes an out-of-range offset (fetches nothing)
  const int hr = lane < 8 ? lane : lane - 8;
  const bool hl = halo && ((lane < NR) || (lane >= 8 && lane < 8 + NR));
  const int hx = lane < 8 ? seg * segw - 1 : seg * segw + segw;
  const unsigned hoff = (hl && hx >= 0 && hx < nx) ? (unsigned)((min(max(rb * RY - 1 + hr, 0), ny - 1) * nx + hx) * 8) : 0x80000000u;
  double hcur = 0.0, hsum = 0.0;
  for (int z = zb; z < ze; ++z) {
    const int zn = min(z + 1, nz - 1);
#pragma unroll
    for (int r = 0; r < NR; ++r) nxt[r] = src[(int64_t)zn * plane + off[r]];
    double hnxt = 0.0;
    if (halo) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (int64_t)zn * plane), 0, (int)(plane * 8), 0x00020000);
      typedef int v2i __attribute__((ext_vector_type(2)));
      const v2i hv = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)hoff, 0, 0);
      hnxt = __hiloint2double(hv.y, hv.x);
    }
    if (out) {
#pragma unroll
      for (int j = 0; j < RY; ++j) {
        const int gy = rb * RY + j;
        if (gy < ny) dst[(int64_t)z * plane + (int64_t)gy * nx + gx] = cur[j] + cur[j + 1] + cur[j + 2] + (lane == 0 ? hcur : 0.0);
      }
    }
    hsum += hcur;
#pragma unroll
    for (int r = 0; r < NR; ++r) cur[r] = nxt[r];
    hcur = hnxt;
  }
  if (hsum == 1.2345e300) dst[0] = hsum;  // (keeps the halo loads alive)
}

int launch_march(beat_ctx* ctx, double* dev, int64_t n, int nx, int ry, int segw, int shift, int blocks, int halo) {
  BEAT_REQUIRE(nx >= 64 && (segw == 62 || segw == 64) && (shift == 0 || shift == -1) && (ry == 2 || ry == 4), "bad march probe arguments");
  const int64_t per_field = n / 2;
  const int nz = (int)(per_field / ((int64_t)nx * nx));
  BEAT_REQUIRE(nz >= 1 && (int64_t)nz * nx * nx == per_field, "the buffer holds two fields of nx x nx x nz doubles");
  const int nseg = (nx + segw - 1) / segw, nrb = (nx + ry - 1) / ry;
  const int64_t per_layer = ((int64_t)nseg * nrb + 3) / 4;
  int nchunks = (int)std::max<int64_t>(1, ((blocks > 0 ? blocks : 4096) + per_layer - 1) / per_layer);
  nchunks = std::min(nchunks, nz);
  const int zc = (nz + nchunks - 1) / nchunks;
  nchunks = (nz + zc - 1) / zc;
  const int total_waves = nseg * nrb * nchunks;
  const unsigned grid = (unsigned)((total_waves + 3) / 4);
  if (ry == 4)
    BEAT_KERNEL((march_kernel<4>), dim3(grid), dim3(BEAT_BLOCK), 0, ctx->stream, (const double*)dev, dev + per_field, nx, nx, nz, segw, shift, zc, total_waves, halo);
  else
    BEAT_KERNEL((march_kernel<2>), dim3(grid), dim3(BEAT_BLOCK), 0, ctx->stream, (const double*)dev, dev + per_field, nx, nx, nz, segw, shift, zc, total_waves, halo);
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
  if (mode == 5)  // the register-row kernels' access pattern: rows = x-nodes per wave (62 | 64), unroll = rows per wave (2 | 4), ld = nx, policy bit 0 = the -1 lane shift, bit 1 = one more load per plane for the x-halo of an aligned segment
    return launch_march(ctx, dev, n, (int)ld, unroll, rows, (policy & 1) ? -1 : 0, blocks, (policy & 2) ? 1 : 0);
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
  beat_set_error("mode must be 0..5");
  return BEAT_EINVAL;
}
