// Micro-benchmark behind the shape of ipc_xfer_kernel (csrc/beat_dist.hip): a 2 MiB plane copied between two device
// buffers by a kernel that ends with "last workgroup raises a flag", alone and beside a streaming kernel that keeps the
// memory system busy on another stream -- for several grid sizes, loads in flight per thread and fence placements.
//   hipcc -O2 --offload-arch=gfx950 tools/xfer_bench.cpp -o tools/xfer_bench.bin && tools/xfer_bench.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(expr)                                                                                  \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess) {                                                                          \
      std::fprintf(stderr, "%s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      std::exit(2);                                                                                  \
    }                                                                                                \
  } while (0)

// FENCE: 0 = system fence in every workgroup, 1 = agent fence in every workgroup + system fence in the last,
//        2 = no fence in the workgroups (the counter's acq_rel only) + system fence in the last
template <int UNROLL, int FENCE>
__global__ __launch_bounds__(256) void xfer(const double2* __restrict__ src, double2* __restrict__ dst, long pairs,
                                            unsigned* counter, unsigned long long* flag, unsigned long long value) {
  const long stride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < pairs; i += UNROLL * stride) {
    double2 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) dst[i + u * stride] = v[u];
  }
  for (; i < pairs; i += stride) dst[i] = src[i];
  if (FENCE == 0) __threadfence_system();
  if (FENCE == 1) __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (done == gridDim.x - 1) {
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void stream_kernel(double* p, long n, int reps) {
  for (int r = 0; r < reps; ++r)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = p[i] * 1.0000001;
}

template <int UNROLL, int FENCE>
static void run(const char* what, int blocks, bool fine, hipStream_t s, hipStream_t busy, double* big, long nbig) {
  const long plane = 512 * 512, pairs = plane / 2;
  double *src, *dst;
  unsigned* counter;
  unsigned long long* flag;
  CHECK(hipMalloc(&src, plane * 8));
  if (fine)
    CHECK(hipExtMallocWithFlags((void**)&dst, plane * 8 + 64, hipDeviceMallocFinegrained));
  else
    CHECK(hipMalloc(&dst, plane * 8 + 64));
  CHECK(hipMalloc(&counter, 4));
  CHECK(hipMemset(counter, 0, 4));
  flag = (unsigned long long*)(dst + plane);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  float res[2];
  for (int contended = 0; contended < 2; ++contended) {
    const int reps = 40;
    CHECK(hipDeviceSynchronize());
    if (contended) stream_kernel<<<4096, 256, 0, busy>>>(big, nbig, 12);  // ~10 ms of streaming beside the copies
    for (int k = 0; k < 3; ++k) xfer<UNROLL, FENCE><<<blocks, 256, 0, s>>>((const double2*)src, (double2*)dst, pairs, counter, flag, 1);
    CHECK(hipEventRecord(e0, s));
    for (int k = 0; k < reps; ++k)
      xfer<UNROLL, FENCE><<<blocks, 256, 0, s>>>((const double2*)src, (double2*)dst, pairs, counter, flag, k + 2);
    CHECK(hipEventRecord(e1, s));
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventElapsedTime(&res[contended], e0, e1));
    res[contended] *= 1e3f / reps;
  }
  std::printf("%-28s blocks %4d  %s  alone %7.1f us   beside a streaming kernel %7.1f us\n", what, blocks, fine ? "fine  " : "coarse", res[0], res[1]);
  CHECK(hipFree(src));
  CHECK(hipFree(dst));
  CHECK(hipFree(counter));
}

int main() {
  hipStream_t s, busy;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&busy, hipStreamNonBlocking));
  const long nbig = 1L << 28;  // 2 GiB
  double* big;
  CHECK(hipMalloc(&big, nbig * 8));
  CHECK(hipMemset(big, 0, nbig * 8));
  for (int fine = 0; fine < 2; ++fine) {
    for (int blocks : {8, 16, 32, 64, 128, 256}) {
      run<1, 0>("1 in flight, fence sys/WG", blocks, fine, s, busy, big, nbig);
      run<4, 0>("4 in flight, fence sys/WG", blocks, fine, s, busy, big, nbig);
      run<4, 1>("4 in flight, fence agent/WG", blocks, fine, s, busy, big, nbig);
      run<4, 2>("4 in flight, fence last only", blocks, fine, s, busy, big, nbig);
      run<8, 2>("8 in flight, fence last only", blocks, fine, s, busy, big, nbig);
    }
  }
  // the same plane by hipMemcpyAsync
  {
    double *a, *b;
    CHECK(hipMalloc(&a, 512 * 512 * 8));
    CHECK(hipMalloc(&b, 512 * 512 * 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int contended = 0; contended < 2; ++contended) {
      CHECK(hipDeviceSynchronize());
      if (contended) stream_kernel<<<4096, 256, 0, busy>>>(big, nbig, 12);
      CHECK(hipEventRecord(e0, s));
      for (int k = 0; k < 40; ++k) CHECK(hipMemcpyAsync(b, a, 512 * 512 * 8, hipMemcpyDeviceToDevice, s));
      CHECK(hipEventRecord(e1, s));
      CHECK(hipDeviceSynchronize());
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      std::printf("hipMemcpyAsync D2D 2 MiB            %s %7.1f us\n", contended ? "beside a streaming kernel" : "alone", ms * 1e3f / 40);
    }
  }
  return 0;
}
