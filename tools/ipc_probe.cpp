// Probe of what the device-to-device ghost-plane transport (csrc/beat_dist.hip, transport "ipc") needs from HIP on this
// pool, with two ordinary processes sharing one GPU: a mailbox in (fine-grained) device memory exported with
// hipIpcGetMemHandle and peer-mapped with hipIpcOpenMemHandle, planes pushed into it with hipMemcpyAsync, and the
// ordering done ON THE DEVICE by sequence flags in that memory: a one-thread kernel stores the flag with a
// system-scope release after the copy, a one-thread kernel of the receiver spins (bounded) on it with acquire loads.
// No host handshake inside the loop -- the two processes run free, so a round that passed proves the flags order it.
//
// (First version of this probe used interprocess events, hipEventInterprocess: fine between two processes, but with
// three ranks the middle one's hipStreamWaitEvent on an opened event failed with "invalid argument", and a passed
// wait proved nothing there because a file handshake had ordered the hosts.)
//
//   hipcc -O2 --offload-arch=gfx950 tools/ipc_probe.cpp -o tools/ipc_probe.bin
//   tools/ipc_probe.bin A /tmp/ipcdir [fine|coarse] &  tools/ipc_probe.bin B /tmp/ipcdir [fine|coarse]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>

static const char* g_role = "?";
#define CHECK(expr)                                                                                                    \
  do {                                                                                                                 \
    hipError_t _e = (expr);                                                                                            \
    if (_e != hipSuccess) {                                                                                            \
      std::fprintf(stderr, "[%s] %s failed: %s (%s:%d)\n", g_role, #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      std::exit(2);                                                                                                    \
    }                                                                                                                  \
  } while (0)

static void put(const std::string& path, const void* data, size_t n) {
  const std::string tmp = path + ".tmp";
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) std::exit(3);
  if (n) std::fwrite(data, 1, n, f);
  std::fclose(f);
  std::rename(tmp.c_str(), path.c_str());
}

static void get(const std::string& path, void* data, size_t n, double timeout_s = 60.0) {
  const auto t0 = std::chrono::steady_clock::now();
  while (true) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (f) {
      const size_t got = n ? std::fread(data, 1, n, f) : 0;
      std::fclose(f);
      if (got == n) return;
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
      std::fprintf(stderr, "[%s] timed out waiting for %s\n", g_role, path.c_str());
      std::exit(4);
    }
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
}

__global__ void fill_kernel(double* p, size_t n, double v) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ void signal_kernel(unsigned long long* flag, unsigned long long value) {
  __threadfence_system();
  __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// spins until *flag >= need or `ticks` of the 100 MHz wall clock have passed (then *err = 1): every wave reaches the end
__global__ void wait_kernel(const unsigned long long* flag, unsigned long long need, long long ticks, int* err) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < need) {
    if (wall_clock64() - t0 > ticks) {
      *err = 1;
      return;
    }
    __builtin_amdgcn_s_sleep(8);
  }
}

__global__ void check_kernel(const double* p, size_t n, double v, int* bad) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n && p[i] != v) atomicAdd(bad, 1);
}

int main(int argc, char** argv) {
  if (argc < 3) return 1;
  g_role = argv[1];
  const bool A = argv[1][0] == 'A';
  const std::string dir = argv[2];
  const bool fine = argc < 4 || std::strcmp(argv[3], "coarse") != 0;
  const size_t plane = 512 * 512;  // doubles: one ghost plane of the 512^3 grid, 2 MiB
  const int slots = 4;
  CHECK(hipSetDevice(0));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  // mailbox: `slots` planes + a page of flags: [0] arrived (written by the peer), [1] freed (written by the peer)
  const size_t bytes = slots * plane * sizeof(double) + 4096;
  char* box = nullptr;
  if (fine)
    CHECK(hipExtMallocWithFlags((void**)&box, bytes, hipDeviceMallocFinegrained));
  else
    CHECK(hipMalloc((void**)&box, bytes));
  CHECK(hipMemsetAsync(box, 0, bytes, s));
  double* src = nullptr;
  double* dst = nullptr;
  int* dev_bad = nullptr;
  int* err = nullptr;
  CHECK(hipMalloc(&src, plane * sizeof(double)));
  CHECK(hipMalloc(&dst, plane * sizeof(double)));
  CHECK(hipMalloc(&dev_bad, sizeof(int)));
  CHECK(hipMemsetAsync(dev_bad, 0, sizeof(int), s));
  CHECK(hipHostMalloc((void**)&err, sizeof(int)));
  *err = 0;
  CHECK(hipStreamSynchronize(s));
  hipIpcMemHandle_t mh, omh;
  CHECK(hipIpcGetMemHandle(&mh, box));
  const std::string me = A ? "a" : "b", other = A ? "b" : "a";
  put(dir + "/" + me + ".mem", &mh, sizeof(mh));
  get(dir + "/" + other + ".mem", &omh, sizeof(omh));
  char* peer = nullptr;
  CHECK(hipIpcOpenMemHandle((void**)&peer, omh, hipIpcMemLazyEnablePeerAccess));
  std::fprintf(stderr, "[%s] %s-grained mailbox exported and the peer's mapped at %p\n", g_role, fine ? "fine" : "coarse", (void*)peer);
  auto data = [&](char* b, int slot) { return (double*)(b + (size_t)slot * plane * sizeof(double)); };
  auto flag = [&](char* b, int k) { return (unsigned long long*)(b + (size_t)slots * plane * sizeof(double)) + k; };
  const long long ticks = 20LL * 100000000LL;  // 20 s

  // both processes send AND receive every round, free-running: round k carries the value 1000 * sender + k
  const int rounds = 400;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const auto t0 = std::chrono::steady_clock::now();
  CHECK(hipEventRecord(e0, s));
  for (int k = 0; k < rounds; ++k) {
    const int slot = k % slots;
    const double mine = (A ? 1000.0 : 2000.0) + k, theirs = (A ? 2000.0 : 1000.0) + k;
    fill_kernel<<<(plane + 255) / 256, 256, 0, s>>>(src, plane, mine);
    // send: wait until the peer has emptied the slot, push the plane, raise the peer's "arrived"
    if (k >= slots) wait_kernel<<<1, 1, 0, s>>>(flag(box, 1), (unsigned long long)(k - slots + 1), ticks, err);
    CHECK(hipMemcpyAsync(data(peer, slot), src, plane * sizeof(double), hipMemcpyDeviceToDevice, s));
    signal_kernel<<<1, 1, 0, s>>>(flag(peer, 0), (unsigned long long)(k + 1));
    // receive: wait for the peer's plane, take it out, tell the peer the slot is free
    wait_kernel<<<1, 1, 0, s>>>(flag(box, 0), (unsigned long long)(k + 1), ticks, err);
    CHECK(hipMemcpyAsync(dst, data(box, slot), plane * sizeof(double), hipMemcpyDeviceToDevice, s));
    signal_kernel<<<1, 1, 0, s>>>(flag(peer, 1), (unsigned long long)(k + 1));
    check_kernel<<<(plane + 255) / 256, 256, 0, s>>>(dst, plane, theirs, dev_bad);
  }
  CHECK(hipEventRecord(e1, s));
  const double enq_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  CHECK(hipStreamSynchronize(s));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  int bad = -1;
  CHECK(hipMemcpy(&bad, dev_bad, sizeof(int), hipMemcpyDeviceToHost));
  std::fprintf(stderr, "[%s] %d free-running rounds (send + receive a 2 MiB plane each): %d wrong values, timeout flag %d; "
               "host enqueue %.1f us per round (8 launches), device %.1f us per round\n", g_role, rounds, bad, *err,
               enq_us / rounds, ms * 1e3 / rounds);
  put(dir + "/" + me + ".done", nullptr, 0);
  get(dir + "/" + other + ".done", nullptr, 0);
  CHECK(hipIpcCloseMemHandle(peer));
  CHECK(hipFree(box));
  const bool ok = bad == 0 && *err == 0;
  std::fprintf(stderr, "[%s] %s\n", g_role, ok ? "ok" : "FAILED");
  return ok ? 0 : 1;
}
