// Slab-decomposed diffusion solve with the communication inside the library: RCCL ghost-plane exchange on a
// side HIP stream overlapped with the interior stencil, RCCL all-reduces of the PCG dot products on the compute
// stream, one C call per solve.  Replaces b.ghostUpdate / KSP.solve / scatter_forward on a partitioned mesh
// (src/beat/base_model.py:203-206,236,242).  See include/beat_hip.h for the contract.
#include "beat_pde_internal.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <mutex>
#include <thread>

namespace {
using namespace beat_pde_detail;

// librccl is opened on first use: the library has no link-time dependency on it (CPU-only hosts can load
// libbeat_hip.so and check its exports; single-GPU users never touch RCCL).
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mutex;

int load_rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.handle != nullptr) return BEAT_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    beat_set_error("cannot open librccl: %s", dlerror());
    return BEAT_EHIP;
  }
#define BEAT_SYM(field, name)                                        \
  do {                                                               \
    *(void**)(&g_rccl.field) = dlsym(h, name);                       \
    if (!g_rccl.field) {                                             \
      beat_set_error("librccl lacks %s", name);                      \
      return BEAT_EHIP;                                              \
    }                                                                \
  } while (0)
  BEAT_SYM(GetUniqueId, "ncclGetUniqueId");
  BEAT_SYM(CommInitRank, "ncclCommInitRank");
  BEAT_SYM(CommDestroy, "ncclCommDestroy");
  BEAT_SYM(CommCount, "ncclCommCount");
  BEAT_SYM(GroupStart, "ncclGroupStart");
  BEAT_SYM(GroupEnd, "ncclGroupEnd");
  BEAT_SYM(Send, "ncclSend");
  BEAT_SYM(Recv, "ncclRecv");
  BEAT_SYM(AllReduce, "ncclAllReduce");
  BEAT_SYM(GetErrorString, "ncclGetErrorString");
#undef BEAT_SYM
  g_rccl.handle = h;
  return BEAT_OK;
}

#define BEAT_RCCL_CHECK(expr)                                                                          \
  do {                                                                                                 \
    ncclResult_t _r = (expr);                                                                          \
    if (_r != ncclSuccess) {                                                                           \
      beat_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__);   \
      return BEAT_EHIP;                                                                                \
    }                                                                                                  \
  } while (0)
}  // namespace

namespace {
// ---- "ipc" transport: ghost planes as device-to-device copies between processes, no RCCL ------------------------
// Every rank owns a mailbox in fine-grained device memory (exported with hipIpcGetMemHandle, peer-mapped by its two
// neighbours): per direction IPC_SLOTS slots of IPC_FIELDS planes, and four sequence flags the neighbours write.  One
// exchange (number s, the same on every rank: all ranks issue their exchanges in the same order) is two launches per
// neighbour on the side stream, ordered ON THE DEVICE by those flags -- the hosts never wait for each other:
//   send kernel:     [spin until the neighbour has emptied slot s % IPC_SLOTS: my freed[d] > s - IPC_SLOTS]  copy the
//                    boundary plane(s) into the neighbour's mailbox; the last workgroup raises the neighbour's
//                    arrived[1 - d] to s + 1 (system-scope release)
//   receive kernel:  spin until my arrived[d] > s (acquire), copy mailbox -> ghost plane(s); the last workgroup
//                    raises the neighbour's freed[1 - d] to s + 1
// Every spin is bounded (BEAT_IPC_TIMEOUT_S, default 30 s on the GPU's 100 MHz wall clock): a wave that gives up
// raises an error word in pinned host memory, which the solve reports after its next synchronisation -- no kernel
// waits forever for a peer that died.  This is what RCCL's point-to-point does over xGMI, minus its channels and
// proxy threads, and three processes sharing ONE GPU can run it -- which RCCL (one rank per device) cannot.
// (A first version ordered plain hipMemcpyAsync copies with interprocess events, hipEventInterprocess: fine between
// two processes, "invalid argument" from hipStreamWaitEvent on the middle one of three; tools/ipc_probe.cpp.)
constexpr int IPC_SLOTS = 4;   // exchanges in flight per direction before a mailbox slot is reused
constexpr int IPC_FIELDS = 2;  // fields per exchange (v_ and the guess increment travel together)
constexpr uint32_t IPC_MAGIC = 0xbea71bc1u;
constexpr int IPC_BLOCKS = 64;  // workgroups per direction of a transfer kernel (tools/xfer_bench.cpp: 32-64 is the sweet spot)

struct IpcFlags {  // in the mailbox, written by the neighbours (peer_lo's side: [0], peer_hi's: [1])
  unsigned long long arrived[2];  // messages from that neighbour that have landed in this rank's mailbox
  unsigned long long freed[2];    // messages this rank sent to that neighbour that it has taken out of its mailbox
};

// All-reduce of the PCG's 1-3 dot products without RCCL (connect_all): every rank stores its values into EVERY rank's
// mailbox (slot s % IPC_AR_SLOTS of all-reduce number s, its own column) and raises that column's sequence flag; every
// rank then waits for all columns of its own mailbox and adds them IN RANK ORDER -- one small kernel on the compute
// stream, the same bits on every rank (the convergence latch must come out alike everywhere), a store and a flag over
// xGMI instead of an RCCL kernel launch with its channel set-up per call.  A rank can be at most one all-reduce ahead of
// the slowest (it needs everybody's contribution to finish one), so four slots never collide.
constexpr int IPC_AR_SLOTS = 4;
constexpr int IPC_AR_VALUES = 4;
struct IpcAr {
  unsigned long long flag[IPC_AR_SLOTS][BEAT_IPC_MAX_RANKS];
  double val[IPC_AR_SLOTS][BEAT_IPC_MAX_RANKS][IPC_AR_VALUES];
};

struct IpcArArgs {
  char* box[BEAT_IPC_MAX_RANKS];  // every rank's mailbox as this rank sees it
  size_t ar_offset;               // of the IpcAr block inside a mailbox
  int rank, world, count;
  unsigned long long seq;
  double* values;                 // in: this rank's partial sums, out: the sums over the ranks
  long long ticks;
  int* err;
};

__global__ __launch_bounds__(64) void ipc_allreduce_kernel(IpcArArgs a) {
  const int j = threadIdx.x;
  const int slot = (int)(a.seq % IPC_AR_SLOTS);
  IpcAr* mine = (IpcAr*)(a.box[a.rank] + a.ar_offset);
  if (j < a.world) {
    IpcAr* dst = (IpcAr*)(a.box[j] + a.ar_offset);
    for (int c = 0; c < a.count; ++c) dst->val[slot][a.rank][c] = a.values[c];
    __threadfence_system();
    __hip_atomic_store(&dst->flag[slot][a.rank], a.seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(&mine->flag[slot][j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.seq + 1) {
      if (wall_clock64() - t0 > a.ticks) {
        __hip_atomic_store(a.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  if (j < a.count) {  // one lane per value, ranks added in order
    double s = 0.0;
    for (int r = 0; r < a.world; ++r) s += __hip_atomic_load(&mine->val[slot][r][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    a.values[j] = s;
  }
}

struct IpcHandle {  // what beat_comm_create_ipc exports (BEAT_IPC_HANDLE_BYTES)
  uint32_t magic, rank;
  int64_t plane_max;
  hipIpcMemHandle_t inbox;
};
static_assert(sizeof(IpcHandle) <= BEAT_IPC_HANDLE_BYTES, "ipc handle size");

struct IpcPeer {  // a neighbour as this rank sees it
  char* box = nullptr;  // its mailbox, peer-mapped (the local one when the neighbour is this rank itself)
  bool connected = false, self = false;
};

struct IpcXfer {  // one direction of one phase (send or receive) of an exchange
  const double* src[IPC_FIELDS];
  double* dst[IPC_FIELDS];
  const unsigned long long* wait_flag;  // nullptr: nothing to wait for
  unsigned long long wait_need;
  unsigned long long* signal_flag;      // nullptr: this direction has no neighbour (its workgroups exit)
  unsigned long long signal_value;
  unsigned int* counter;  // local, zero between launches: workgroups that have finished their share
};
struct IpcXferArgs {
  IpcXfer dir[2];  // blockIdx.y: towards / from peer_lo, peer_hi
  int nf;
  int64_t plane;
  long long ticks;
  int* err;
};

// One phase of one exchange, both directions in one launch (blockIdx.y): wait for the flag, copy nf planes, the last
// workgroup of the direction to finish raises the other side's flag.  Every workgroup polls the flag itself (thread
// 0; no workgroup depends on another one being scheduled).
__global__ __launch_bounds__(256) void ipc_xfer_kernel(IpcXferArgs args) {
  const IpcXfer& a = args.dir[blockIdx.y];
  if (a.signal_flag == nullptr) return;
  if (a.wait_flag != nullptr) {
    if (threadIdx.x == 0) {
      const long long t0 = wall_clock64();
      while (__hip_atomic_load(a.wait_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.wait_need) {
        if (wall_clock64() - t0 > args.ticks) {  // the peer is gone: say so and carry on (the data are wrong, the host will know)
          __hip_atomic_store(args.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    __syncthreads();
  }
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, first = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = 0; k < args.nf; ++k) {
    const double* __restrict__ src = a.src[k];
    double* __restrict__ dst = a.dst[k];
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {  // 16 bytes per lane where the planes allow it, 4 loads in flight
      const int64_t pairs = args.plane >> 1;
      const double2* __restrict__ s2 = (const double2*)src;
      double2* __restrict__ d2 = (double2*)dst;
      int64_t i = first;
      for (; i + 3 * stride < pairs; i += 4 * stride) {
        const double2 v0 = s2[i], v1 = s2[i + stride], v2 = s2[i + 2 * stride], v3 = s2[i + 3 * stride];
        d2[i] = v0;
        d2[i + stride] = v1;
        d2[i + 2 * stride] = v2;
        d2[i + 3 * stride] = v3;
      }
      for (; i < pairs; i += stride) d2[i] = s2[i];
      if ((args.plane & 1) && first == 0) dst[args.plane - 1] = src[args.plane - 1];
    } else {
      for (int64_t i = first; i < args.plane; i += stride) dst[i] = src[i];
    }
  }
  // Every wave waits for its own stores to be acknowledged, the barrier collects the workgroup, and ONE release per
  // workgroup (the counter's, agent scope) publishes them; the last workgroup adds the system-scope release in front
  // of the flag.  (A fence in every wave is what made the first version slow: 256 workgroups fencing at agent or
  // system scope take 26 us for a 2 MiB plane, this form 5-8 us, alone or beside a streaming kernel -- tools/xfer_bench.cpp.)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int done = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (done == gridDim.x - 1) {
      __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __threadfence_system();
      __hip_atomic_store(a.signal_flag, a.signal_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

inline size_t ipc_data_bytes(int64_t plane_max) {
  return ((sizeof(double) * 2 * IPC_SLOTS * IPC_FIELDS * (size_t)plane_max + 255) / 256) * 256;
}
inline double* ipc_slot(char* box, int64_t plane_max, int dir, int slot, int field) {
  return (double*)box + (((int64_t)dir * IPC_SLOTS + slot) * IPC_FIELDS + field) * plane_max;
}
inline IpcFlags* ipc_flags(char* box, int64_t plane_max) { return (IpcFlags*)(box + ipc_data_bytes(plane_max)); }
inline size_t ipc_ar_offset(int64_t plane_max) { return ipc_data_bytes(plane_max) + 256; }
inline size_t ipc_box_bytes(int64_t plane_max) { return ipc_ar_offset(plane_max) + sizeof(IpcAr); }

struct ProfSpan {
  hipEvent_t a, b;
  int kind;  // 0: ghost-plane transfer (side stream), 1: all-reduce (compute stream), 2: compute stream stalled on the ghost planes
};

}  // namespace

struct beat_comm {
  beat_ctx* ctx = nullptr;
  int rank = 0, world = 1, peer_lo = -1, peer_hi = -1;
  bool rccl = false;    // RCCL send/recv for the ghost planes
  bool serial = false;  // ... on the compute stream, on the all-reduce communicator (one stream, one communicator)
  bool ipc = false;     // ghost planes by interprocess device copies
  ncclComm_t p2p = nullptr, coll = nullptr;
  hipStream_t side = nullptr;          // ghost-plane traffic (non-blocking stream owned by the communicator)
  hipEvent_t ev_ready = nullptr;       // compute -> side: the planes to send are final
  hipEvent_t ev_halo = nullptr;        // side -> compute: the ghost planes have arrived
  beat_halo_fn halo = nullptr;
  beat_allreduce_fn allreduce = nullptr;
  void* user = nullptr;
  // ipc transport
  int64_t ipc_plane_max = 0;
  char* ipc_box = nullptr;             // this rank's mailbox: [2 directions][IPC_SLOTS][IPC_FIELDS][plane_max] doubles, then IpcFlags
  unsigned int* ipc_counters = nullptr;  // device, local: [send lo, send hi, recv lo, recv hi]
  int* ipc_err = nullptr;              // pinned host memory: set by a transfer kernel that gave up waiting
  IpcPeer ipc_peer[2];
  char* ipc_all[BEAT_IPC_MAX_RANKS] = {};  // every rank's mailbox (connect_all): the all-reduce goes through them
  bool ipc_all_connected = false;
  uint64_t ipc_ar_seq = 0;
  uint64_t ipc_seq = 0;
  long long ipc_ticks = 30LL * 100000000LL;
  bool ipc_local = false;  // connected with beat_comm_ipc_connect_local: the other mailboxes are this process's own
  long long merged_solves = 0;  // solves that ran the single-reduction iteration (BEAT_DIST_MERGED=1)
  // profiling (beat_comm_profile)
  bool profiling = false;
  std::vector<ProfSpan> spans;
};

namespace {
void prof_clear(beat_comm* c) {
  for (ProfSpan& s : c->spans) {
    (void)hipEventDestroy(s.a);
    (void)hipEventDestroy(s.b);
  }
  c->spans.clear();
}

// open a span on `stream`; returns its index or -1 (profiling off / no events to be had: profiling never fails a solve)
int prof_begin(beat_comm* c, int kind, hipStream_t stream) {
  if (!c->profiling || c->spans.size() >= 200000) return -1;
  ProfSpan s{nullptr, nullptr, kind};
  if (hipEventCreate(&s.a) != hipSuccess) return -1;
  if (hipEventCreate(&s.b) != hipSuccess) {
    (void)hipEventDestroy(s.a);
    return -1;
  }
  (void)hipEventRecord(s.a, stream);
  c->spans.push_back(s);
  return (int)c->spans.size() - 1;
}
void prof_end(beat_comm* c, int idx, hipStream_t stream) {
  if (idx >= 0) (void)hipEventRecord(c->spans[idx].b, stream);
}

void ipc_disconnect(beat_comm* c, IpcPeer& P) {
  if (P.connected && !P.self && P.box) (void)hipIpcCloseMemHandle(P.box);
  P = IpcPeer();
}

// releases whatever a (possibly half-built) communicator holds
void comm_free(beat_comm* c) {
  if (c == nullptr) return;
  if (c->ctx) (void)hipSetDevice(c->ctx->device);
  if (c->side) (void)hipStreamSynchronize(c->side);
  if (c->ctx && (c->rccl || c->ipc || c->coll)) (void)hipStreamSynchronize(c->ctx->stream);
  prof_clear(c);
  if (c->p2p) (void)g_rccl.CommDestroy(c->p2p);
  if (c->coll) (void)g_rccl.CommDestroy(c->coll);
  if (c->ipc) {
    if (c->ipc_all_connected) {  // the neighbours' mappings are among these: closed once, here
      for (int r = 0; r < c->world && r < BEAT_IPC_MAX_RANKS; ++r)
        if (r != c->rank && c->ipc_all[r] && !c->ipc_local) (void)hipIpcCloseMemHandle(c->ipc_all[r]);
      c->ipc_peer[0] = IpcPeer();
      c->ipc_peer[1] = IpcPeer();
    }
    ipc_disconnect(c, c->ipc_peer[0]);
    ipc_disconnect(c, c->ipc_peer[1]);
    if (c->ipc_box) (void)hipFree(c->ipc_box);
    if (c->ipc_counters) (void)hipFree(c->ipc_counters);
    if (c->ipc_err) (void)hipHostFree(c->ipc_err);
  }
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  if (c->ev_halo) (void)hipEventDestroy(c->ev_halo);
  if (c->side) (void)hipStreamDestroy(c->side);
  (void)hipGetLastError();  // whatever the teardown left behind (RCCL's included) is not the next launch's business
  delete c;
}

struct CommGuard {  // destroys a half-built communicator on every early return of a create function
  beat_comm* c;
  ~CommGuard() { comm_free(c); }
  beat_comm* release() {
    beat_comm* r = c;
    c = nullptr;
    return r;
  }
};

int side_stream_and_events(beat_comm* c) {
  // BEAT_DIST_SIDE_PRIORITY=1: a high-priority stream (hipStreamCreateWithPriority).  Measured on the 512 x 512 x 64
  // slab with a self-neighbour, same box: every cross-stream dependency then costs 150-300 us instead of ~10 (an RCCL
  // exchange alone 178 us against 12 on the compute stream; 3.5 against 1.85 ms per solve) -- not the default.
  const char* pe = std::getenv("BEAT_DIST_SIDE_PRIORITY");
  if (pe && pe[0] == '1') {
    int least = 0, greatest = 0;
    BEAT_HIP_CHECK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    BEAT_HIP_CHECK(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, greatest));
  } else {
    BEAT_HIP_CHECK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
  }
  BEAT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
  BEAT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming));
  return BEAT_OK;
}
}  // namespace

extern "C" int beat_comm_unique_id(void* host_id_out) {
  BEAT_REQUIRE(host_id_out != nullptr, "null argument");
  static_assert(sizeof(ncclUniqueId) == BEAT_UNIQUE_ID_BYTES, "unique id size");
  if (int rc = load_rccl()) return rc;
  ncclUniqueId* ids = (ncclUniqueId*)host_id_out;
  BEAT_RCCL_CHECK(g_rccl.GetUniqueId(&ids[0]));
  BEAT_RCCL_CHECK(g_rccl.GetUniqueId(&ids[1]));
  return BEAT_OK;
}

static int check_peers(int rank, int world, int peer_lo, int peer_hi) {
  BEAT_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank %d of %d", rank, world);
  BEAT_REQUIRE(peer_lo >= -1 && peer_lo < world && peer_hi >= -1 && peer_hi < world, "peers (%d, %d) out of range",
               peer_lo, peer_hi);
  return BEAT_OK;
}

static beat_comm* new_comm(beat_ctx* ctx, int rank, int world, int peer_lo, int peer_hi) {
  beat_comm* c = new beat_comm();
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  c->peer_lo = peer_lo;
  c->peer_hi = peer_hi;
  return c;
}

extern "C" int beat_comm_create_rccl_ex(beat_ctx* ctx, int rank, int world, int peer_lo, int peer_hi,
                                        const void* host_id, int flags, beat_comm** out) {
  BEAT_REQUIRE(ctx != nullptr && host_id != nullptr && out != nullptr, "null argument");
  BEAT_REQUIRE((flags & ~BEAT_COMM_SERIAL) == 0, "unknown flags %d", flags);
  if (int rc = check_peers(rank, world, peer_lo, peer_hi)) return rc;
  if (int rc = load_rccl()) return rc;
  BEAT_HIP_CHECK(hipSetDevice(ctx->device));
  CommGuard g{new_comm(ctx, rank, world, peer_lo, peer_hi)};
  beat_comm* c = g.c;
  c->rccl = true;
  c->serial = (flags & BEAT_COMM_SERIAL) != 0;
  const ncclUniqueId* ids = (const ncclUniqueId*)host_id;
  // serial: ONE communicator carries the ghost planes and the all-reduces, all on the compute stream -- every rank
  // enqueues the same operations in the same order on one stream, the ordering RCCL guarantees free of deadlock
  if (!c->serial) BEAT_RCCL_CHECK(g_rccl.CommInitRank(&c->p2p, world, ids[0], rank));
  BEAT_RCCL_CHECK(g_rccl.CommInitRank(&c->coll, world, ids[1], rank));
  if (int rc = side_stream_and_events(c)) return rc;
  *out = g.release();
  return BEAT_OK;
}

extern "C" int beat_comm_create_rccl(beat_ctx* ctx, int rank, int world, int peer_lo, int peer_hi, const void* host_id,
                                     beat_comm** out) {
  return beat_comm_create_rccl_ex(ctx, rank, world, peer_lo, peer_hi, host_id, 0, out);
}

extern "C" int beat_comm_create_callbacks(beat_ctx* ctx, int rank, int world, int peer_lo, int peer_hi,
                                          beat_halo_fn halo, beat_allreduce_fn allreduce, void* user, beat_comm** out) {
  BEAT_REQUIRE(ctx != nullptr && halo != nullptr && allreduce != nullptr && out != nullptr, "null argument");
  if (int rc = check_peers(rank, world, peer_lo, peer_hi)) return rc;
  beat_comm* c = new_comm(ctx, rank, world, peer_lo, peer_hi);
  c->halo = halo;
  c->allreduce = allreduce;
  c->user = user;
  *out = c;
  return BEAT_OK;
}

extern "C" int beat_comm_create_ipc(beat_ctx* ctx, int rank, int world, int peer_lo, int peer_hi, int64_t max_plane_doubles,
                                    const void* host_rccl_id, beat_allreduce_fn allreduce, void* user, void* host_handle_out,
                                    beat_comm** out) {
  BEAT_REQUIRE(ctx != nullptr && out != nullptr && host_handle_out != nullptr, "null argument");
  BEAT_REQUIRE(max_plane_doubles > 0, "max_plane_doubles must be positive");
  BEAT_REQUIRE(!(host_rccl_id != nullptr && allreduce != nullptr),
               "at most one of host_rccl_id (all-reduces by RCCL) and allreduce (by the caller); neither: by the mailboxes (beat_comm_ipc_connect_all)");
  BEAT_REQUIRE(world <= BEAT_IPC_MAX_RANKS || host_rccl_id != nullptr || allreduce != nullptr, "the ipc all-reduce takes at most %d ranks",
               BEAT_IPC_MAX_RANKS);
  if (int rc = check_peers(rank, world, peer_lo, peer_hi)) return rc;
  BEAT_REQUIRE(!(peer_lo >= 0 && peer_lo == peer_hi && peer_lo != rank),
               "both neighbours are rank %d: the ipc transport opens a neighbour's handle once", peer_lo);
  BEAT_HIP_CHECK(hipSetDevice(ctx->device));
  CommGuard g{new_comm(ctx, rank, world, peer_lo, peer_hi)};
  beat_comm* c = g.c;
  c->ipc = true;
  c->ipc_plane_max = max_plane_doubles;
  if (const char* e = std::getenv("BEAT_IPC_TIMEOUT_S")) c->ipc_ticks = (long long)(std::max(0.1, std::atof(e)) * 1.0e8);
  if (host_rccl_id) {
    if (int rc = load_rccl()) return rc;
    const ncclUniqueId* ids = (const ncclUniqueId*)host_rccl_id;
    BEAT_RCCL_CHECK(g_rccl.CommInitRank(&c->coll, world, ids[1], rank));
  } else {
    c->allreduce = allreduce;
    c->user = user;
  }
  if (int rc = side_stream_and_events(c)) return rc;
  static_assert(sizeof(IpcFlags) <= 256, "flags fit their slot");
  const size_t bytes = ipc_box_bytes(max_plane_doubles);
  // fine-grained: what a neighbouring GPU writes over xGMI must be seen by this GPU's caches (coarse-grained memory is
  // only coherent at kernel boundaries of the writing device).  BEAT_IPC_COARSE=1: plain hipMalloc (A/B on one GPU)
  const char* coarse = std::getenv("BEAT_IPC_COARSE");
  if (coarse && coarse[0] == '1')
    BEAT_HIP_CHECK(hipMalloc((void**)&c->ipc_box, bytes));
  else
    BEAT_HIP_CHECK(hipExtMallocWithFlags((void**)&c->ipc_box, bytes, hipDeviceMallocFinegrained));
  BEAT_HIP_CHECK(hipMemsetAsync(c->ipc_box, 0, bytes, c->side));
  BEAT_HIP_CHECK(hipMalloc((void**)&c->ipc_counters, 4 * sizeof(unsigned int)));
  BEAT_HIP_CHECK(hipMemsetAsync(c->ipc_counters, 0, 4 * sizeof(unsigned int), c->side));
  BEAT_HIP_CHECK(hipHostMalloc((void**)&c->ipc_err, sizeof(int)));
  *c->ipc_err = 0;
  BEAT_HIP_CHECK(hipStreamSynchronize(c->side));
  IpcHandle h;
  std::memset(&h, 0, sizeof(h));
  h.magic = IPC_MAGIC;
  h.rank = (uint32_t)rank;
  h.plane_max = max_plane_doubles;
  BEAT_HIP_CHECK(hipIpcGetMemHandle(&h.inbox, c->ipc_box));
  std::memset(host_handle_out, 0, BEAT_IPC_HANDLE_BYTES);
  std::memcpy(host_handle_out, &h, sizeof(h));
  *out = g.release();
  return BEAT_OK;
}

static int ipc_open_peer(beat_comm* c, IpcPeer& P, int peer, const void* host_handle) {
  if (peer < 0) return BEAT_OK;
  if (peer == c->rank) {  // a rank that is its own neighbour (periodic one-rank tests): no handle to open
    P.self = true;
    P.box = c->ipc_box;
    P.connected = true;
    return BEAT_OK;
  }
  BEAT_REQUIRE(host_handle != nullptr, "no handle given for neighbour %d", peer);
  IpcHandle h;
  std::memcpy(&h, host_handle, sizeof(h));
  BEAT_REQUIRE(h.magic == IPC_MAGIC && (int)h.rank == peer, "handle is not rank %d's ipc handle", peer);
  BEAT_REQUIRE(h.plane_max == c->ipc_plane_max, "neighbour %d sized its mailbox for planes of %lld doubles, this rank for %lld",
               peer, (long long)h.plane_max, (long long)c->ipc_plane_max);
  BEAT_HIP_CHECK(hipIpcOpenMemHandle((void**)&P.box, h.inbox, hipIpcMemLazyEnablePeerAccess));
  P.connected = true;
  return BEAT_OK;
}

extern "C" int beat_comm_ipc_connect(beat_comm* c, const void* host_handle_lo, const void* host_handle_hi) {
  BEAT_REQUIRE(c != nullptr && c->ipc, "not an ipc communicator");
  BEAT_REQUIRE(!c->ipc_peer[0].connected && !c->ipc_peer[1].connected, "already connected");
  BEAT_HIP_CHECK(hipSetDevice(c->ctx->device));
  if (int rc = ipc_open_peer(c, c->ipc_peer[0], c->peer_lo, host_handle_lo)) return rc;
  if (int rc = ipc_open_peer(c, c->ipc_peer[1], c->peer_hi, host_handle_hi)) return rc;
  return BEAT_OK;
}

extern "C" int beat_comm_ipc_connect_all(beat_comm* c, const void* host_handles, int count) {
  BEAT_REQUIRE(c != nullptr && c->ipc && host_handles != nullptr, "not an ipc communicator");
  BEAT_REQUIRE(count == c->world && c->world <= BEAT_IPC_MAX_RANKS, "%d handles for %d ranks (at most %d)", count, c->world,
               BEAT_IPC_MAX_RANKS);
  BEAT_REQUIRE(!c->ipc_peer[0].connected && !c->ipc_peer[1].connected && !c->ipc_all_connected, "already connected");
  BEAT_HIP_CHECK(hipSetDevice(c->ctx->device));
  c->ipc_all_connected = true;  // from here on comm_free closes what has been opened
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) {
      c->ipc_all[r] = c->ipc_box;
      continue;
    }
    IpcHandle h;
    std::memcpy(&h, (const char*)host_handles + (size_t)r * BEAT_IPC_HANDLE_BYTES, sizeof(h));
    BEAT_REQUIRE(h.magic == IPC_MAGIC && (int)h.rank == r, "handle %d is not rank %d's ipc handle", r, r);
    BEAT_REQUIRE(h.plane_max == c->ipc_plane_max, "rank %d sized its mailbox for planes of %lld doubles, this rank for %lld", r,
                 (long long)h.plane_max, (long long)c->ipc_plane_max);
    BEAT_HIP_CHECK(hipIpcOpenMemHandle((void**)&c->ipc_all[r], h.inbox, hipIpcMemLazyEnablePeerAccess));
  }
  const int peers[2] = {c->peer_lo, c->peer_hi};
  for (int d = 0; d < 2; ++d) {
    if (peers[d] < 0) continue;
    c->ipc_peer[d].box = c->ipc_all[peers[d]];
    c->ipc_peer[d].self = peers[d] == c->rank;
    c->ipc_peer[d].connected = true;
  }
  return BEAT_OK;
}

// Ranks that live in ONE process (threads, each with its own context and streams: the rehearsal of 8 and 16 ranks on a box
// that lets six processes at most onto its GPU): the mailboxes are this process's own allocations, so there is nothing
// to export or open -- every rank is handed the other communicators themselves.  Same kernels, same flags, same slots.
extern "C" int beat_comm_ipc_connect_local(beat_comm* c, beat_comm* const* host_comms, int count) {
  BEAT_REQUIRE(c != nullptr && c->ipc && host_comms != nullptr, "not an ipc communicator");
  BEAT_REQUIRE(count == c->world && c->world <= BEAT_IPC_MAX_RANKS, "%d communicators for %d ranks (at most %d)", count, c->world,
               BEAT_IPC_MAX_RANKS);
  BEAT_REQUIRE(!c->ipc_peer[0].connected && !c->ipc_peer[1].connected && !c->ipc_all_connected, "already connected");
  for (int r = 0; r < c->world; ++r) {
    const beat_comm* o = host_comms[r];
    BEAT_REQUIRE(o != nullptr && o->ipc && o->rank == r && o->world == c->world && o->ipc_box != nullptr, "entry %d is not rank %d's ipc communicator", r, r);
    BEAT_REQUIRE(o->ipc_plane_max == c->ipc_plane_max && o->ctx->device == c->ctx->device, "rank %d: another plane size or another device", r);
    c->ipc_all[r] = o->ipc_box;
  }
  c->ipc_all_connected = true;
  c->ipc_local = true;  // (nothing was opened: comm_free closes nothing; the owners free their mailboxes)
  const int peers[2] = {c->peer_lo, c->peer_hi};
  for (int d = 0; d < 2; ++d) {
    if (peers[d] < 0) continue;
    c->ipc_peer[d].box = c->ipc_all[peers[d]];
    c->ipc_peer[d].self = true;  // never closed
    c->ipc_peer[d].connected = true;
  }
  return BEAT_OK;
}

extern "C" int beat_comm_destroy(beat_comm* c) {
  comm_free(c);
  return BEAT_OK;
}

extern "C" int beat_comm_info(beat_comm* c, int* host_out) {
  BEAT_REQUIRE(c != nullptr && host_out != nullptr, "null argument");
  host_out[0] = c->ipc ? BEAT_TRANSPORT_IPC : c->rccl ? (c->serial ? BEAT_TRANSPORT_RCCL_SERIAL : BEAT_TRANSPORT_RCCL)
                                                      : BEAT_TRANSPORT_CALLBACKS;
  int count = 0;
  if (c->coll) BEAT_RCCL_CHECK(g_rccl.CommCount(c->coll, &count));  // ranks as RCCL itself counts them
  host_out[1] = count;
  host_out[2] = c->world;
  host_out[3] = c->coll != nullptr ? 1 : (c->ipc && c->allreduce == nullptr) ? 2 : 0;  // all-reduces: RCCL, mailboxes, the caller
  return BEAT_OK;
}

extern "C" int64_t beat_comm_merged_solves(const beat_comm* c) { return c ? (int64_t)c->merged_solves : 0; }

extern "C" int beat_comm_profile(beat_comm* c, int enable) {
  BEAT_REQUIRE(c != nullptr, "null argument");
  if (enable) {
    if (c->side) BEAT_HIP_CHECK(hipStreamSynchronize(c->side));
    BEAT_HIP_CHECK(hipStreamSynchronize(c->ctx->stream));
    prof_clear(c);
  }
  c->profiling = enable != 0;
  return BEAT_OK;
}

extern "C" int beat_comm_profile_read(beat_comm* c, double* host_out) {
  BEAT_REQUIRE(c != nullptr && host_out != nullptr, "null argument");
  if (c->side) BEAT_HIP_CHECK(hipStreamSynchronize(c->side));
  BEAT_HIP_CHECK(hipStreamSynchronize(c->ctx->stream));
  for (int k = 0; k < 6; ++k) host_out[k] = 0.0;
  for (const ProfSpan& s : c->spans) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s.a, s.b) != hipSuccess) continue;  // a span whose end was never recorded
    host_out[2 * s.kind] += ms;
    host_out[2 * s.kind + 1] += 1.0;
  }
  return BEAT_OK;
}

namespace {
int ipc_check(beat_comm* c) {
  if (c->ipc && c->ipc_err && *(volatile int*)c->ipc_err) {
    beat_set_error("ipc transport: rank %d gave up waiting for a neighbour (peers %d, %d): the exchanged planes are not valid",
                   c->rank, c->peer_lo, c->peer_hi);
    return BEAT_EHIP;
  }
  return BEAT_OK;
}

int ipc_halo_start(beat_comm* c, double* const* fields, int nf, int64_t n, int64_t plane) {
  BEAT_REQUIRE(plane <= c->ipc_plane_max, "plane of %lld doubles, the mailboxes hold %lld", (long long)plane,
               (long long)c->ipc_plane_max);
  BEAT_REQUIRE(nf <= IPC_FIELDS, "at most %d fields per exchange", IPC_FIELDS);
  const int peers[2] = {c->peer_lo, c->peer_hi};
  for (int d = 0; d < 2; ++d)
    BEAT_REQUIRE(peers[d] < 0 || c->ipc_peer[d].connected, "beat_comm_ipc_connect has not been called");
  if (int rc = ipc_check(c)) return rc;
  const uint64_t s = c->ipc_seq++;
  const int slot = (int)(s % IPC_SLOTS);
  const int64_t pm = c->ipc_plane_max;
  IpcFlags* mine = ipc_flags(c->ipc_box, pm);
  BEAT_HIP_CHECK(hipEventRecord(c->ev_ready, c->ctx->stream));
  BEAT_HIP_CHECK(hipStreamWaitEvent(c->side, c->ev_ready, 0));
  const int span = prof_begin(c, 0, c->side);
  const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(IPC_BLOCKS, (plane + 2047) / 2048));
  // all sends first (to either neighbour, one launch), then the receives (one launch): no rank waits for a message
  // before it has posted its own
  IpcXferArgs snd{}, rcv{};
  snd.nf = rcv.nf = nf;
  snd.plane = rcv.plane = plane;
  snd.ticks = rcv.ticks = c->ipc_ticks;
  snd.err = rcv.err = c->ipc_err;
  for (int d = 0; d < 2; ++d) {
    if (peers[d] < 0) continue;
    IpcPeer& P = c->ipc_peer[d];
    const int od = 1 - d;  // this rank is the neighbour's neighbour in the opposite direction
    IpcXfer& x = snd.dir[d];
    for (int k = 0; k < nf; ++k) {
      x.src[k] = d == 0 ? fields[k] : fields[k] + n - plane;
      x.dst[k] = ipc_slot(P.box, pm, od, slot, k);
    }
    if (s >= (uint64_t)IPC_SLOTS) {  // the slot's previous message must have left the neighbour's mailbox
      x.wait_flag = &mine->freed[d];
      x.wait_need = s - IPC_SLOTS + 1;
    }
    x.signal_flag = &ipc_flags(P.box, pm)->arrived[od];
    x.signal_value = s + 1;
    x.counter = c->ipc_counters + d;
    IpcXfer& y = rcv.dir[d];
    for (int k = 0; k < nf; ++k) {
      y.src[k] = ipc_slot(c->ipc_box, pm, d, slot, k);
      y.dst[k] = d == 0 ? fields[k] - plane : fields[k] + n;
    }
    y.wait_flag = &mine->arrived[d];
    y.wait_need = s + 1;
    y.signal_flag = &ipc_flags(P.box, pm)->freed[od];
    y.signal_value = s + 1;
    y.counter = c->ipc_counters + 2 + d;
  }
  BEAT_KERNEL(ipc_xfer_kernel, dim3(blocks, 2), dim3(256), 0, c->side, snd);
  BEAT_KERNEL(ipc_xfer_kernel, dim3(blocks, 2), dim3(256), 0, c->side, rcv);
  BEAT_LAUNCH_CHECK();
  prof_end(c, span, c->side);
  BEAT_HIP_CHECK(hipEventRecord(c->ev_halo, c->side));
  return BEAT_OK;
}
}  // namespace

// Start the exchange of the boundary planes of `f` (interior pointer, n doubles, ghost planes around it) -- and of a
// second field `f2` in the same RCCL group when given (one group latency for both) -- on the side stream once the
// compute stream has produced the planes.  With callbacks the exchange completes here.
static int halo_start(beat_comm* c, double* f, int64_t n, int64_t plane, double* f2 = nullptr) {
  if (c->peer_lo < 0 && c->peer_hi < 0) return BEAT_OK;
  double* fields[2] = {f, f2};
  const int nf = f2 ? 2 : 1;
  if (c->ipc) return ipc_halo_start(c, fields, nf, n, plane);
  if (!c->rccl) {
    for (int k = 0; k < nf; ++k) {
      double* g = fields[k];
      const int rc = c->halo(c->user, c->peer_lo >= 0 ? g : nullptr, c->peer_lo >= 0 ? g - plane : nullptr,
                             c->peer_hi >= 0 ? g + n - plane : nullptr, c->peer_hi >= 0 ? g + n : nullptr, plane);
      if (rc) {
        beat_set_error("halo callback failed (%d)", rc);
        return BEAT_EHIP;
      }
    }
    return BEAT_OK;
  }
  hipStream_t stream = c->serial ? c->ctx->stream : c->side;
  ncclComm_t comm = c->serial ? c->coll : c->p2p;
  if (!c->serial) {
    BEAT_HIP_CHECK(hipEventRecord(c->ev_ready, c->ctx->stream));
    BEAT_HIP_CHECK(hipStreamWaitEvent(c->side, c->ev_ready, 0));
  }
  const int span = prof_begin(c, 0, stream);
  // posting order: both sends, then the receives in the opposite order -- between two different ranks messages of
  // one direction are matched in the order posted (field by field on both sides); on a one-rank communicator whose
  // two peers are the rank itself (tests) it makes the exchange periodic (ghost_hi <- first plane, ghost_lo <- last)
  BEAT_RCCL_CHECK(g_rccl.GroupStart());
  for (int k = 0; k < nf; ++k) {
    double* g = fields[k];
    const double* first = c->peer_lo >= 0 ? g : nullptr;
    double* ghost_lo = c->peer_lo >= 0 ? g - plane : nullptr;
    const double* last = c->peer_hi >= 0 ? g + n - plane : nullptr;
    double* ghost_hi = c->peer_hi >= 0 ? g + n : nullptr;
    if (first) BEAT_RCCL_CHECK(g_rccl.Send(first, (size_t)plane, ncclDouble, c->peer_lo, comm, stream));
    if (last) BEAT_RCCL_CHECK(g_rccl.Send(last, (size_t)plane, ncclDouble, c->peer_hi, comm, stream));
    if (ghost_hi) BEAT_RCCL_CHECK(g_rccl.Recv(ghost_hi, (size_t)plane, ncclDouble, c->peer_hi, comm, stream));
    if (ghost_lo) BEAT_RCCL_CHECK(g_rccl.Recv(ghost_lo, (size_t)plane, ncclDouble, c->peer_lo, comm, stream));
  }
  BEAT_RCCL_CHECK(g_rccl.GroupEnd());
  prof_end(c, span, stream);
  if (!c->serial) BEAT_HIP_CHECK(hipEventRecord(c->ev_halo, c->side));
  return BEAT_OK;
}

// Make the compute stream wait for the ghost planes of the exchange started last.
static int halo_wait(beat_comm* c) {
  if (!(c->rccl || c->ipc) || c->serial || (c->peer_lo < 0 && c->peer_hi < 0)) return BEAT_OK;
  const int span = prof_begin(c, 2, c->ctx->stream);
  BEAT_HIP_CHECK(hipStreamWaitEvent(c->ctx->stream, c->ev_halo, 0));
  prof_end(c, span, c->ctx->stream);
  return BEAT_OK;
}

// In-place sum over the ranks.  Inside beat_pde_solve_dist the all-reduces of iterations enqueued beyond the one that
// latched STOP still run (the stage kernels around them exit at once): they re-sum slots that were already reduced
// (PQ, RZN, RRN grow by a factor `world` per such iteration); nothing reads those slots afterwards -- ITERS, RR, BB,
// REASON and NUPD are written by the scalar step, which the latch stops.
static int allreduce_sum(beat_comm* c, double* dev, int count) {
  if (c->ipc && c->ipc_all_connected && c->coll == nullptr && c->allreduce == nullptr) {
    BEAT_REQUIRE(count >= 1 && count <= IPC_AR_VALUES, "the ipc all-reduce takes 1..%d values", IPC_AR_VALUES);
    if (int rc = ipc_check(c)) return rc;
    IpcArArgs a{};
    for (int r = 0; r < c->world; ++r) a.box[r] = c->ipc_all[r];
    a.ar_offset = ipc_ar_offset(c->ipc_plane_max);
    a.rank = c->rank;
    a.world = c->world;
    a.count = count;
    a.seq = c->ipc_ar_seq++;
    a.values = dev;
    a.ticks = c->ipc_ticks;
    a.err = c->ipc_err;
    const int span = prof_begin(c, 1, c->ctx->stream);
    BEAT_KERNEL(ipc_allreduce_kernel, dim3(1), dim3(64), 0, c->ctx->stream, a);
    BEAT_LAUNCH_CHECK();
    prof_end(c, span, c->ctx->stream);
    return BEAT_OK;
  }
  if (c->coll) {
    const int span = prof_begin(c, 1, c->ctx->stream);
    BEAT_RCCL_CHECK(g_rccl.AllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, c->coll, c->ctx->stream));
    prof_end(c, span, c->ctx->stream);
    return BEAT_OK;
  }
  BEAT_REQUIRE(c->allreduce != nullptr, "this ipc communicator reduces through the mailboxes: connect it with beat_comm_ipc_connect_all");
  const int rc = c->allreduce(c->user, dev, count);
  if (rc) {
    beat_set_error("all-reduce callback failed (%d)", rc);
    return BEAT_EHIP;
  }
  return BEAT_OK;
}

extern "C" int beat_comm_halo_exchange(beat_comm* comm, double* dev_field, int64_t n, int64_t plane_doubles) {
  BEAT_REQUIRE(comm != nullptr && dev_field != nullptr && plane_doubles > 0 && n >= plane_doubles, "bad argument");
  if (int rc = halo_start(comm, dev_field, n, plane_doubles)) return rc;
  return halo_wait(comm);
}

extern "C" int beat_comm_allreduce_sum(beat_comm* comm, double* dev_values, int count) {
  BEAT_REQUIRE(comm != nullptr && dev_values != nullptr && count > 0, "bad argument");
  return allreduce_sum(comm, dev_values, count);
}

// ---- the decomposed solve in two halves (round 5: as beat_solve_begin / beat_solve_end of the single slab) ------------------------------
// begin: ghost planes, right-hand side, the iterations the previous solve needed + 1 with their exchanges and all-reduces, and the
// copy of the scalar state are in the streams when it returns; end: the host looks at the latch, enqueues more iterations if the
// residual asks for them, and closes the solve.  Between the two a caller may enqueue the next ionic launch behind the solve
// (beat_ode_step_pending with pending = -1): every rank sees the same all-reduced scalars, so every rank's launch does the same.
namespace {
// iterations [o.launched, o.launched + count) of the open decomposed solve
int dist_enqueue_iterations(beat_pde* pde, int count) {
  beat_pde::OpenSolve& o = pde->open;
  beat_comm* comm = (beat_comm*)o.comm;
  const int64_t n = pde->n, plane = pde->g.plane, fld = beat_pde_field_stride(pde);
  double* r = o.work + plane;
  double* q = r + fld;
  double* ring = q + 2 * fld;  // [r, q, z, ring...]: z is unused by the Jacobi path
  const int PR = pde->ring;     // (6 on a decomposed grid; a one-rank communicator on a single slab of per-node rows: 12)
  double* st = pde->d_st;
  double* dev_x = o.x;
  double* rbuf[2] = {r, q};  // rr: the residual update writes out of place
  const bool rr = o.rr, merged = o.merged, vpdot = o.vpdot;
  int rc;
  for (int it = 0; it < count; ++it) {
    const int i = o.launched + it, slot = i % PR;
    double* p_cur = ring + (int64_t)slot * fld;
    double* p_next = ring + (int64_t)((i + 1) % PR) * fld;
    if (merged) {
      // u_i . A u_i, r_i . u_i, r_i . r_i in one pass over r_i (interior planes while its ghost planes travel, then the
      // boundary planes), ONE all-reduce, the scalar step (stopping test, beta_i, alpha_i), then p_i = u_i + beta_i p_{i-1}
      // and r_{i+1} = r_i - alpha_i A p_i in one pass without a dot product; the ghost planes of r_{i+1} travel behind it
      const double* p_old = ring + (int64_t)((i + PR - 1) % PR) * fld;
      double* r_cur = rbuf[i & 1];
      double* r_new = rbuf[(i + 1) & 1];
      if ((rc = beat_rr_udot_part(pde, st, r_cur, 0))) return rc;
      if ((rc = halo_wait(comm))) return rc;
      if ((rc = beat_rr_udot_part(pde, st, r_cur, 1))) return rc;
      if ((rc = allreduce_sum(comm, st + PQ, 3))) return rc;
      if ((rc = beat_rr_merged_next(pde, st, slot))) return rc;
      if ((rc = beat_rr_prupd(pde, st, r_cur, p_old, p_cur, r_new))) return rc;
      if ((rc = halo_start(comm, r_new, n, plane))) return rc;
      if (slot == PR - 1) {
        if ((rc = beat_pde_x_flush_terms(pde, st, dev_x, ring, fld, i + 1 - PR, 1, beat_guess_terms(pde, i + 1 - PR))))
          return rc;
      }
      continue;
    }
    if (rr) {
      // p_i = D^-1 r_i + beta p_{i-1} and p_i . A p_i: the planes that need no ghost data while the ghost planes of
      // r_i travel, then the boundary planes, which also keep p_i on the ghost planes (no exchange of p)
      const double* p_old = ring + (int64_t)((i + PR - 1) % PR) * fld;
      double* r_cur = rbuf[i & 1];
      double* r_new = rbuf[(i + 1) & 1];
      if ((rc = beat_rr_pdot_part(pde, st, r_cur, p_old, p_cur, 0))) return rc;
      if ((rc = halo_wait(comm))) return rc;
      if ((rc = beat_rr_pdot_part(pde, st, r_cur, p_old, p_cur, 1))) return rc;
      if ((rc = allreduce_sum(comm, st + PQ, 1))) return rc;
      if ((rc = beat_rr_rupd(pde, st, r_cur, r_new, p_cur, slot, false))) return rc;  // r_{i+1}, local r.z and r.r
      if ((rc = halo_start(comm, r_new, n, plane))) return rc;  // travels behind the reductions and the next part 0
      if ((rc = allreduce_sum(comm, st + RZN, 2))) return rc;
      if (slot == PR - 1) {
        if ((rc = beat_pde_x_flush_terms(pde, st, dev_x, ring, fld, i + 1 - PR, 1, beat_guess_terms(pde, i + 1 - PR))))
          return rc;
      }
      if ((rc = beat_rr_next(pde, st))) return rc;
      continue;
    }
    if (vpdot) {
      // p_i = D^-1 r_i + beta p_{i-1}, q = A p_i and p_i . q in one pass over the coefficient rows: the tiles that need no ghost
      // plane while the ghost planes of r_i travel, then the boundary tiles (which keep p_i on the ghost planes); the residual
      // update in place, its ghost planes travelling behind the second reduction and the next pass's first part
      const double* p_old = ring + (int64_t)((i + PR - 1) % PR) * fld;
      if ((rc = beat_vtl_pdot_part(pde, st, r, p_old, p_cur, q, i == 0, 0))) return rc;
      if ((rc = halo_wait(comm))) return rc;
      if ((rc = beat_vtl_pdot_part(pde, st, r, p_old, p_cur, q, i == 0, 1))) return rc;
      if ((rc = allreduce_sum(comm, st + PQ, 1))) return rc;
      if ((rc = beat_pde_cg_update_r(pde, st, r, q, slot))) return rc;
      if ((rc = halo_start(comm, r, n, plane))) return rc;
      if ((rc = allreduce_sum(comm, st + RZN, 2))) return rc;
      if (slot == PR - 1) {
        if ((rc = beat_pde_x_flush_terms(pde, st, dev_x, ring, fld, i + 1 - PR, 1, beat_guess_terms(pde, i + 1 - PR))))
          return rc;
      }
      if ((rc = beat_rr_next(pde, st))) return rc;  // the scalar roll (beta, iteration count, latch)
      continue;
    }
    if ((rc = halo_start(comm, p_cur, n, plane))) return rc;                  // ghost planes of p travel ...
    if ((rc = beat_pde_spmv_dot_part(pde, p_cur, q, st, 0))) return rc;       // ... while the interior is computed
    if ((rc = halo_wait(comm))) return rc;
    if ((rc = beat_pde_spmv_dot_part(pde, p_cur, q, st, 1))) return rc;       // boundary planes + local p.q
    if ((rc = allreduce_sum(comm, st + PQ, 1))) return rc;
    if ((rc = beat_pde_cg_update_r(pde, st, r, q, slot))) return rc;
    if ((rc = allreduce_sum(comm, st + RZN, 2))) return rc;
    if (slot == PR - 1) {
      if ((rc = beat_pde_x_flush_terms(pde, st, dev_x, ring, fld, i + 1 - PR, 1, beat_guess_terms(pde, i + 1 - PR))))
        return rc;
    }
    if ((rc = beat_pde_cg_next_oop(pde, st, r, p_cur, p_next))) return rc;
  }
  o.launched += count;
  return BEAT_OK;
}
}  // namespace

int beat_dist_solve_begin(beat_pde* pde, beat_comm* comm, const double* dev_v_prev, const double* const* host_dev_stim_w,
                          const double* host_stim_amp, int n_stim, double* dev_x, double* dev_work, double rtol, double atol, int max_it) {
  BEAT_REQUIRE(pde != nullptr && comm != nullptr && dev_v_prev && dev_x && dev_work, "null argument");
  BEAT_REQUIRE(pde->ctx == comm->ctx, "operator and communicator belong to different contexts");
  BEAT_REQUIRE((pde->g.z_lo_phys != 0) == (comm->peer_lo < 0) && (pde->g.z_hi_phys != 0) == (comm->peer_hi < 0),
               "slab faces (lo_phys=%d, hi_phys=%d) do not match the communicator's peers (%d, %d)",
               pde->g.z_lo_phys, pde->g.z_hi_phys, comm->peer_lo, comm->peer_hi);
  BEAT_REQUIRE(pde->pc_ncoef == 1, "the in-library decomposed solve is Jacobi-PCG (polynomial preconditioner: stage functions)");
  BEAT_REQUIRE(max_it >= 0, "max_it must be >= 0");
  BEAT_REQUIRE(!pde->open.on, "the previous solve has not been finished (beat_pde_solve_end)");
  beat_ctx* ctx = pde->ctx;
  if (pde->h_st == nullptr) {
    BEAT_HIP_CHECK(hipHostMalloc((void**)&pde->h_st, sizeof(double) * 16, hipHostMallocDefault));
    BEAT_HIP_CHECK(hipEventCreateWithFlags(&pde->ev_st, hipEventDisableTiming));
  }
  const int64_t n = pde->n, plane = pde->g.plane, fld = beat_pde_field_stride(pde);
  double* r = dev_work + plane;
  double* q = r + fld;
  double* ring = q + 2 * fld;  // [r, q, z, ring...]: z is unused by the Jacobi path
  double* st = pde->d_st;
  double* h = ctx->h_pinned;
  int rc;
  const bool rr = beat_rr_available(pde);  // constant coefficients: the kernels that never store q = A p
  // ghost planes of v_ for the right-hand side (the reference's scatter_forward after the previous solve) and, in the
  // same exchange, of the guess increment e (written by the x update of the previous solve): they travel while the
  // right-hand side is built on the planes that need neither, the one or two boundary planes follow
  const bool guess_path = (rr || pde->var) && pde->guess_order != 0 && pde->d_guess != nullptr && pde->hist_n >= 1;
  if (rr || pde->var) {
    BEAT_REQUIRE(pde->have_dt, "beat_pde_set_timestep has not been called");
    BEAT_REQUIRE(n_stim >= 0 && n_stim <= BEAT_MAX_STIM, "at most %d stimuli", BEAT_MAX_STIM);
    BEAT_REQUIRE(!pde->guess_pending, "the previous solve's deferred update has not been applied");
    beat_guess_begin(pde);
    BEAT_REQUIRE(!pde->guess.use_e || guess_path, "guess increment without ghost planes");
  }
  if ((rc = halo_start(comm, const_cast<double*>(dev_v_prev), n, plane, guess_path ? pde->d_guess : nullptr))) return rc;
  static const bool split_rhs = [] {  // BEAT_DIST_RHS_SPLIT=0: wait for the planes first, one launch (A/B runs)
    const char* e = std::getenv("BEAT_DIST_RHS_SPLIT");
    return !(e && e[0] == '0');
  }();
  if (rr) {
    if (split_rhs) {
      if ((rc = beat_rr_rhs(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, r, st, 0))) return rc;
      if ((rc = halo_wait(comm))) return rc;
      rc = beat_rr_rhs(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, r, st, 1);
    } else {
      if ((rc = halo_wait(comm))) return rc;
      rc = beat_rr_rhs(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, r, st);
    }
  } else if (pde->var) {
    const double* e = pde->guess.use_e ? pde->guess.e : nullptr;
    if (split_rhs) {
      if ((rc = beat_var_rhs(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, r, ring, st, e, 0))) return rc;
      if ((rc = halo_wait(comm))) return rc;
      rc = beat_var_rhs(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, r, ring, st, e, 1);
    } else {
      if ((rc = halo_wait(comm))) return rc;
      rc = beat_var_rhs(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, r, ring, st, e);
    }
  } else {
    if ((rc = halo_wait(comm))) return rc;
    rc = beat_pde_rhs(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, r, ring, st);
  }
  if (rc) return rc;
  if ((rc = allreduce_sum(comm, st + BB, 3))) return rc;
  if ((rc = beat_pde_cg_begin(pde, st, rtol, atol, max_it))) return rc;
  // beat_pde_set_single_reduction / BEAT_DIST_MERGED=1 (constant coefficients): ONE all-reduce per iteration
  // (beat_pde_rr.hip, beat_rr_udot_part); the environment is read per solve.  The pass that finds r_k converged is one more than the k updates: the first chunk and the limit count it.
  const char* merged_env = std::getenv("BEAT_DIST_MERGED");
  const bool merged = rr && (pde->single_reduction >= 0 ? pde->single_reduction == 1 : (merged_env != nullptr && merged_env[0] == '1'));
  comm->merged_solves += merged ? 1 : 0;
  if (rr && (rc = halo_start(comm, r, n, plane))) return rc;  // ghost planes of r_0
  // per-node rows: the fused tile pass (direction formed while loading, csrc/beat_pde_vtl.hip) on the slab -- like the
  // register-row path it exchanges r, forms the direction on the ghost planes itself and never exchanges p.  It needs the
  // centre coefficients of the neighbours' boundary planes: one exchange per operator, through the work field q (free until the
  // first iteration).
  // Whether the pass is to be had is a property of the rank's slab (tiles of 8 rows, a tile list), and all ranks must walk the
  // same sequence of exchanges and reductions: they agree once per operator (a sum over the ranks of "I can").
  const bool var_dist = !rr && pde->var && !(pde->g.z_lo_phys && pde->g.z_hi_phys);
  if (var_dist && !pde->v_gc0_valid) {
    h[0] = beat_vtl_pdot_dist_available(pde) ? 1.0 : 0.0;
    BEAT_HIP_CHECK(hipMemcpyAsync(ctx->d_small, h, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = allreduce_sum(comm, ctx->d_small, 1))) return rc;
    BEAT_HIP_CHECK(hipMemcpyAsync(h, ctx->d_small, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    pde->v_pdot_dist = h[0] == (double)comm->world;
    if (pde->v_pdot_dist) {
      BEAT_HIP_CHECK(hipMemcpyAsync(q, pde->v_A, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));  // slot 0 of the rows
      if ((rc = halo_start(comm, q, n, plane))) return rc;
      if ((rc = halo_wait(comm))) return rc;
      BEAT_HIP_CHECK(hipMemcpyAsync(pde->v_gc0, q - plane, sizeof(double) * (size_t)plane, hipMemcpyDeviceToDevice, ctx->stream));
      BEAT_HIP_CHECK(hipMemcpyAsync(pde->v_gc0 + plane, q + n, sizeof(double) * (size_t)plane, hipMemcpyDeviceToDevice, ctx->stream));
    }
    pde->v_gc0_valid = true;
  }
  const bool vpdot = var_dist && pde->v_pdot_dist;
  if (vpdot) {
    if ((rc = halo_start(comm, r, n, plane))) return rc;  // ghost planes of r_0
  }
  beat_pde::OpenSolve& o = pde->open;
  o = beat_pde::OpenSolve{};
  o.comm = comm;
  o.rr = rr;
  o.merged = merged;
  o.vpdot = vpdot;
  o.v_prev = dev_v_prev;
  o.x = dev_x;
  o.work = dev_work;
  o.rtol = rtol;
  o.atol = atol;
  o.max_it = max_it;
  o.limit = max_it + (merged ? 1 : 0);
  const int chunk = std::min(beat_pde_first_chunk(pde) + (merged ? 1 : 0), o.limit);
  if ((rc = dist_enqueue_iterations(pde, chunk))) return rc;
  BEAT_HIP_CHECK(hipMemcpyAsync(pde->h_st, st, sizeof(double) * 16, hipMemcpyDeviceToHost, ctx->stream));
  BEAT_HIP_CHECK(hipEventRecord(pde->ev_st, ctx->stream));
  o.on = true;
  return BEAT_OK;
}

int beat_dist_solve_end(beat_pde* pde, int defer_flush, beat_ksp_info* info, int* host_pending, bool* needed_more) {
  beat_pde::OpenSolve& o = pde->open;
  beat_comm* comm = (beat_comm*)o.comm;
  beat_ctx* ctx = pde->ctx;
  double* h = pde->h_st;
  double* st = pde->d_st;
  const int64_t fld = beat_pde_field_stride(pde);
  double* ring = o.work + pde->g.plane + 3 * fld;
  const int PR = pde->ring;
  int rc;
  BEAT_HIP_CHECK(hipEventSynchronize(pde->ev_st));
  if ((rc = ipc_check(comm))) {
    o.on = false;
    return rc;
  }
  if (needed_more && h[STOP] == 0.0) *needed_more = true;  // unlatched at the first look: the launch behind the solve did nothing (beat_solve_end)
  while (!(h[STOP] != 0.0 || o.launched >= o.limit)) {
    if ((rc = dist_enqueue_iterations(pde, std::min(2, o.limit - o.launched)))) {
      o.on = false;
      return rc;
    }
    BEAT_HIP_CHECK(hipMemcpyAsync(h, st, sizeof(double) * 16, hipMemcpyDeviceToHost, ctx->stream));
    BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if ((rc = ipc_check(comm))) {
      o.on = false;
      return rc;
    }
  }
  o.on = false;
  pde->applied_behind = false;
  if (o.rr || o.vpdot) {  // the exchange started after the last residual update has no consumer: drain it before anything else
    if ((rc = halo_wait(comm))) return rc;  // touches those ghost planes
  }
  const int nupd = (int)h[NUPD], base = (nupd / PR) * PR;
  const GuessTerms last = beat_guess_terms(pde, base);
  beat_guess_observe(pde, (int)h[ITERS]);
  if (beat_guess_end(pde, nupd, defer_flush != 0)) {  // the last partial ring cycle and / or the guess increment
    if (defer_flush) {
      host_pending[0] = base;
      host_pending[1] = nupd % PR;
      pde->last_base = base;
    } else if ((rc = beat_pde_x_flush_terms(pde, st, o.x, ring, fld, base, 0, last))) {
      return rc;
    }
  }
  const int iters = (int)h[ITERS];
  pde->last_iters = iters;
  int reason = (int)h[REASON];
  if (h[STOP] == 0.0) reason = -3;
  pde->last_info.iterations = iters;
  pde->last_info.converged_reason = reason;
  pde->last_info.residual_norm = std::sqrt(h[RR]);
  pde->last_info.rhs_norm = std::sqrt(h[BB]);
  if (info) *info = pde->last_info;
  pde->last_rc = BEAT_OK;
  if (reason < 0) {
    beat_set_error("PCG did not converge in %d iterations (||r|| = %.3e, ||b|| = %.3e)", iters, std::sqrt(h[RR]),
                   std::sqrt(h[BB]));
    pde->last_rc = BEAT_ENOTCONV;
  }
  return pde->last_rc;
}

extern "C" int beat_pde_solve_dist_begin(beat_pde* pde, beat_comm* comm, const double* dev_v_prev, const double* const* host_dev_stim_w,
                                         const double* host_stim_amp, int n_stim, double* dev_x, double* dev_work, double rtol, double atol,
                                         int max_it) {
  return beat_dist_solve_begin(pde, comm, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, dev_work, rtol, atol, max_it);
}

extern "C" int beat_pde_solve_dist(beat_pde* pde, beat_comm* comm, const double* dev_v_prev,
                                   const double* const* host_dev_stim_w, const double* host_stim_amp, int n_stim,
                                   double* dev_x, double* dev_work, double rtol, double atol, int max_it,
                                   int defer_flush, beat_ksp_info* info, int* host_pending) {
  BEAT_REQUIRE(!defer_flush || host_pending != nullptr, "defer_flush needs host_pending[2]");
  if (host_pending) host_pending[0] = host_pending[1] = 0;
  if (int rc = beat_dist_solve_begin(pde, comm, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, dev_work, rtol, atol, max_it)) return rc;
  return beat_dist_solve_end(pde, defer_flush, info, host_pending, nullptr);
}
