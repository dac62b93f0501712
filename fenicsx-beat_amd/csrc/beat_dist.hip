// Slab-decomposed diffusion solve with the communication inside the library: RCCL ghost-plane exchange on a
// side HIP stream overlapped with the interior stencil, RCCL all-reduces of the PCG dot products on the compute
// stream, one C call per solve.  Replaces b.ghostUpdate / KSP.solve / scatter_forward on a partitioned mesh
// (src/beat/base_model.py:203-206,236,242).  See include/beat_hip.h for the contract.
#include "beat_pde_internal.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <mutex>

namespace {
using namespace beat_pde_detail;

// librccl is opened on first use: the library has no link-time dependency on it (CPU-only hosts can load
// libbeat_hip.so and check its exports; single-GPU users never touch RCCL).
struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
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

struct beat_comm {
  beat_ctx* ctx = nullptr;
  int rank = 0, world = 1, peer_lo = -1, peer_hi = -1;
  bool rccl = false;
  ncclComm_t p2p = nullptr, coll = nullptr;
  hipStream_t side = nullptr;          // ghost-plane traffic (non-blocking stream owned by the communicator)
  hipEvent_t ev_ready = nullptr;       // compute -> side: the planes to send are final
  hipEvent_t ev_halo = nullptr;        // side -> compute: the ghost planes have arrived
  beat_halo_fn halo = nullptr;
  beat_allreduce_fn allreduce = nullptr;
  void* user = nullptr;
};

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

extern "C" int beat_comm_create_rccl(beat_ctx* ctx, int rank, int world, int peer_lo, int peer_hi, const void* host_id,
                                     beat_comm** out) {
  BEAT_REQUIRE(ctx != nullptr && host_id != nullptr && out != nullptr, "null argument");
  if (int rc = check_peers(rank, world, peer_lo, peer_hi)) return rc;
  if (int rc = load_rccl()) return rc;
  BEAT_HIP_CHECK(hipSetDevice(ctx->device));
  beat_comm* c = new beat_comm();
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  c->peer_lo = peer_lo;
  c->peer_hi = peer_hi;
  c->rccl = true;
  const ncclUniqueId* ids = (const ncclUniqueId*)host_id;
  BEAT_RCCL_CHECK(g_rccl.CommInitRank(&c->p2p, world, ids[0], rank));
  BEAT_RCCL_CHECK(g_rccl.CommInitRank(&c->coll, world, ids[1], rank));
  BEAT_HIP_CHECK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
  BEAT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
  BEAT_HIP_CHECK(hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming));
  *out = c;
  return BEAT_OK;
}

extern "C" int beat_comm_create_callbacks(beat_ctx* ctx, int rank, int world, int peer_lo, int peer_hi,
                                          beat_halo_fn halo, beat_allreduce_fn allreduce, void* user, beat_comm** out) {
  BEAT_REQUIRE(ctx != nullptr && halo != nullptr && allreduce != nullptr && out != nullptr, "null argument");
  if (int rc = check_peers(rank, world, peer_lo, peer_hi)) return rc;
  beat_comm* c = new beat_comm();
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  c->peer_lo = peer_lo;
  c->peer_hi = peer_hi;
  c->halo = halo;
  c->allreduce = allreduce;
  c->user = user;
  *out = c;
  return BEAT_OK;
}

extern "C" int beat_comm_destroy(beat_comm* c) {
  if (c == nullptr) return BEAT_OK;
  if (c->rccl) {
    (void)hipSetDevice(c->ctx->device);
    (void)hipStreamSynchronize(c->side);
    (void)hipStreamSynchronize(c->ctx->stream);
    if (c->p2p) (void)g_rccl.CommDestroy(c->p2p);
    if (c->coll) (void)g_rccl.CommDestroy(c->coll);
    (void)hipEventDestroy(c->ev_ready);
    (void)hipEventDestroy(c->ev_halo);
    (void)hipStreamDestroy(c->side);
  }
  delete c;
  return BEAT_OK;
}

// Start the exchange of the boundary planes of `f` (interior pointer, n doubles, ghost planes around it) -- and of a
// second field `f2` in the same RCCL group when given (one group latency for both) -- on the side stream once the
// compute stream has produced the planes.  With callbacks the exchange completes here.
static int halo_start(beat_comm* c, double* f, int64_t n, int64_t plane, double* f2 = nullptr) {
  if (c->peer_lo < 0 && c->peer_hi < 0) return BEAT_OK;
  double* fields[2] = {f, f2};
  const int nf = f2 ? 2 : 1;
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
  BEAT_HIP_CHECK(hipEventRecord(c->ev_ready, c->ctx->stream));
  BEAT_HIP_CHECK(hipStreamWaitEvent(c->side, c->ev_ready, 0));
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
    if (first) BEAT_RCCL_CHECK(g_rccl.Send(first, (size_t)plane, ncclDouble, c->peer_lo, c->p2p, c->side));
    if (last) BEAT_RCCL_CHECK(g_rccl.Send(last, (size_t)plane, ncclDouble, c->peer_hi, c->p2p, c->side));
    if (ghost_hi) BEAT_RCCL_CHECK(g_rccl.Recv(ghost_hi, (size_t)plane, ncclDouble, c->peer_hi, c->p2p, c->side));
    if (ghost_lo) BEAT_RCCL_CHECK(g_rccl.Recv(ghost_lo, (size_t)plane, ncclDouble, c->peer_lo, c->p2p, c->side));
  }
  BEAT_RCCL_CHECK(g_rccl.GroupEnd());
  BEAT_HIP_CHECK(hipEventRecord(c->ev_halo, c->side));
  return BEAT_OK;
}

// Make the compute stream wait for the ghost planes of the exchange started last.
static int halo_wait(beat_comm* c) {
  if (!c->rccl || (c->peer_lo < 0 && c->peer_hi < 0)) return BEAT_OK;
  BEAT_HIP_CHECK(hipStreamWaitEvent(c->ctx->stream, c->ev_halo, 0));
  return BEAT_OK;
}

static int allreduce_sum(beat_comm* c, double* dev, int count) {
  if (c->rccl) {
    BEAT_RCCL_CHECK(g_rccl.AllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, c->coll, c->ctx->stream));
    return BEAT_OK;
  }
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

extern "C" int beat_pde_solve_dist(beat_pde* pde, beat_comm* comm, const double* dev_v_prev,
                                   const double* const* host_dev_stim_w, const double* host_stim_amp, int n_stim,
                                   double* dev_x, double* dev_work, double rtol, double atol, int max_it,
                                   int defer_flush, beat_ksp_info* info, int* host_pending) {
  BEAT_REQUIRE(!defer_flush || host_pending != nullptr, "defer_flush needs host_pending[2]");
  if (host_pending) host_pending[0] = host_pending[1] = 0;
  BEAT_REQUIRE(pde != nullptr && comm != nullptr && dev_v_prev && dev_x && dev_work, "null argument");
  BEAT_REQUIRE(pde->ctx == comm->ctx, "operator and communicator belong to different contexts");
  BEAT_REQUIRE((pde->g.z_lo_phys != 0) == (comm->peer_lo < 0) && (pde->g.z_hi_phys != 0) == (comm->peer_hi < 0),
               "slab faces (lo_phys=%d, hi_phys=%d) do not match the communicator's peers (%d, %d)",
               pde->g.z_lo_phys, pde->g.z_hi_phys, comm->peer_lo, comm->peer_hi);
  BEAT_REQUIRE(pde->pc_ncoef == 1, "the in-library decomposed solve is Jacobi-PCG (polynomial preconditioner: stage functions)");
  BEAT_REQUIRE(max_it >= 0, "max_it must be >= 0");
  beat_ctx* ctx = pde->ctx;
  const int64_t n = pde->n, plane = pde->g.plane, fld = n + 2 * plane;
  double* r = dev_work + plane;
  double* q = r + fld;
  double* ring = q + 2 * fld;  // [r, q, z, ring...]: z is unused by the Jacobi path
  double* st = pde->d_st;
  double* h = ctx->h_pinned;
  int rc;
  const bool rr = beat_rr_available(pde);  // constant coefficients: the kernels that never store q = A p
  // ghost planes of v_ for the right-hand side (the reference's scatter_forward after the previous solve) and, in the
  // same exchange, of the guess increment e (written by the x update of the previous solve)
  const bool guess_path = (rr || pde->var) && pde->guess_order != 0 && pde->d_guess != nullptr && pde->hist_n >= 1;
  if ((rc = halo_start(comm, const_cast<double*>(dev_v_prev), n, plane, guess_path ? pde->d_guess : nullptr))) return rc;
  if ((rc = halo_wait(comm))) return rc;
  if (rr) {
    BEAT_REQUIRE(pde->have_dt, "beat_pde_set_timestep has not been called");
    BEAT_REQUIRE(n_stim >= 0 && n_stim <= BEAT_MAX_STIM, "at most %d stimuli", BEAT_MAX_STIM);
    BEAT_REQUIRE(!pde->guess_pending, "the previous solve's deferred update has not been applied");
    beat_guess_begin(pde);
    BEAT_REQUIRE(!pde->guess.use_e || guess_path, "guess increment without ghost planes");
    rc = beat_rr_rhs(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, r, st);
  } else if (pde->var) {
    BEAT_REQUIRE(pde->have_dt, "beat_pde_set_timestep has not been called");
    BEAT_REQUIRE(n_stim >= 0 && n_stim <= BEAT_MAX_STIM, "at most %d stimuli", BEAT_MAX_STIM);
    BEAT_REQUIRE(!pde->guess_pending, "the previous solve's deferred update has not been applied");
    beat_guess_begin(pde);
    BEAT_REQUIRE(!pde->guess.use_e || guess_path, "guess increment without ghost planes");
    rc = beat_var_rhs(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, r, ring, st,
                      pde->guess.use_e ? pde->guess.e : nullptr);
  } else {
    rc = beat_pde_rhs(pde, dev_v_prev, host_dev_stim_w, host_stim_amp, n_stim, dev_x, r, ring, st);
  }
  if (rc) return rc;
  if ((rc = allreduce_sum(comm, st + BB, 3))) return rc;
  if ((rc = beat_pde_cg_begin(pde, st, rtol, atol, max_it))) return rc;
  int launched = 0;
  int chunk = beat_pde_first_chunk(pde);
  double* rbuf[2] = {r, q};  // rr: the residual update writes out of place
  if (rr && (rc = halo_start(comm, rbuf[0], n, plane))) return rc;  // ghost planes of r_0
  while (true) {
    chunk = std::min(chunk, max_it - launched);
    for (int it = 0; it < chunk; ++it) {
      const int i = launched + it, slot = i % PRING;
      double* p_cur = ring + (int64_t)slot * fld;
      double* p_next = ring + (int64_t)((i + 1) % PRING) * fld;
      if (rr) {
        // p_i = D^-1 r_i + beta p_{i-1} and p_i . A p_i: the planes that need no ghost data while the ghost planes of
        // r_i travel, then the boundary planes, which also keep p_i on the ghost planes (no exchange of p)
        const double* p_old = ring + (int64_t)((i + PRING - 1) % PRING) * fld;
        double* r_cur = rbuf[i & 1];
        double* r_new = rbuf[(i + 1) & 1];
        if ((rc = beat_rr_pdot_part(pde, st, r_cur, p_old, p_cur, 0))) return rc;
        if ((rc = halo_wait(comm))) return rc;
        if ((rc = beat_rr_pdot_part(pde, st, r_cur, p_old, p_cur, 1))) return rc;
        if ((rc = allreduce_sum(comm, st + PQ, 1))) return rc;
        if ((rc = beat_rr_rupd(pde, st, r_cur, r_new, p_cur, slot, false))) return rc;  // r_{i+1}, local r.z and r.r
        if ((rc = halo_start(comm, r_new, n, plane))) return rc;  // travels behind the reductions and the next part 0
        if ((rc = allreduce_sum(comm, st + RZN, 2))) return rc;
        if (slot == PRING - 1) {
          if ((rc = beat_pde_x_flush_terms(pde, st, dev_x, ring, fld, i + 1 - PRING, 1, beat_guess_terms(pde, i + 1 - PRING))))
            return rc;
        }
        if ((rc = beat_rr_next(pde, st))) return rc;
        continue;
      }
      if ((rc = halo_start(comm, p_cur, n, plane))) return rc;                  // ghost planes of p travel ...
      if ((rc = beat_pde_spmv_dot_part(pde, p_cur, q, st, 0))) return rc;       // ... while the interior is computed
      if ((rc = halo_wait(comm))) return rc;
      if ((rc = beat_pde_spmv_dot_part(pde, p_cur, q, st, 1))) return rc;       // boundary planes + local p.q
      if ((rc = allreduce_sum(comm, st + PQ, 1))) return rc;
      if ((rc = beat_pde_cg_update_r(pde, st, r, q, slot))) return rc;
      if ((rc = allreduce_sum(comm, st + RZN, 2))) return rc;
      if (slot == PRING - 1) {
        if ((rc = beat_pde_x_flush_terms(pde, st, dev_x, ring, fld, i + 1 - PRING, 1, beat_guess_terms(pde, i + 1 - PRING))))
          return rc;
      }
      if ((rc = beat_pde_cg_next_oop(pde, st, r, p_cur, p_next))) return rc;
    }
    launched += chunk;
    BEAT_HIP_CHECK(hipMemcpyAsync(h, st, sizeof(double) * 16, hipMemcpyDeviceToHost, ctx->stream));
    BEAT_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (h[STOP] != 0.0 || launched >= max_it) break;
    chunk = 2;
  }
  if (rr) {  // the exchange started after the last residual update has no consumer: drain it before anything else
    if ((rc = halo_wait(comm))) return rc;  // touches those ghost planes
  }
  const int nupd = (int)h[NUPD], base = (nupd / PRING) * PRING;
  const GuessTerms last = beat_guess_terms(pde, base);
  beat_guess_observe(pde, (int)h[ITERS]);
  if (beat_guess_end(pde, nupd, defer_flush != 0)) {  // the last partial ring cycle and / or the guess increment
    if (defer_flush) {
      host_pending[0] = base;
      host_pending[1] = nupd % PRING;
    } else if ((rc = beat_pde_x_flush_terms(pde, st, dev_x, ring, fld, base, 0, last))) {
      return rc;
    }
  }
  const int iters = (int)h[ITERS];
  pde->last_iters = iters;
  int reason = (int)h[REASON];
  if (h[STOP] == 0.0) reason = -3;
  if (info) {
    info->iterations = iters;
    info->converged_reason = reason;
    info->residual_norm = std::sqrt(h[RR]);
    info->rhs_norm = std::sqrt(h[BB]);
  }
  if (reason < 0) {
    beat_set_error("PCG did not converge in %d iterations (||r|| = %.3e, ||b|| = %.3e)", iters, std::sqrt(h[RR]),
                   std::sqrt(h[BB]));
    return BEAT_ENOTCONV;
  }
  return BEAT_OK;
}
