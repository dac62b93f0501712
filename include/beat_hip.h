/*
 * beat_hip.h -- C ABI of libbeat_hip.so, the MI355X (gfx950) device library behind the
 * operator-split monodomain hot path of fenicsx-beat.
 *
 * The reference is pure Python and has no FFI of its own; each entry point below replaces the
 * arithmetic that one reference call site delegates to NumPy / DOLFINx / PETSc, cited as
 * file:line into the reference tree (src/beat/...).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - every function returns 0 on success, a negative BEAT_E* code otherwise; no exception
 *    crosses the ABI; beat_last_error() returns the text of the most recent failure
 *    (thread-local).
 *  - all `double*` named dev_* are DEVICE pointers owned by the caller (the Python layer
 *    allocates them as torch tensors / hipMalloc blocks); host pointers are named host_* and
 *    are only read during the call.
 *  - calls enqueue work on the context's HIP stream and return immediately unless the
 *    description says "synchronises".
 *  - an N-vector ("field") is the nodal array of one slab of a structured grid, x fastest:
 *    id = ix + nx*(iy + ny*iz_local).  Every field passed to a PDE function must be
 *    addressable one xy-plane (nx*ny doubles) BEFORE its first and AFTER its last element:
 *    those are the ghost planes of the z-slab decomposition (filled by the halo exchange on
 *    interior slab faces, ignored on physical boundaries, but they must hold finite numbers).
 *  - one host thread drives one context; contexts are not thread-safe.
 */
#ifndef BEAT_HIP_H
#define BEAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BEAT_OK 0
#define BEAT_EINVAL -1      /* bad argument (shape, null pointer, unknown model) */
#define BEAT_EHIP -2        /* a HIP runtime call failed */
#define BEAT_ENOTCONV -3    /* PCG hit max_it without meeting the tolerance */
#define BEAT_ENOMEM -4

#define BEAT_ABI_VERSION 1

/* cell-model ids (ionic step kernels) */
#define BEAT_MODEL_SIMPLE_ODE 0 /* v' = -a s, s' = b v, forward Euler; tests/test_odesolver.py:11-17 */
#define BEAT_MODEL_FHN_DEMO 1   /* demos/fitzhughnagumo.py:45-80,224-225 (10 params, states [s,V]) */
#define BEAT_MODEL_FHN_README 2 /* README.md:58-89 (11 params, states [s,v]) */
#define BEAT_MODEL_TP06_GRL1 3  /* odes/tentusscher_panfilov_2006 (.ode) + gotranx GRL1 (demos/niederer_benchmark.py:82-99) */

#define BEAT_MODEL_TORORD_DYNCL_GRL1 4 /* odes/torord/ToRORd_dynCl_endo.ode + gotranx GRL1 (demos/biv_endocardial.py:124); 45 states, 112 parameters */
#define BEAT_MODEL_TORORD_LAND_GRL1 5  /* odes/torord/ToRORd_dynCl_endo_Land.ode (the same cell + Land contraction model) + GRL1; 52 states, 140 parameters */

#define BEAT_STENCIL_POINTS 15
#define BEAT_NODE_TYPES 27
#define BEAT_MAX_STIM 8

typedef struct beat_ctx beat_ctx;
typedef struct beat_pde beat_pde;

/* KSP-like result of one linear solve; mirrors what telemetry.py:67-76 reads from PETSc
 * (getIterationNumber / getResidualNorm / getConvergedReason). */
typedef struct beat_ksp_info {
  int32_t iterations;
  int32_t converged_reason; /* >0 converged (2 = rtol, 3 = atol), 0 still iterating, -3 = max_it */
  double residual_norm;     /* ||r||_2 (global, unpreconditioned) */
  double rhs_norm;          /* ||b||_2 (global) */
} beat_ksp_info;

/* ---- library / context ------------------------------------------------------------------ */
int beat_abi_version(void);
const char* beat_last_error(void);
/* device: HIP device ordinal.  hip_stream: an existing hipStream_t (e.g. torch's current
 * stream) or NULL to use the null stream. */
int beat_ctx_create(int device, void* hip_stream, beat_ctx** out);
int beat_ctx_destroy(beat_ctx* ctx);
int beat_ctx_set_stream(beat_ctx* ctx, void* hip_stream);
int beat_ctx_synchronize(beat_ctx* ctx); /* synchronises */

/* plain device-memory helpers so the library is usable from bare ctypes (no torch) */
int beat_malloc(beat_ctx* ctx, size_t bytes, void** dev_out); /* zero-filled */
int beat_free(beat_ctx* ctx, void* dev_ptr);
int beat_memcpy_h2d(beat_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes); /* synchronises */
int beat_memcpy_d2h(beat_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes); /* synchronises */

/* ---- ionic (reaction) step: replaces  states[:] = fun(states, t, parameters, dt)
 *      src/beat/odesolver.py:67-79 ----------------------------------------------------------- */
/* A cell model the library does not ship, given as SOURCE: a C++ struct `name` with the interface csrc/beat_ode_kernel.h expects of
 * a Model (NS, NP, V_INDEX, Derived, derive, step -- what beat.models.from_ode writes from a gotran .ode file: the reference takes any
 * gotranx-generated ``fun``, demos/niederer_benchmark.py:82-99, src/beat/odesolver.py:67-79).  Returns an id >= 100 that beat_ode_step
 * (uniform vector or all per-node rows), beat_ode_step_pending, beat_ode_step_classes / beat_ode_class_table_*, beat_ode_run,
 * beat_split_steps[_big] and beat_ode_model_info accept like a built-in one; each kernel instance is compiled by hipcc at the first
 * call that needs it and cached (see beat_ode_jit_stats).  Not for it: beat_ode_step_rows (the instance compiled for a few varying
 * rows).  Registering the same name and source again returns the same id. */
int beat_ode_model_register(const char* name, const char* source, int num_states, int num_params, int v_index, int* model_id_out);
/* Static description of a built-in (or registered) cell model. */
int beat_ode_model_info(int model_id, int* num_states, int* num_params);
/* One explicit / GRL1 update of all num_states states at n nodes.
 *  dev_states : (num_states, ld) row-major ("state-major" SoA as odesolver.py:149-153), updated
 *               in place.
 *  host_params: (num_params) uniform parameters, or NULL when dev_params_per_node is given.
 *  dev_params_per_node : optional (num_params, params_ld) per-node parameters
 *               (demos/pace_train.py:133-137), else NULL.
 *  dev_v_copy : optional field; when non-NULL the updated row `v_index` is also written there
 *               (fuses ode.to_dolfin + ode_to_pde + pde.assign_previous,
 *               monodomain_solver.py:70-79). */
int beat_ode_step(beat_ctx* ctx, int model_id, double* dev_states, int64_t n, int64_t ld,
                  const double* host_params, int num_params, const double* dev_params_per_node,
                  int64_t params_ld, double t, double dt, int v_index, double* dev_v_copy);

/* The same step when the preceding diffusion solve deferred its final update of the potential
 * (beat_pde_solve_ex with defer_flush, or a stage-driven solve that skipped the last beat_pde_x_flush):
 * row v_index is read as  V + sum_{j < pending} alpha_j p_j  with p_j = dev_ring0 + j*field_stride and the step
 * lengths kept by `pde`; the new value is stored complete.  Fuses the x += sum alpha_j p_j pass of the
 * deferred-x PCG into the next ionic kernel, which is fp64-issue bound and has HBM bandwidth to spare.
 * pending = -1: `pde` has an open solve (beat_pde_solve_begin): see there. */
int beat_ode_step_pending(beat_ctx* ctx, int model_id, double* dev_states, int64_t n, int64_t ld,
                          const double* host_params, int num_params, const double* dev_params_per_node,
                          int64_t params_ld, double t, double dt, int v_index, double* dev_v_copy,
                          beat_pde* pde, const double* dev_ring0, int64_t field_stride, int pending);

/* Per-node parameters of which only a few rows vary -- a smooth gradient in one or two parameters over otherwise uniform
 * tissue; the reference hands ``fun`` the whole (P, N) array, src/beat/odesolver.py:67-79, demos/pace_train.py:133-167 --:
 * host_params = the (P,) vector of the parameters that do not vary, dev_rows = (num_rows <= 16, rows_ld) = the rows that do
 * (more than 4 only where run-time compilation is available, beat_ode_jit_stats: the shipped kernel takes four),
 * host_row_params[j] = the parameter index of row j.  Same values as beat_ode_step with all P rows on the device (the
 * kernel builds the node's parameter set from both and runs the per-node step); 8 num_rows bytes per node are read
 * instead of 8 P.  pde / dev_ring0 / field_stride / pending as in beat_ode_step_pending (pde = NULL: a plain step). */
int beat_ode_step_rows(beat_ctx* ctx, int model_id, double* dev_states, int64_t n, int64_t ld,
                       const double* host_params, int num_params, const int* host_row_params, int num_rows,
                       const double* dev_rows, int64_t rows_ld, double t, double dt, int v_index, double* dev_v_copy,
                       beat_pde* pde, const double* dev_ring0, int64_t field_stride, int pending);
/* Run-time compilation behind beat_ode_step_rows: the kernel instance whose varying parameter indices are compile-time
 * constants is written, compiled by hipcc for gfx950 and cached at first use (csrc/beat_ode_jit.h; environment: BEAT_JIT=0 off,
 * BEAT_JIT_CACHE directory, BEAT_JIT_SRC kernel sources, BEAT_HIPCC compiler, BEAT_JIT_TIMEOUT_S, BEAT_JIT_VERBOSE).  host_out[4] = kernels loaded
 * in this process, hipcc runs, code objects taken from the cache directory, failures; returns 1 where it is usable (sources,
 * compiler, cache directory found), 0 where beat_ode_step_rows runs its run-time-index kernel instead. */
int beat_ode_jit_stats(long long* host_out);

/* Cell types / parameter classes in ONE launch (src/beat/odesolver.py:306-310 loops over the markers and calls `fun`
 * once per marker; demos/biv_endocardial.py:187-282: endo / mid / epi): one (S, n) state array, a byte per node that
 * names the node's class (0 .. classes-1; 255: the node belongs to none and is not advanced -- its potential still
 * receives a pending update; 254: a padding entry, nothing is read or written for it), and a device table with
 * one uniform parameter set per class.  beat_ode_class_table_doubles gives the size of a table entry (the parameters
 * followed by the model's per-launch constants), beat_ode_class_table_fill builds `classes` entries from
 * host_params (classes x num_params, row-major) into dev_table (synchronises).  beat_ode_step_classes is
 * beat_ode_step_pending with the class table in place of the parameters: wavefronts whose nodes share a class read
 * their set with scalar loads, as the uniform kernel does; a wavefront on a class boundary runs the step once per class
 * present.  Piecewise-constant per-node parameters (demos/pace_train.py:133-167: two conductances zeroed in half of the
 * cable) are the same thing: the distinct parameter columns are the classes.
 * dev_node_map / dev_v_field (both or neither): the state array holds only the n nodes that carry a cell model (the
 * wall of a voxelised geometry inside its box -- an ionic kernel is bound by fp64 issue, so idle lanes cost what busy ones
 * do); entry i belongs to node dev_node_map[i] of the PDE grid, the potential is READ from dev_v_field there (with the
 * pending update of that field applied) and the new one written to the state array's row and to the field: the
 * reference's to_dolfin / from_dolfin for every marker (odesolver.py:280-292) inside the one launch. */
#define BEAT_MAX_CLASSES 32
int beat_ode_class_table_doubles(int model_id, int* doubles_per_class);
int beat_ode_class_table_fill(beat_ctx* ctx, int model_id, const double* host_params, int num_params, int classes,
                              double* dev_table);
int beat_ode_step_classes(beat_ctx* ctx, int model_id, double* dev_states, int64_t n, int64_t ld,
                          const double* dev_table, int classes, const unsigned char* dev_markers, double t, double dt,
                          int v_index, double* dev_v_copy, const int* dev_node_map, double* dev_v_field,
                          beat_pde* pde, const double* dev_ring0, int64_t field_stride, int pending);

/* nbeats x nsteps updates in ONE launch with the node's states held in registers: replaces the Python
 * loops of src/beat/single_cell.py:42-65 (solve_with_save / solve_without_save; t restarts at t0 for every
 * beat and is t0 + j*dt within it) and of src/beat/odesolver.py:24-43.  Optionally records `ntrack` (<= 8)
 * states every `save_freq` steps into dev_trace, shape (rows, ntrack, n), rows = nbeats*ceil(nsteps/save_freq). */
int beat_ode_run(beat_ctx* ctx, int model_id, double* dev_states, int64_t n, int64_t ld,
                 const double* host_params, int num_params, const double* dev_params_per_node,
                 int64_t params_ld, double t0, double dt, int64_t nsteps, int nbeats, int save_freq,
                 const int* host_track_idx, int ntrack, double* dev_trace);

/* row/field transfers: v_ode.x.array[:] = values[v_index] etc. (odesolver.py:164-170,
 * utils.py:52-54, monodomain_model.py:59-60) */
int beat_copy(beat_ctx* ctx, double* dev_dst, const double* dev_src, int64_t n);
int beat_fill(beat_ctx* ctx, double* dev_dst, double value, int64_t n);
/* Streaming probe: ONE launch of a kernel that only moves bytes over dev[0, n) (n even, 16-byte aligned), so that the achieved
 * rate of the hot kernels can be set against what the memory system gives THIS library's own loads and stores (bench.py's
 * roofline.inplace_stream, tools/stream_probe.py).  The reference publishes no throughput (its demos print wall times,
 * demos/external_operator_gotranx.py:187-256); this is measurement infrastructure, not part of the step.
 *  mode  : 0 in place x = 1.0 * x (values unchanged), 1 read only, 2 write only (FILLS dev with 0), 3 copy first half -> second
 *          half, 4 in place over `rows` rows of a (rows, ld) array with all loads of an index ahead of its stores (the ionic
 *          kernels' access pattern; rows in {1, 4, 8, 19, 45}, n = row length, ld even)
 *  policy: bit 0 non-temporal loads, bit 1 non-temporal stores, bit 2 raw-buffer instructions instead of global ones (modes 0-3)
 *  unroll: 1, 2 or 4 independent 16-byte accesses in flight per lane (modes 0-3)
 *  blocks: workgroups of 256 threads walking the array with a grid-stride loop; 0 = as many as the array has chunks */
int beat_stream_probe(beat_ctx* ctx, double* dev, int64_t n, int mode, int policy, int unroll, int blocks, int rows,
                      int64_t ld);
/* dst[i] = src[idx[i]] (gather) / dst[idx[i]] = src[i] (scatter): marker-wise state transfer of
 * DolfinMultiODESolver (odesolver.py:280-292). */
int beat_gather(beat_ctx* ctx, double* dev_dst, const double* dev_src, const int64_t* dev_idx, int64_t n);
int beat_scatter(beat_ctx* ctx, double* dev_dst, const double* dev_src, const int64_t* dev_idx, int64_t n);

/* dst[j] = w[2j] src[idx[2j]] + w[2j+1] src[idx[2j+1]]: interpolation between the PDE's P1 space and an ODE space
 * with other degrees of freedom (P2: vertices and edge midpoints, DG1: vertices per cell) -- utils.local_project,
 * utils.py:26-58, as used by ode_to_pde / pde_to_ode (odesolver.py:101-115). */
int beat_interp2(beat_ctx* ctx, double* dev_dst, const double* dev_src, const int64_t* dev_idx, const double* dev_w,
                 int64_t n);

/* ---- diffusion (PDE) step: replaces LinearProblem assembly + KSP solve
 *      src/beat/base_model.py:114-124,188-245 ---------------------------------------------- */
/* Structured-grid operator of one z-slab.
 *  n[3]        : local node counts (nx, ny, nz_local); unused axes = 1.
 *  z_lo_phys / z_hi_phys : 1 if the slab's low / high z face is a physical (Neumann)
 *               boundary, 0 if another slab continues there (ghost plane is then live).
 *  host_mass_tab / host_stiff_tab : (27, 15) coefficient tables of the P1 consistent mass
 *               matrix and of K = int M grad phi_j . grad phi_i for the 27 node types
 *               (type = tx + 3 ty + 9 tz, t = 0 low face, 1 interior, 2 high face); column k
 *               multiplies the node at offset beat_stencil_offsets()[k].  They are derived by
 *               literal element assembly on DOLFINx's 6-tet (2-triangle) subdivision
 *               (monodomain_model.py:68-98). */
int beat_pde_create(beat_ctx* ctx, const int64_t n[3], int z_lo_phys, int z_hi_phys,
                    const double* host_mass_tab, const double* host_stiff_tab, beat_pde** out);
/* Same operators with per-node coefficients: voxel-masked domains (the reference's ventricular meshes,
 * demos/biv_endocardial.py, voxelised onto the structured grid) and spatially varying conductivity
 * M(x) = s_l f0 f0^T + ... from a fibre field (conductivities.py:107-118, monodomain_model.py:68-98).
 *  dev_mass / dev_stiff : DEVICE arrays (15, ld), coefficient-major: entry [k*ld + i] multiplies the node
 *               at offset beat_stencil_offsets()[k] in row i.  Borrowed for the lifetime of the handle.
 *               Rows must not couple to nodes outside the box; a row whose mass diagonal is 0 (node not
 *               touched by any active element) is treated as an identity row by every operator.
 * All other beat_pde_* entry points work unchanged on such a handle (the polynomial preconditioner is
 * Jacobi-only there). */
/* Device-side assembly of such rows from per-voxel data: every voxel (box cell) is split into simplices as in
 * geometry.py:121-139 and carries a conductivity tensor and an in/out flag.
 *  n[3]     : local nodes of the slab;  cells[3]: GLOBAL voxels per axis (1 for unused axes);
 *  z0       : global plane index of the slab's first plane
 *  host_T   : [8][8][9] element tensor, K_e[a][b] = sum_ij T[a][b][3i+j] M_ij (corner k at offsets
 *             (k&1, (k>>1)&1, (k>>2)&1));  host_Me: [8][8] element mass matrix of one voxel
 *  dev_M    : DEVICE (nvoxels, 9) row-major tensors, or NULL to use the constant host_M_const[9]
 *  dev_active : DEVICE (nvoxels) bytes, 0 = outside the tissue, or NULL (all inside)
 *  dev_mass / dev_stiff : DEVICE (15, ld) outputs.  Synchronises. */
int beat_pde_assemble_rows(beat_ctx* ctx, const int64_t n[3], const int64_t cells[3], int64_t z0,
                           const double* host_T, const double* host_Me, const double* dev_M,
                           const double* host_M_const, const unsigned char* dev_active,
                           double* dev_mass, double* dev_stiff, int64_t ld);
/* Dirichlet conditions on per-node rows by symmetric elimination (dolfinx.fem.dirichletbc in the Laplace
 * problems of utils.py:115-355, expand_layer / expand_layer_biv): rows of flagged nodes become identity rows,
 * couplings of the other rows to flagged nodes are zeroed and moved to the right-hand side
 *   dev_f[i] = dev_g[i] (flagged)  |  -sum_{j flagged} K_ij dev_g[j] (free).
 * dev_rows (15, ld) is modified in place; dev_flag holds one byte per node.  Single slab (no ghost planes). */
int beat_rows_apply_dirichlet(beat_ctx* ctx, const int64_t n[3], double* dev_rows, int64_t ld,
                              const unsigned char* dev_flag, const double* dev_g, double* dev_f);
int beat_pde_create_var(beat_ctx* ctx, const int64_t n[3], int z_lo_phys, int z_hi_phys,
                        const double* dev_mass, const double* dev_stiff, int64_t ld, beat_pde** out);
int beat_pde_destroy(beat_pde* pde);
/* Node type along z (0 = low face of the whole grid, 1 = interior, 2 = high face) of the two ghost planes of a slab
 * whose z faces are not physical: 1 unless the neighbouring rank owns a single plane that is itself a face of the
 * grid.  Needed by the decomposed constant-coefficient solve, which forms the search direction on the ghost planes
 * itself (D^-1 there depends on the type).  Default (1, 1). */
int beat_pde_set_ghost_types(beat_pde* pde, int ghost_lo_type, int ghost_hi_type);
/* (dx,dy,dz) of the 15 stencil points, 45 ints. */
const int* beat_stencil_offsets(void);
/* A = C_m*Mass + theta*dt*K ; B = C_m*Mass - (1-theta)*dt*K  (base_model.py:188-194, called
 * when dt changes, :225-230). */
int beat_pde_set_timestep(beat_pde* pde, double C_m, double theta, double dt);
/* y = A x  /  y = B x  /  y = Mass x  /  y = K x   (which = 0,1,2,3); ghost planes of x must be
 * current.  Returns nothing else; used by tests and by the distributed PCG. */
int beat_pde_apply(beat_pde* pde, int which, const double* dev_x, double* dev_y);

/* --- Jacobi-PCG, expressed as stage functions so that the same kernels serve the single-slab
 * solve (beat_pde_solve) and the slab-decomposed solve, where the caller all-reduces the marked
 * slots of `dev_st` across ranks (RCCL) and exchanges ghost planes between the stages.
 *
 * dev_st: caller-owned device array of >= 16 doubles, the scalar state of one solve:
 *   [0] b.b  [1] r.z  [2] r.r  [3] p.q  [4] r.z (new)  [5] r.r (new)  [6] tol^2  [7] beta
 *   [8] stop latch (0/1)  [9] iterations  [10] converged reason  [11] rtol  [12] atol  [13] max_it
 * Once the latch [8] is set every stage function becomes a no-op, so a caller may enqueue more
 * iterations than needed without synchronising and read [8..10] back later.
 *
 * Right-hand side build (base_model.py:196-206 + _G_stim :247-248), in residual form with
 * initial guess x0 = v_:
 *     b  = B v_ + dt * sum_k amp[k] * w_k          (never stored; only b.b is reduced)
 *     r  = b - A v_ = dt * ( -K v_ + sum_k amp[k] w_k )
 *     x  = v_ (skipped when dev_x == dev_v_prev),  p = z = D^-1 r
 * dev_stim_w[k] are the nodal weight fields int_{dz_k} phi_i (NULL entries / amp 0 skipped).
 * Writes the LOCAL sums b.b, r.z, r.r to dev_st[0..2]  -> all-reduce dev_st[0:3]. */
int beat_pde_rhs(beat_pde* pde, const double* dev_v_prev, const double* const* host_dev_stim_w,
                 const double* host_stim_amp, int n_stim, double* dev_x, double* dev_r,
                 double* dev_p, double* dev_st);
/* tol^2 = max(rtol^2 b.b, atol^2); latch if r.r <= tol^2 (PETSc-style ||r|| <= max(rtol ||b||, atol)). */
int beat_pde_cg_begin(beat_pde* pde, double* dev_st, double rtol, double atol, int max_it);
/* q = A p (ghost planes of p must be current); LOCAL p.q -> dev_st[3]  -> all-reduce dev_st[3:4]. */
int beat_pde_spmv_dot(beat_pde* pde, const double* dev_p, double* dev_q, double* dev_st);
/* The same SpMV in two parts, to overlap the halo exchange of p with compute on a decomposed grid:
 * part 0 computes the planes that need no ghost data (enqueue it while the ghost planes travel),
 * part 1 the one or two slab-boundary planes, then reduces LOCAL p.q -> dev_st[3]. */
int beat_pde_spmv_dot_part(beat_pde* pde, const double* dev_p, double* dev_q, double* dev_st, int part);
/* alpha = st[1]/st[3]; x += alpha p; r -= alpha q; LOCAL r.D^-1 r, r.r -> dev_st[4..5]
 * -> all-reduce dev_st[4:6]. */
int beat_pde_cg_update(beat_pde* pde, double* dev_st, double* dev_x, double* dev_r,
                       const double* dev_p, const double* dev_q);
/* beta = st[4]/st[1]; roll the scalars, count the iteration, set the latch on convergence or
 * max_it; then p = D^-1 r + beta p   -> exchange ghost planes of p. */
int beat_pde_cg_next(beat_pde* pde, double* dev_st, const double* dev_r, double* dev_p);

/* Deferred-x variant of the update stages (what beat_pde_solve uses internally, exposed for the
 * slab-decomposed solve): search directions live in a ring of beat_pde_ring_size() fields, iteration i
 * uses p_i = ring[i % size]; x is brought up to date by beat_pde_x_flush when the ring is full
 * (only_if_full = 1, enqueue it right after the update of iteration i with i % size == size-1, before
 * that slot is overwritten) and once more after the solve for the partially filled last cycle. */
/* beat_pde_ring_size(): the default ring (6).  A per-node-row operator on a single slab keeps 12 directions (its solves take 9-12
 * iterations at the reference's dt = 0.05 ms, demos/biv_endocardial.py:137; with 6 every solve paid an in-loop flush);
 * beat_pde_work_fields(pde) - 3 is an operator's own ring, and the number of ring fields its dev_work holds. */
int beat_pde_ring_size(void);
/* alpha = st[1]/st[3] (remembered for slot); r -= alpha q; LOCAL r.D^-1 r, r.r -> dev_st[4..5]
 * -> all-reduce dev_st[4:6].  dev_st[14] counts the executed updates. */
int beat_pde_cg_update_r(beat_pde* pde, double* dev_st, double* dev_r, const double* dev_q, int slot);
/* scalar roll as beat_pde_cg_next, then p_next = D^-1 r + beta p_cur (out of place). */
int beat_pde_cg_next_oop(beat_pde* pde, double* dev_st, const double* dev_r, const double* dev_p_cur,
                         double* dev_p_next);
int beat_pde_x_flush(beat_pde* pde, const double* dev_st, double* dev_x, const double* dev_ring0,
                     int64_t field_stride, int ring_base, int only_if_full);

/* Optional polynomial (Chebyshev-Jacobi) preconditioner  z = sum_k c_k (D^-1 A)^k D^-1 r,
 * k < ncoef <= 8.  ncoef = 1 is plain Jacobi (the default).  On this bandwidth-bound path a PCG
 * iteration moves 72 B/node in vector updates but only 16-24 B/node per stencil pass, so trading
 * iterations for a few extra stencil passes lowers the bytes per solve.  The coefficients come from
 * the caller (beat/_engine.py derives them from Gershgorin bounds on D^-1 A). */
int beat_pde_set_preconditioner(beat_pde* pde, int ncoef, const double* host_coef);
int beat_pde_pc_num_passes(beat_pde* pde); /* ncoef - 1 */
/* Horner pass j of the preconditioner: reads r (j = 0, ghost planes of r current) or the previous
 * output (ghost planes of that field current) and writes dev_q / dev_z alternately such that the
 * last pass writes dev_z; the last pass also stores the LOCAL r.z in *dev_red (dev_st+1 for the
 * initial residual, dev_st+4 inside the iteration, where it replaces the Jacobi value written by
 * beat_pde_cg_update). */
int beat_pde_pc_pass(beat_pde* pde, int j, const double* dev_r, double* dev_z, double* dev_q,
                     double* dev_st, double* dev_red);
/* p = z (first search direction, beta = 0 after beat_pde_cg_begin). */
int beat_pde_cg_first_z(beat_pde* pde, double* dev_st, const double* dev_z, double* dev_p);
/* as beat_pde_cg_next with p = z + beta p. */
int beat_pde_cg_next_z(beat_pde* pde, double* dev_st, const double* dev_z, double* dev_p);

/* Whole single-slab step: rhs build + Jacobi-PCG to ||r|| <= max(rtol*||b||, atol), with all
 * scalars kept on the device (one host synchronisation at the end to fill `info`).
 * Replaces _update_rhs + KSP.solve of base_model.py:232-236 when the grid is not decomposed.
 * The Jacobi path defers the update of x: search directions are kept in a ring of 6 fields (the
 * p-update writes out of place) and x += sum_i alpha_i p_i is applied once per solve, which removes
 * 24 B/node of traffic per iteration.
 * dev_work: beat_pde_work_fields() fields (r, q, z, ring[6]) laid out beat_pde_field_stride() doubles apart, each with
 * its own ghost planes: size fields * beat_pde_field_stride(pde) doubles. Synchronises. */
int beat_pde_work_fields(beat_pde* pde);
/* Distance in doubles between consecutive fields of dev_work (and the `field_stride` of the deferred-update calls when the
 * ring lives there): n_local + 2*nx*ny, plus BEAT_FIELD_SKEW doubles of padding if that environment variable is set
 * (an experiment against memory-channel aliasing of fields a multiple of 4 KiB apart; measured: no effect, default none).
 * The PETSc Vecs of the reference (src/beat/base_model.py:107-124) carry no such layout; it is this library's. */
int64_t beat_pde_field_stride(const beat_pde* pde);
/* beat_pde_solve with the option to leave the last, partially filled ring cycle of search directions unapplied:
 * with defer_flush != 0, host_pending[0] = first iteration of that cycle (ring_base for beat_pde_x_flush) and
 * host_pending[1] = number of directions still to be added to dev_x (0: dev_x is complete).  The caller must
 * consume them before the next solve: beat_ode_step_pending, or beat_pde_x_flush(pde, NULL, dev_x, ring, stride,
 * host_pending[0], 0) where dev_st = NULL selects the handle's own scalar state and ring = dev_work + plane +
 * 3*(n + 2*plane) is the first search direction of the work array. */
/* Initial guess from the previous solves (what PETSc offers as KSPGuess / -ksp_guess_type; the reference leaves
 * it off and starts from 0, base_model.py:141-151).  With order m in 1..4 the operator keeps the increments
 * d = x - v_ of the last solves and starts the next one from x0 = v_ + e, e = the polynomial extrapolation of degree
 * m - 1 through the last m increments (m = 1: d1; 2: 2 d1 - d2; 3: 3 d1 - 3 d2 + d3; 4: 4 d1 - 6 d2 + 4 d3 - d4; fewer
 * while fewer are on record), instead of x0 = v_: the same system is solved to the same ||r|| <= rtol ||b||, in
 * fewer iterations when consecutive solves are consecutive time steps (512^3 TP06, rtol 1e-8: 4.95 -> 1.7 per step
 * behind the initial perturbation, 8.0 -> 3.75 on a travelling front with m = 3).  Supported by
 * beat_pde_solve[_ex] and beat_pde_solve_dist with Jacobi on the register-row and the per-node-row kernels; the stage
 * functions, the LDS-tiled constant-coefficient loop and the polynomial preconditioner always start from x0 = v_ and
 * drop the history.  The guess costs no pass of its own: it is part of the deferred update (x += e + sum alpha_j
 * p_j), which also records the new increment and prepares the next e in place.  Consequences for callers that
 * defer (defer_flush != 0): the update may be due even when host_pending[1] == 0 -- ask beat_pde_guess_pending --
 * and it must be applied through this operator (beat_ode_step_pending with `pde`, or beat_pde_x_flush), which clears
 * the flag.  order = -1 chooses m per solve by hill climbing over 1..4: it keeps a running mean of the iteration
 * counts each order has been costing, uses the current one, tries one of its neighbours every 12th solve and moves
 * when the neighbour is cheaper (the cubic wins on a travelling front, m = 1 once the increments are smooth and the
 * rtol-sized noise they carry is all an extrapolation amplifies; iteration counts are global, so all ranks of a
 * decomposed solve decide alike).  beat_pde_set_timestep and beat_pde_guess_reset drop the history (call the latter when the potential is
 * overwritten between steps; a stale history costs iterations, not accuracy).  Memory: max(m - 1, 1) + 1 more fields,
 * allocated when the order is set.  Default order: 0 (the Python layer's BaseModel asks for -1). */
int beat_pde_set_guess_order(beat_pde* pde, int order);
/* PETSc's -ksp_cg_single_reduction (KSPCGUseSingleReduction) for beat_pde_solve_dist: on = 1 runs the PCG iteration with
 * ONE all-reduce of three values (u.Au, r.u, r.r with u = D^-1 r; step lengths by Chronopoulos & Gear's recurrence)
 * instead of two dependent ones; 0 the classic iteration; -1 (default) as the environment says (BEAT_DIST_MERGED=1).
 * Constant-coefficient operators only (per-node rows keep the classic iteration); beat_pde_solve on one rank is not
 * affected.  Same stopping test, iteration counts within one, k + 2 instead of 2 k + 1 all-reduces per solve of k iterations. */
int beat_pde_set_single_reduction(beat_pde* pde, int on);
/* 1 when the decomposed solves of this per-node-row operator run the fused tile pass (direction formed while loading, on the
 * ghost planes too: r is exchanged, p never) -- every rank of the decomposition has to be able to (tiles of 8 rows, a tile list;
 * agreed by a sum over the ranks when the operator changes; BEAT_VTL_PDOT_DIST=0 switches it off) -- 0 for the three-kernel
 * iteration with an exchange of p, and before the first decomposed solve. */
int beat_pde_fused_dist_pass(const beat_pde* pde);
/* Which kernels a per-node-row operator on an undivided grid solves with (tests assert the route they mean to check): bit 0 the
 * workgroup-tile product (beat_pde_spmv_dot and the solves), bit 1 its fused pass (direction formed while loading), bit 2 the
 * right-hand side on the tiles, bit 3 the ring of 12 search directions.  0 for a constant-coefficient operator. */
int beat_pde_tile_route(const beat_pde* pde);
int beat_pde_guess_reset(beat_pde* pde);
int beat_pde_guess_pending(const beat_pde* pde);
/* the last recorded increment, the guess increment prepared for the next solve, and the number of solves on record
 * since the history was dropped, capped at 4 (tests, checkpoints) */
int beat_pde_guess_history(const beat_pde* pde, double** dev_d, double** dev_e, int* count);
/* What the x update of the last solve (the pending one, if it was deferred to the next ionic kernel) moves for the
 * guess's bookkeeping: host_out[4] = {fields it reads (e, the increments its extrapolation uses), fields it writes
 * (d, e: 2, or 0 without a guess), the order in use (the adaptive policy's current one for order -1), 1 if that
 * update is still pending, 2 if it was applied by a launch enqueued behind the open solve (pending = -1)}.  8 bytes per node each:
 * what bench.py charges the ionic kernel for. */
int beat_pde_guess_traffic(const beat_pde* pde, int* host_out);

/* Small grids (constant coefficients, Jacobi, undivided, <= 8192 nodes -- the reference's own CPU-sized cases, e.g.
 * the Niederer slab at dx = 0.5 mm): beat_pde_solve[_ex] runs the whole solve -- right-hand side, every iteration,
 * update of dev_x, bookkeeping of the initial guess -- in one launch of one workgroup (search direction in LDS, dot
 * products as block reductions), because a multi-launch iteration there is nothing but launch latency
 * (demos/niederer_benchmark.py: 0.38 -> 0.15 ms per split step).  Same iteration and stopping test, different
 * summation order in the dot products; nothing is ever left pending (host_pending = {0, 0}).  On by default
 * (BEAT_SMALL=0 disables it for the process), per operator: */
int beat_pde_set_small_grid_solve(beat_pde* pde, int enable);
int beat_pde_small_grid_solve_active(const beat_pde* pde);

/* n_steps whole split steps of such a grid in ONE call (src/beat/monodomain_solver.py:33-79 with theta = 1, n_steps
 * times: the ionic step on the (S, ld) state array, then the diffusion step in place on row
 * v_index; uniform parameters).  Per step the library enqueues the ionic kernel, the one-launch solve and -- with
 * n_probe > 0 -- one row of point values (beat_field_probe_record: dev_probe_out is (n_steps, n_probe)); the host is
 * not consulted between steps and synchronises once at the end to fill host_info[0..n_steps) (NULL: not wanted).
 * host_t0[s], host_dt[s]: start time and length of step s as the ionic step is given them (t1 - t0 of the caller's
 * loop, which may differ from step to step in the last bit; the diffusion step uses the operator's dt,
 * beat_pde_set_timestep); host_stim_amp is (n_steps, n_stim): the amplitudes at host_t0[s] + theta_pde dt, which the caller knows in advance
 * (base_model.py:196-201 evaluates the stimulus expression at that time).  The initial guess of each solve follows
 * beat_pde_set_guess_order; the adaptive order keeps its choice through a batch and is
 * re-decided between batches from the batch's mean iteration count.  Returns BEAT_ENOTCONV if a
 * solve of the batch ran out of iterations (the later steps have run on its last iterate, as the reference's loop
 * would without ksp_error_if_not_converged).  At most BEAT_MAX_BATCH steps per call. */
#define BEAT_MAX_BATCH 1024
int beat_split_steps(beat_ctx* ctx, int model_id, double* dev_states, int64_t n, int64_t ld,
                     const double* host_params, int num_params, int v_index, beat_pde* pde, int n_steps,
                     const double* host_t0, const double* host_dt, const double* const* host_dev_stim_w, const double* host_stim_amp, int n_stim,
                     double rtol, double atol, int max_it, const int64_t* host_probe_idx, const double* host_probe_w,
                     int n_probe, double* dev_probe_out, beat_ksp_info* host_info);
/* The same for a grid of any size on one rank (theta = 1): per step one beat_ode_step_pending (applying what the previous solve
 * deferred) and one beat_pde_solve_ex(defer_flush = 1) in place on the potential row -- the loop a caller would write
 * (src/beat/monodomain_solver.py:53-66 calling :33-79), run inside the library so that nothing but the wake-up of the convergence
 * check and a launch lies between two steps (through Python: 0.15-0.18 ms per step of a 512^3 grid with the device idle).
 * dev_work: as for beat_pde_solve; pending_in: search directions of an earlier deferred solve still to be applied to the row
 * (host_pending[1] of that solve, 0 for none); host_pending[3]: [0], [1] what the last solve that ran left pending (apply it with
 * the next ionic launch or beat_pde_x_flush), [2] the number of steps done; host_info[n_steps]; host_ode_ms[n_steps] or NULL:
 * duration of every ionic launch (-1 for steps not run).
 * A solve that runs out of iterations ENDS the batch: BEAT_ENOTCONV is returned with host_pending[2] = the failing step + 1 and
 * the state as that step left it -- the caller decides (PETSc's ksp_error_if_not_converged raises there; the reference's loop
 * without it goes on, src/beat/base_model.py:236-239) and calls again for the remaining steps. */
int beat_split_steps_big(beat_ctx* ctx, int model_id, double* dev_states, int64_t n, int64_t ld, const double* host_params,
                         int num_params, int v_index, beat_pde* pde, double* dev_work, int n_steps, const double* host_t0,
                         const double* host_dt, const double* const* host_dev_stim_w, const double* host_stim_amp, int n_stim,
                         double rtol, double atol, int max_it, int pending_in, beat_ksp_info* host_info, int* host_pending,
                         float* host_ode_ms);

/* beat_pde_solve_ex(defer_flush = 1) in two halves (round 5): _begin enqueues the right-hand side, as many iterations as the
 * previous solve needed + 1 and the copy of the solve's scalar state, and returns WITHOUT waiting; _end waits for that copy,
 * enqueues more iterations if the solve has not latched (and waits again), does the host's bookkeeping and reports as
 * beat_pde_solve_ex does (info, host_pending; BEAT_ENOTCONV).  Between the two the caller may enqueue the next ionic step behind
 * the open solve -- beat_ode_step_pending / _rows / _classes with pending = -1: the kernel reads what is pending from the solve's
 * state on the device and does nothing if the solve has not latched; the call then finishes the solve itself (as _end would) and
 * repeats the launch if it has to -- so that the device never idles between the solve's last kernel and the ionic kernel while the
 * host wakes up (MonodomainSplittingSolver.step against .solve: 13.71 against 13.46 ms per 512^3 step, BENCH_r04).  PETSc's KSPSolve
 * (src/beat/base_model.py:236) returns when the solve is done; what the reference's caller does with the result -- nothing, unless a
 * monitor is attached (base_model.py:239) -- can wait a step.  Jacobi on one slab only (beat_pde_solve_is_open says whether a
 * solve is open; _end without one returns the last finished solve's record).  Every other entry point that takes the operator
 * refuses while a solve is open. */
int beat_pde_solve_begin(beat_pde* pde, const double* dev_v_prev, const double* const* host_dev_stim_w,
                         const double* host_stim_amp, int n_stim, double* dev_x, double* dev_work, double rtol, double atol, int max_it);
int beat_pde_solve_end(beat_pde* pde, beat_ksp_info* info, int* host_pending);
int beat_pde_solve_is_open(const beat_pde* pde);
int beat_pde_solve_can_open(const beat_pde* pde); /* 1: beat_pde_solve_begin takes this operator (Jacobi, one slab, not the one-launch path) */
int beat_pde_solve_ex(beat_pde* pde, const double* dev_v_prev, const double* const* host_dev_stim_w,
                      const double* host_stim_amp, int n_stim, double* dev_x, double* dev_work, double rtol,
                      double atol, int max_it, int defer_flush, beat_ksp_info* info, int* host_pending);
int beat_pde_solve(beat_pde* pde, const double* dev_v_prev, const double* const* host_dev_stim_w,
                   const double* host_stim_amp, int n_stim, double* dev_x, double* dev_work,
                   double rtol, double atol, int max_it, beat_ksp_info* info);

/* ---- slab-decomposed diffusion solve: one C call per solve, communication inside the library -------------
 * Replaces, on a grid cut into z-slabs (one rank per GPU), what PETSc does inside KSP.solve on a partitioned
 * mesh: the neighbour exchange of ghost values (b.ghostUpdate / VecScatter, src/beat/base_model.py:203-206,242)
 * and the all-reduces of the dot products (KSP.solve, base_model.py:236).
 *
 * A beat_comm is the transport of one rank:
 *  - RCCL (the product; RCCL = the NCCL API over xGMI): two communicators, one for the ghost-plane
 *    send/recv pairs -- issued on a library-owned, non-blocking side HIP stream and ordered against the compute
 *    stream with events, so that they overlap with the stencil on the interior planes -- and one for the
 *    all-reduces, which sit on the critical path and are issued on the context's compute stream.  librccl is
 *    opened with dlopen when the first communicator is created (the library itself has no link-time dependency
 *    on it).  The caller distributes the 128-byte unique id of rank 0 (beat_comm_unique_id) to all ranks by
 *    whatever means it has (torch.distributed, MPI, a file).
 *  - callbacks (tests / rehearsal on hardware where RCCL cannot run, e.g. several ranks sharing one GPU): the
 *    two operations are delegated to the caller; they must be complete (stream-ordered on the context's stream)
 *    when the callback returns.
 * peer_lo / peer_hi: rank owning the slab below / above this one, -1 on a physical boundary. */
typedef struct beat_comm beat_comm;
#define BEAT_UNIQUE_ID_BYTES 128
int beat_comm_unique_id(void* host_id_out /* 2 * BEAT_UNIQUE_ID_BYTES: ids of the two communicators */);
int beat_comm_create_rccl(beat_ctx* ctx, int rank, int world, int peer_lo, int peer_hi, const void* host_id,
                          beat_comm** out);
/* flags = BEAT_COMM_SERIAL: ONE communicator carries ghost planes and all-reduces, everything on the compute stream
 * -- no overlap, but every rank enqueues the same RCCL operations in the same order on one stream, the ordering
 * that cannot deadlock whatever the GPUs' schedulers do with two concurrent RCCL kernels (bench.py's launcher
 * falls back to it when a run with the two-communicator transport stops making progress). */
#define BEAT_COMM_SERIAL 1
int beat_comm_create_rccl_ex(beat_ctx* ctx, int rank, int world, int peer_lo, int peer_hi, const void* host_id,
                             int flags, beat_comm** out);
/* Exchange the slab-boundary planes of one field: the first interior plane goes to peer_lo and the plane
 * received from it lands in the ghost plane below the field; likewise the last plane / upper ghost plane with
 * peer_hi.  dev_* point at the four planes (each plane_doubles long); a side without a peer passes NULL. */
typedef int (*beat_halo_fn)(void* user, const double* dev_first, double* dev_ghost_lo, const double* dev_last,
                            double* dev_ghost_hi, int64_t plane_doubles);
typedef int (*beat_allreduce_fn)(void* user, double* dev_values, int count); /* in-place sum over the ranks */
int beat_comm_create_callbacks(beat_ctx* ctx, int rank, int world, int peer_lo, int peer_hi, beat_halo_fn halo,
                               beat_allreduce_fn allreduce, void* user, beat_comm** out);
/*  - ipc: ghost planes as device-to-device copies between processes, without RCCL: every rank owns a mailbox in
 *    fine-grained device memory that its two neighbours map (hipIpcGetMemHandle / hipIpcOpenMemHandle); a transfer
 *    kernel on the side stream copies the boundary planes into the neighbour's mailbox and raises a sequence flag
 *    there, the neighbour's transfer kernel spins (bounded) on that flag and moves the planes into its ghost planes --
 *    ordering is done on the device, the hosts never wait for each other.  Over xGMI between the GPUs of one node,
 *    inside one GPU when several ranks share it (where RCCL, one rank per device, cannot run: that is how the overlap
 *    of the exchange with the interior stencil is measured on a one-GPU box).  The all-reduces go through RCCL
 *    (host_rccl_id = the 2 x 128-byte id of beat_comm_unique_id; its second communicator is created), through the
 *    caller (allreduce / user), or -- both NULL, at most BEAT_IPC_MAX_RANKS ranks -- through the mailboxes as well:
 *    one small kernel per all-reduce stores the rank's 1-4 values and a sequence flag into every rank's mailbox, waits
 *    (bounded) for every rank's flag in its own and adds the values in rank order, so all ranks hold the same bits; no
 *    RCCL at all then.  Set-up is two steps: every rank creates its side and gets BEAT_IPC_HANDLE_BYTES to publish;
 *    once it holds its neighbours' handles it connects (beat_comm_ipc_connect: pass NULL for an absent neighbour; a
 *    rank that is its own neighbour needs none) -- or, for the mailbox all-reduce, the handles of ALL ranks in rank
 *    order, world x BEAT_IPC_HANDLE_BYTES (beat_comm_ipc_connect_all; the rank's own entry is not opened).  All ranks
 *    must issue their exchanges and all-reduces in the same order (the decomposed solve does).  A rank that stops
 *    responding is reported (BEAT_IPC_TIMEOUT_S, default 30 s), not waited for. */
#define BEAT_IPC_HANDLE_BYTES 2048
#define BEAT_IPC_MAX_RANKS 16
int beat_comm_create_ipc(beat_ctx* ctx, int rank, int world, int peer_lo, int peer_hi, int64_t max_plane_doubles,
                         const void* host_rccl_id, beat_allreduce_fn allreduce, void* user, void* host_handle_out,
                         beat_comm** out);
int beat_comm_ipc_connect(beat_comm* comm, const void* host_handle_lo, const void* host_handle_hi);
int beat_comm_ipc_connect_all(beat_comm* comm, const void* host_handles, int count);
/* The same for ranks that are THREADS of one process (each with its own beat_ctx; rehearsals of 8 and 16 ranks on one
 * GPU, tests/_ipc_ranks_script.py): host_comms[r] = rank r's ipc communicator; nothing is exported or opened, every
 * communicator must outlive the others' last use of it.  Reference analogue: the PETSc reductions and ghost updates inside
 * KSP.solve / scatter_forward, src/beat/base_model.py:203-206,236,242, which the reference's CI runs under mpirun -n 2. */
int beat_comm_ipc_connect_local(beat_comm* comm, beat_comm* const* host_comms, int count);
int beat_comm_destroy(beat_comm* comm);
/* host_out[4]: transport (BEAT_TRANSPORT_*), ranks of the all-reduce communicator as RCCL itself counts them
 * (ncclCommCount; 0 without RCCL), world as given at creation, who sums: 1 RCCL, 2 the ipc mailboxes, 0 the caller. */
#define BEAT_TRANSPORT_CALLBACKS 0
#define BEAT_TRANSPORT_RCCL 1
#define BEAT_TRANSPORT_RCCL_SERIAL 2
#define BEAT_TRANSPORT_IPC 3
int beat_comm_info(beat_comm* comm, int* host_out);
/* Number of beat_pde_solve_dist calls on this communicator that ran the single-reduction iteration (environment
 * BEAT_DIST_MERGED=1, constant-coefficient operators: one all-reduce of three values per PCG iteration instead of two
 * dependent ones -- what PETSc offers as -ksp_type pipecg / groppcg next to the cg of src/beat/base_model.py:199-206). */
int64_t beat_comm_merged_solves(const beat_comm* comm);
/* Event timing of the communication inside the decomposed solve (RCCL and ipc transports): enable = 1 drops what
 * was collected and starts, 0 stops.  beat_comm_profile_read synchronises and fills host_out[6] = {ms the ghost-plane
 * transfers took on their stream, their number, ms the all-reduces took on the compute stream (waiting for the
 * slowest rank included), their number, ms the compute stream stood waiting for ghost planes, number of waits}.  The
 * events cost a few microseconds each: profile a few extra steps, not the timed region. */
int beat_comm_profile(beat_comm* comm, int enable);
int beat_comm_profile_read(beat_comm* comm, double* host_out);
/* The two operations on their own (set-up code, tests): complete in stream order on the context's stream. */
int beat_comm_halo_exchange(beat_comm* comm, double* dev_field, int64_t n, int64_t plane_doubles);
int beat_comm_allreduce_sum(beat_comm* comm, double* dev_values, int count);
/* beat_pde_solve_ex on a decomposed grid: same arguments, results and deferred-flush contract; every rank calls
 * it with its own slab handle (z_lo_phys / z_hi_phys = whether peer_lo / peer_hi is absent).  Per iteration: ghost
 * planes of p travel on the side stream while q = A p is computed on the planes that need none, then the one or
 * two boundary planes; all-reduce of p.q; r -= alpha q; all-reduce of (r.z, r.r); p = z + beta p.  On
 * constant-coefficient grids (the kernels that never store q, beat_pde_rr.hip) it is the ghost planes of r that
 * travel -- overlapped with the two reductions and the interior part of the next p-and-dot pass -- and every rank
 * forms p on its ghost planes itself.  The host enqueues as many iterations as the previous solve needed and reads
 * the device-side convergence latch once (iterations and the latch are identical on all ranks because they derive
 * from all-reduced values only). */
int beat_pde_solve_dist(beat_pde* pde, beat_comm* comm, const double* dev_v_prev,
                        const double* const* host_dev_stim_w, const double* host_stim_amp, int n_stim,
                        double* dev_x, double* dev_work, double rtol, double atol, int max_it, int defer_flush,
                        beat_ksp_info* info, int* host_pending);
/* The decomposed solve's first half, as beat_pde_solve_begin is beat_pde_solve_ex's (round 5): ghost planes, right-hand side, the
 * iterations the previous solve needed + 1 with their exchanges and all-reduces and the copy of the scalar state are in the
 * streams when it returns; beat_pde_solve_end (or the next ionic step with pending = -1, enqueued behind the solve) finishes it.
 * Every rank sees the same all-reduced scalars -- on the device as on the host -- so every rank's launch behind the solve does
 * the same.  Between the ranks' steps the host was on the critical path once per step: a wake-up, the read of the latch and the
 * ionic launch, 75 us of a 1.9 ms step on one rank's 512 x 512 x 64 share of the 512^3 grid (profiles/r05_slab64.md). */
int beat_pde_solve_dist_begin(beat_pde* pde, beat_comm* comm, const double* dev_v_prev, const double* const* host_dev_stim_w,
                              const double* host_stim_amp, int n_stim, double* dev_x, double* dev_work, double rtol, double atol,
                              int max_it);

/* P1 point evaluation: out[k] = sum_j w[k,j] * field[idx[k,j]], j < 4
 * (scifem.evaluate_function stand-in, demos/niederer_benchmark.py:285).  Synchronises. */
int beat_field_probe(beat_ctx* ctx, const double* dev_field, const int64_t* host_idx,
                     const double* host_w, int npts, double* host_out);
/* The same into a device buffer (npts doubles), enqueued like any kernel: a time loop that looks at its probes every
 * step (demos/niederer_benchmark.py:285-291) records them per step and reads the record back once in a while
 * instead of synchronising every step. */
int beat_field_probe_record(beat_ctx* ctx, const double* dev_field, const int64_t* host_idx,
                            const double* host_w, int npts, double* dev_out);
/* sum_i x_i y_i on the local slab: lead integrals of the ECG recovery, ecg.py:295-298 (assemble_scalar of
 * (1/(4 pi sigma_b)) Im / |x - p| dx = weights . Im with precomputed nodal weights).  Synchronises. */
int beat_field_dot(beat_ctx* ctx, const double* dev_x, const double* dev_y, int64_t n, double* host_out);
/* min / max of a field (demos read v.max(), v.min() every step; avoids a full D2H); NaN-propagating as
 * numpy.min / numpy.max: a NaN anywhere makes both results NaN.  Synchronises. */
int beat_field_minmax(beat_ctx* ctx, const double* dev_field, int64_t n, double* host_min, double* host_max);

#ifdef __cplusplus
}
#endif
#endif /* BEAT_HIP_H */
