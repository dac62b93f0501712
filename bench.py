#!/usr/bin/env python3
"""Headline benchmark: operator-split monodomain step (TP06 GRL1 ionic step + theta-rule
diffusion solve) on a 512^3 anisotropic-fibre slab, dt = 0.01 ms  (BASELINE.json configs[3]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; the grid is cut into z-slabs (strong scaling: the 512^3 grid is fixed, the
slabs shrink as N grows).  A "step" is one call of ``MonodomainSplittingSolver.step``
(src/beat/monodomain_solver.py:53-116, theta_split = 1, theta_pde = 0.5) of THIS package's drop-in
classes -- ``MonodomainModel`` + ``DolfinODESolver`` + ``MonodomainSplittingSolver`` built exactly as the
reference's demos build them: ionic step on every node, then right-hand-side build + Jacobi-PCG to
``--rtol`` relative to ||b||.  (``--direct`` drives the same kernels by bare C-ABI calls instead; at
this size the two agree to the run-to-run spread.)  Inputs are synthetic and already resident in HBM when the timed
region starts.  Rank 0 prints ONE JSON line.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "fenicsx-beat_amd"))
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0  # measured streaming-copy ceiling, same guide

S_L, S_T = 9.5301e-4, 1.2576e-4  # uA/mV, Niederer harmonic-mean conductivities / chi (mm mesh)
C_M = 0.01  # uF/mm^2
THETA = 0.5
DT = 0.01
H = 0.1  # mm


ISOTROPIC = False  # --iso: M = s_l * I (BASELINE.json configs[2]); default: fibre at 30 degrees in the xy plane


def conductivity():
    if ISOTROPIC:
        return S_L * np.eye(3)
    f0 = np.array([np.cos(np.pi / 6.0), np.sin(np.pi / 6.0), 0.0])
    return S_L * np.outer(f0, f0) + S_T * (np.eye(3) - np.outer(f0, f0))


def tp06_defaults():
    from beat.models import tp06

    ic = tp06.init_state_values(
        V=-85.23, Xr1=0.00621, Xr2=0.4712, Xs=0.0095, m=0.00172, h=0.7444, j=0.7045, d=3.373e-05, f=0.7888,
        f2=0.9755, fCass=0.9953, s=0.999998, r=2.42e-08, Ca_i=0.000126, R_prime=0.9073, Ca_SR=3.64,
        Ca_ss=0.00036, Na_i=8.604, K_i=136.89,
    )  # demos/niederer_benchmark.py:44-67
    return ic, tp06.init_parameter_values(stim_amplitude=0.0), tp06.state_index("V")


def init_states(ctx, states, ic, v_index, n_glob, slab, seed, nz_glob=None):
    """TP06 resting state everywhere, V raised by a 60 mV Gaussian bump (sigma 2 mm) at the box
    centre so that a depolarisation front travels through the timed steps, every other state
    multiplied by (1 + 0.01 u), u ~ U(-1, 1) (seeded per rank)."""
    torch = ctx.torch
    nx = ny = n_glob
    gen = torch.Generator(device=ctx.device)
    gen.manual_seed(seed + slab.rank)
    plane = nx * ny
    zc = torch.arange(slab.z0, slab.z1, device=ctx.device, dtype=torch.float64) * H
    yc = torch.arange(ny, device=ctx.device, dtype=torch.float64) * H
    xc = torch.arange(nx, device=ctx.device, dtype=torch.float64) * H
    c = 0.5 * (n_glob - 1) * H
    cz = 0.5 * ((nz_glob or n_glob) - 1) * H
    for k in range(states.S):
        row = states.rows[k]
        if k == v_index:  # one broadcast expression over the slab (no per-plane device copies)
            r2 = (zc[:, None, None] - cz) ** 2 + (yc[None, :, None] - c) ** 2 + (xc[None, None, :] - c) ** 2
            row.view(slab.nz, ny, nx).copy_(float(ic[k]) + 60.0 * torch.exp(-r2 / (2.0 * 2.0**2)))
            del r2
        else:
            u = torch.rand(row.shape[0], generator=gen, device=ctx.device, dtype=torch.float64)
            u.mul_(2.0).sub_(1.0)  # in place, same values as before: one temporary row, not three (1024^3: 8.6 GB each)
            u.mul_(0.01).add_(1.0).mul_(float(ic[k]))
            row.copy_(u)
            del u


def developed_front_profile(ctx, n, ic, params, v_index, rtol):
    """(19, n) TP06 states of a travelling depolarisation front along x, from a 1-D pre-run ON THE DEVICE: an n-node
    cable with the 3-D problem's h, dt and xx-conductivity, left end raised to +20 mV, stepped (same ionic kernel,
    same theta-rule solve) until the front has reached the middle of the cable.  Returns (profile tensor, steps)."""
    import ctypes as C

    from beat import _hip, _stencil
    from beat._device import StateArray
    from beat._engine import DiffusionSolver, HipOps, Slab

    mxx = float(conductivity()[0, 0])
    mt, kt = _stencil.stencil_tables(1, (H,), np.array([[mxx]]))
    ops = HipOps(ctx, (n, 1, 1), True, True, mt, kt)
    ops.set_timestep(C_M, THETA, DT)
    solver = DiffusionSolver(ops, Slab(1))
    sa = StateArray(ctx, len(ic), n, n)
    for k in range(len(ic)):
        sa.rows[k].fill_(float(ic[k]))
    sa.rows[v_index][:20].fill_(20.0)
    v = sa.row_field(v_index)
    p_host = np.ascontiguousarray(params)
    steps, t = 0, 0.0
    while steps < 20000:
        for _ in range(200):
            _hip.check(ctx.lib.beat_ode_step(ctx.handle, _hip.MODEL_TP06_GRL1, sa.ptr, n, sa.ld, p_host.ctypes.data_as(C.c_void_p),
                                             len(p_host), None, 0, t, DT, v_index, None))
            solver.solve(v, [], [], v, rtol=rtol, atol=1e-50, max_it=500)
            t += DT
        steps += 200
        up = (sa.rows[v_index] > -40.0).nonzero()
        if up.numel() and int(up.max()) >= n // 2:
            break
    return sa.rows.clone(), steps


def cpu_baseline(n_side: int, steps: int, rtol: float):
    """The oracle's C restatement (oracle/beat_oracle.c: TP06 GRL1 + 15-point-stencil Jacobi-PCG, OpenMP over
    nodes) timed on this host's cores on an n_side^3 sample of the same workload; the single-threaded NumPy
    oracle (how the reference evaluates ``fun``: whole-array expressions) is timed on a smaller sample."""
    import subprocess

    from oracle import fem, ionic

    M = conductivity()
    mt, kt = fem.stencil_table(3, (H, H, H), M, 1.0, 1.0)
    A, B = C_M * mt + THETA * DT * kt, C_M * mt - (1.0 - THETA) * DT * kt
    ic = ionic.tp06_init_state_values()
    P = ionic.tp06_init_parameter_values(stim_amplitude=0.0)
    vi = ionic.tp06_state_index("V")

    def sample_states(n):
        g = (np.arange(n) - 0.5 * (n - 1)) * H
        r2 = g[:, None, None] ** 2 + g[None, :, None] ** 2 + g[None, None, :] ** 2
        S = np.repeat(ic[:, None], n**3, axis=1)
        S[vi] += 60.0 * np.exp(-r2.ravel() / (2.0 * (0.2 * n * H) ** 2))
        return np.ascontiguousarray(S)

    out = {}
    try:
        from oracle import cport

        try:
            cport.load()
        except FileNotFoundError:
            subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
            cport.load()
        n = n_side
        S = sample_states(n)
        work = np.empty(5 * n**3)
        v = np.empty(n**3)
        threads = cport.load().oracle_num_threads()
        cport.tp06_grl1(S.copy(), 0.0, DT, P)  # touch pages / spin up the thread pool
        tic = time.perf_counter()
        its, t = 0, 0.0
        for _ in range(steps):
            cport.tp06_grl1(S, t, DT, P)
            v[:] = S[vi]
            its += cport.theta_step(A, B, (n, n, n), v, None, 0.0, rtol, work=work)
            S[vi] = v
            t += DT
        wall = time.perf_counter() - tic
        # the same port on ONE core, on a smaller sample of the same workload
        n1, steps1 = min(n, 64), max(2, min(steps, 10))
        cport.load().oracle_set_num_threads(1)
        S1 = sample_states(n1)
        work1, v1 = np.empty(5 * n1**3), np.empty(n1**3)
        tic1 = time.perf_counter()
        t1 = 0.0
        for _ in range(steps1):
            cport.tp06_grl1(S1, t1, DT, P)
            v1[:] = S1[vi]
            cport.theta_step(A, B, (n1, n1, n1), v1, None, 0.0, rtol, work=work1)
            S1[vi] = v1
            t1 += DT
        wall1 = time.perf_counter() - tic1
        cport.load().oracle_set_num_threads(threads)
        c_single = {"value": n1**3 * steps1 / wall1, "unit": "node-updates/s", "cores": 1,
                    "sample": f"{n1}^3 nodes x {steps1} steps, C oracle on one thread, {wall1:.1f} s"}
        out = {
            "value": n**3 * steps / wall,
            "unit": "node-updates/s",
            "cores": threads,
            "kind": "port",
            "sample": f"{n}^3 nodes x {steps} steps, TP06 GRL1 + P1 theta-rule Jacobi-PCG (avg {its / steps:.1f} its), "
                      f"C/OpenMP oracle (oracle/beat_oracle.c), {wall:.1f} s",
            "c_single_core": c_single,
        }
    except Exception as exc:  # no C toolchain on this host: fall back to the NumPy oracle alone
        out = {"value": None, "unit": "node-updates/s", "cores": 0, "kind": "port", "sample": f"C oracle unavailable: {exc}"}

    # NumPy oracle, one thread (the reference's own evaluation style for the ionic step)
    n = min(n_side, 40)
    mesh = fem.BoxMesh((n - 1,) * 3, ((n - 1) * H,) * 3)
    model = fem.OracleMonodomainModel(mesh, M, [], C_m=C_M, theta=THETA, default_timestep=DT, solver="pcg", rtol=rtol)
    S = sample_states(n)
    ionic.tp06_generalized_rush_larsen(S[:, :8], 0.0, DT, P)  # builds the SymPy-derived Jacobian once (not timed)
    tic = time.perf_counter()
    t = 0.0
    nsteps = 3
    for _ in range(nsteps):
        S = ionic.tp06_generalized_rush_larsen(S, t, DT, P)
        model.state[:] = S[vi]
        model.assign_previous()
        model.step((t, t + DT))
        S[vi] = model.state
        t += DT
    wall = time.perf_counter() - tic
    out["numpy_single_core"] = {"value": n**3 * nsteps / wall, "unit": "node-updates/s", "cores": 1,
                                "sample": f"{n}^3 nodes x {nsteps} steps, NumPy/SciPy oracle, {wall:.1f} s"}
    if out["value"] is None:
        out.update(value=out["numpy_single_core"]["value"], cores=1, sample=out["numpy_single_core"]["sample"])
    return out


def progress(msg: str) -> None:
    """One line on stderr per phase: what the launching parent's watchdog (launch_ranks) listens for."""
    print(f"[bench rank {os.environ.get('RANK', '0')} +{time.perf_counter() - _T0:.1f}s] {msg}", file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def _free_port() -> int:
    """A port for a rendezvous on this host, from BELOW the kernel's ephemeral range (32768-60999): a port handed out by
    bind(0) can be taken by any process's outgoing connection between this probe and the launcher's own bind -- seen once as
    EADDRINUSE from torchrun's TCPStore in the GPU suite -- while nothing but another listener takes one of these."""
    import random
    import socket

    for _ in range(64):
        port = random.randint(20000, 32000)
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", port))
            except OSError:
                continue
            return port
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class ClockSampler:
    """Shader clock, package power and temperatures of this rank's GPU while a timed region runs, read from the amdgpu hwmon
    files in sysfs (no tool is started, nothing touches the GPU): the ionic kernel runs at the package power limit, so the
    clock it is given decides the headline to a few percent -- the line says what it was.  Best effort: `summary()` is None
    when the files cannot be found or read."""

    def __init__(self, device_index: int):
        import glob

        self.dir = None
        try:
            import torch

            pr = torch.cuda.get_device_properties(device_index)
            want = f"{int(pr.pci_domain_id):04x}:{int(pr.pci_bus_id):02x}:{int(pr.pci_device_id):02x}."
            for card in glob.glob("/sys/class/drm/card[0-9]*"):
                if "-" in os.path.basename(card):
                    continue
                if os.path.basename(os.path.realpath(card + "/device")).startswith(want):
                    hw = glob.glob(card + "/device/hwmon/hwmon*")
                    if hw:
                        self.dir = hw[0]
                    break
        except Exception:  # noqa: BLE001
            self.dir = None
        self.samples, self._stop, self._thread = [], None, None

    def _read(self, name):
        with open(os.path.join(self.dir, name)) as f:
            return float(f.read().strip())

    def _loop(self):
        while not self._stop.is_set():
            try:
                self.samples.append((self._read("freq1_input") / 1e6, self._read("power1_input") / 1e6,
                                     self._read("temp2_input") / 1e3, self._read("temp3_input") / 1e3))
            except Exception:  # noqa: BLE001
                return
            self._stop.wait(0.02)

    def start(self):
        import threading

        if self.dir is None:
            return
        self.samples = []
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._loop, daemon=True)
        self._thread.start()

    def stop(self):
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=2.0)
            self._thread = None

    def summary(self):
        if not self.samples:
            return None
        a = np.array(self.samples)
        return {"sclk_mhz": {"mean": float(a[:, 0].mean()), "min": float(a[:, 0].min()), "max": float(a[:, 0].max())},
                "power_w": {"mean": float(a[:, 1].mean()), "max": float(a[:, 1].max())},
                "junction_c": float(a[:, 2].max()), "hbm_c": float(a[:, 3].max()), "samples": int(len(a)),
                "source": "amdgpu hwmon (freq1_input, power1_input, temp2_input, temp3_input), polled every 20 ms over the timed steps"}


def ordering_of(comm_info) -> str:
    """``config.ordering`` of an N > 1 line: "overlapped" = ghost planes on a side stream behind the interior stencil (the
    design of DESIGN section 5: transports rccl and ipc), "serial" = every communication operation on the compute stream
    (rccl-serial; the host-staged callbacks; the stage-driven fallback without a library communicator)."""
    if comm_info is None:
        return "serial"
    return "overlapped" if comm_info.get("transport") in ("rccl", "ipc") else "serial"


def launch_ranks(args, argv) -> int:
    """``python bench.py --gpus N`` called directly (no launcher around it, as the driver calls it; the reference's CI
    starts its parallel job itself too: .github/workflows/main-mpi.yml:33): this process runs no GPU work -- it counts the
    devices (which may initialise the HIP runtime here when torch has no amdsmi to ask; the ranks are fresh children
    either way), starts N fresh rank processes (``python -m torch.distributed.run --nproc-per-node N bench.py ...``), relays rank
    0's one JSON line and their return code, and watches them: a job that prints nothing for longer than the watchdog
    allows is killed (whole process group), and ONE more attempt is made in fresh children with the deadlock-proof
    communication order (``BEAT_DIST_SERIAL=1``: ghost planes and all-reduces on one RCCL communicator, one stream).
    Returns the exit code."""
    import signal
    import subprocess
    import threading

    n = args.gpus
    backend = os.environ.get("BEAT_DIST_BACKEND", "nccl")
    if backend == "nccl":  # one device per rank; counting devices does not initialise the GPU
        try:
            import torch

            ndev = torch.cuda.device_count()
        except Exception as exc:  # noqa: BLE001
            print(f"bench.py: cannot count GPUs ({exc})", file=sys.stderr)
            return 2
        if ndev < n:
            print(f"bench.py: --gpus {n} needs {n} visible GPUs, this host shows {ndev} "
                  "(BEAT_DIST_BACKEND=gloo rehearses N ranks on fewer devices; its numbers mean nothing)", file=sys.stderr)
            return 2
    t_start = float(os.environ.get("BEAT_BENCH_WATCHDOG_START_S", "420"))  # first output: N x `import torch` on a cold box
    t_phase = float(os.environ.get("BEAT_BENCH_WATCHDOG_S", str(max(180.0, 0.5 * (args.steps + args.warmup)))))
    attempts = [{}]
    if os.environ.get("BEAT_DIST_SERIAL", "0") != "1" and os.environ.get("BEAT_BENCH_NO_RETRY", "0") != "1":
        attempts.append({"BEAT_DIST_SERIAL": "1"})
    history = []
    k = -1
    port_retries = 0
    while k + 1 < len(attempts):
        k += 1
        extra = attempts[k]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve()), *argv]
        env = dict(os.environ, **extra)
        env["BEAT_BENCH_SELF_LAUNCHED"] = "1"  # the ranks are watched: they may start with the overlapped ordering
        env.setdefault("OMP_NUM_THREADS", "4")
        print(f"[bench launcher] attempt {k + 1}/{len(attempts)}: {n} ranks" + (f" with {extra}" if extra else ""), file=sys.stderr, flush=True)
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
        state = {"last": time.monotonic(), "seen": False}
        out_lines = []

        def pump(stream, sink, relay):
            for line in stream:
                state["last"] = time.monotonic()
                state["seen"] = True
                if relay:
                    if "EADDRINUSE" in line or "address already in use" in line:
                        state["port_taken"] = True
                    sys.stderr.write(line)
                    sys.stderr.flush()
                else:
                    sink.append(line)

        threads = [threading.Thread(target=pump, args=(proc.stdout, out_lines, False), daemon=True),
                   threading.Thread(target=pump, args=(proc.stderr, None, True), daemon=True)]
        for th in threads:
            th.start()
        timed_out = False
        tic = time.monotonic()
        while proc.poll() is None:
            time.sleep(0.25)
            silent = time.monotonic() - state["last"]
            if silent > (t_phase if state["seen"] else t_start):
                timed_out = True
                print(f"[bench launcher] no output for {silent:.0f} s: killing the ranks", file=sys.stderr, flush=True)
                for sig in (signal.SIGTERM, signal.SIGKILL):
                    try:
                        os.killpg(proc.pid, sig)  # the launcher and every rank (own session = own process group)
                    except ProcessLookupError:
                        break
                    try:
                        proc.wait(timeout=10)
                        break
                    except subprocess.TimeoutExpired:
                        continue
                break
        rc = proc.wait()
        for th in threads:
            th.join(timeout=5)
        lines = [ln for ln in out_lines if ln.strip()]
        history.append({"attempt": k + 1, "env": extra, "rc": rc, "timed_out": timed_out, "wall_s": round(time.monotonic() - tic, 1)})
        if rc == 0 and not timed_out and len(lines) == 1:
            try:
                out = json.loads(lines[0])
                out.setdefault("config", {})["launch"] = {"by": "bench.py --gpus N (self-launched ranks)", "attempts": history}
                print(json.dumps(out), flush=True)
            except ValueError:
                print(lines[0], end="", flush=True)
            return 0
        if rc != 0 and not timed_out and len(lines) == 1:
            # ranks that measured and then left non-zero because the decomposed run differs from the undivided one
            # (`multi_rank_parity.ok` false): a result, not a hang -- relay the line, keep the exit code, no second attempt
            try:
                out = json.loads(lines[0])
            except ValueError:
                out = None
            if isinstance(out, dict) and out.get("multi_rank_parity", {}).get("ok") is False:
                out.setdefault("config", {})["launch"] = {"by": "bench.py --gpus N (self-launched ranks)", "attempts": history}
                print(json.dumps(out), flush=True)
                print("[bench launcher] multi-rank parity FAILED (see multi_rank_parity in the line)", file=sys.stderr, flush=True)
                return rc
        print(f"[bench launcher] attempt {k + 1} failed: rc {rc}" + (", watchdog" if timed_out else "")
              + (f", {len(lines)} stdout lines" if len(lines) != 1 else ""), file=sys.stderr, flush=True)
        if state.get("port_taken") and not timed_out and port_retries < 3:
            # the rendezvous port was taken between the probe and the launcher's bind: the same attempt again on another
            # port (this is not what the serial fallback is for)
            port_retries += 1
            history[-1]["port_taken"] = True
            k -= 1
    return history[-1]["rc"] or 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", "--n", dest="n", type=int, default=512,
                    help="nodes per axis of the cubic slab (use --size under torch.distributed.run: --n is ambiguous there)")
    ap.add_argument("--rtol", type=float, default=1e-8)
    ap.add_argument("--cpu-sample", type=int, default=160, help="side of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-steps", type=int, default=40)
    ap.add_argument("--pc-degree", type=int, default=int(os.environ.get("BEAT_PC_DEGREE", "1")),
                    help="1 = Jacobi-PCG, m >= 2 = Chebyshev polynomial preconditioner with m terms")
    ap.add_argument("--iso", action="store_true", help="isotropic conductivity (configs[2], use with --n 256)")
    ap.add_argument("--no-defer", action="store_true", help="apply x += sum alpha_j p_j in its own pass after every "
                    "solve instead of inside the next ionic kernel")
    ap.add_argument("--guess-order", type=int, default=int(os.environ.get("BEAT_GUESS_ORDER", "-1")), choices=[-1, 0, 1, 2, 3, 4],
                    help="initial guess of each diffusion solve: 0 = the ionic step's potential, m = plus the degree-(m-1) "
                    "extrapolation in time of the last m diffusion increments, -1 = order 1-4 chosen per solve by the "
                    "iteration counts seen (beat_pde_set_guess_order; -1 is the package default)")
    ap.add_argument("--no-front", action="store_true", help="skip the second, developed-front measurement")
    ap.add_argument("--direct", action="store_true", help="drive the kernels by bare C-ABI calls (beat_ode_step_pending + "
                    "DiffusionSolver.solve) instead of the public API's MonodomainSplittingSolver.step")
    ap.add_argument("--size-z", "--nz", dest="nz", type=int, default=0, help="z planes of the global grid (default: --n); e.g. --nz 64 with "
                    "BEAT_FORCE_DISTRIBUTED=1 rehearses on one GPU the slab one of 8 ranks owns at 512^3")
    args = ap.parse_args()
    global ISOTROPIC
    ISOTROPIC = args.iso
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # called without a launcher (``python bench.py --gpus N``): start the ranks as fresh children of a parent that
        # stays away from the GPU, watch them, relay rank 0's line
        raise SystemExit(launch_ranks(args, sys.argv[1:]))
    # stdout carries exactly one line, the JSON result: libraries that write to file descriptor 1 themselves (RCCL
    # prints a five-line version banner when its communicator is created) are sent to stderr instead
    sys.stdout.flush()
    result_stream = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    if os.environ.get("BEAT_BENCH_TEST_HANG") == "1" and args.gpus > 1:  # tests of the launcher's watchdog: a rank that never reports
        time.sleep(3600)
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # BEAT_DIST_BACKEND=gloo: rehearsal of the multi-process path with several ranks sharing one GPU (host-staged
    # transport, beat._engine._HostStagedDist); the numbers of such a run mean nothing
    backend = os.environ.get("BEAT_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    # Ranks started by somebody else's launcher (``python -m torch.distributed.run ... bench.py --gpus N``) have no watchdog
    # of ours behind them: a first measurement that hangs would be the end of the run.  They therefore measure the
    # headline first on the ordering that cannot deadlock (ghost planes and all-reduces on ONE RCCL communicator and ONE
    # stream), then time the overlapped ordering (side stream, second communicator) and the ipc transport under the
    # deadline below, and re-measure the headline on whichever proved more than 3 % faster (`config.transport_choice`).
    # `python bench.py --gpus N` launches and watches its own ranks and starts with the overlapped ordering.
    external_launch = (world > 1 and backend == "nccl" and os.environ.get("BEAT_BENCH_SELF_LAUNCHED") != "1"
                       and "BEAT_DIST_SERIAL" not in os.environ and "BEAT_DIST_TRANSPORT" not in os.environ)
    if external_launch:
        os.environ["BEAT_DIST_SERIAL"] = "1"
    # BEAT_FORCE_DISTRIBUTED=1 rehearses the collective code path (RCCL all-reduces, stage kernels driven
    # from Python) on a single rank; the reported numbers are then NOT the single-GPU headline.
    force_dist = os.environ.get("BEAT_FORCE_DISTRIBUTED", "0") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from beat import _hip

    progress(f"process group up ({backend}, world {world})" if (world > 1 or force_dist) else "started")
    # a checkout without the built artefacts (the .so is not tracked): rank 0 builds, everybody waits -- the barrier is
    # unconditional, so a rank that only looks once the file is already there cannot skip it and leave rank 0 waiting
    if rank == 0 and not _hip.library_path().is_file():
        import __graft_entry__ as entry

        entry.build()
    if world > 1 or force_dist:
        dist.barrier()
    from beat import _stencil
    from beat._device import Context, StateArray
    from beat._engine import DiffusionSolver, HipOps, Slab

    # N > 1: correctness before speed.  Two small problems (a TP06 slab with constant coefficients, a voxel shell with per-node
    # rows and two parameter classes) run DECOMPOSED on the N ranks and UNDIVIDED on rank 0 through the public API, 25 split
    # steps each, on the transport the headline is about to be measured on; the alternatives get the same check before they are
    # timed (below).  `multi_rank_parity` in the line; a difference above bench_parity.TOLERANCE makes every rank exit non-zero
    # after the line is printed.  (The reference's parallel CI runs its ACCURACY tests under mpirun -n 2,
    # .github/workflows/main-mpi.yml:33.)
    parity, parity_hook = None, None
    if world > 1 and os.environ.get("BEAT_BENCH_PARITY", "1") == "1":
        import bench_parity

        class parity_hook:  # noqa: N801 -- installs a transport on the case's own operator, names it, removes it afterwards
            def __init__(self, name=None, serial=False):
                self.name, self.serial, self.label, self.alt = name, serial, None, None

            def __call__(self, pde_):
                from beat._engine import LibComm

                d = pde_._diffusion
                if self.name is not None:
                    self.alt = LibComm(pde_._ctx, pde_._mesh.slab, dist, None, self.name, serial=self.serial, plane_doubles=pde_._ops.plane)
                    self.default, d.libcomm = d.libcomm, self.alt
                lc = d.libcomm
                self.label = lc.info()["transport"] if lc is not None else "stage-driven"

            def done(self, pde_):
                if self.alt is not None:
                    pde_._ops.flush_pending()
                    torch.cuda.synchronize()
                    pde_._diffusion.libcomm = self.default
                    self.alt.close()
                    self.alt = None

        tic = time.perf_counter()
        hk = parity_hook()
        verdicts = bench_parity.run_cases(dist, rank, world, hook=hk)
        parity = {hk.label: dict(verdicts, seconds=time.perf_counter() - tic)}
        progress("multi-rank parity on " + hk.label + ": " + ", ".join(
            f"{c} {v.get('max_rel_diff', float('nan')):.1e}" for c, v in verdicts.items()) + f" ({time.perf_counter() - tic:.1f} s)")

    n = args.n
    nz_glob = args.nz or n
    plane = n * n
    ic, params, v_index = tp06_defaults()
    use_api = not (args.direct or args.no_defer)
    api_solver = mon = None
    if use_api:
        # the reference's own construction sequence (demos/niederer_benchmark.py:101-225) on this package's classes
        import beat
        from beat import grid as g

        class EventMonitor(beat.telemetry.BaseMonitor):
            """HIP events around the ionic and the diffusion stage of the fused step (its track_time keys
            "ode_step" / "pde_step"), on the stream the kernels are launched on."""

            def __init__(self):
                self.events, self.armed = {}, None

            def track_time(self, name):
                mon_, key = self, (self.armed, name)

                class Region:
                    def __enter__(self_r):
                        if mon_.armed is not None and name in ("ode_step", "pde_step"):
                            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                            mon_.events[key] = ev
                            ev[0].record()
                        return self_r

                    def __exit__(self_r, *exc):
                        if key in mon_.events and mon_.armed is not None and name in ("ode_step", "pde_step"):
                            mon_.events[key][1].record()
                        return False

                return Region()

        mesh = g.create_box(g.COMM_WORLD, [np.zeros(3), np.array([(n - 1) * H, (n - 1) * H, (nz_glob - 1) * H])],
                            [n - 1, n - 1, nz_glob - 1])
        slab = mesh.slab
        time_c = g.Constant(mesh, 0.0)
        pde = beat.MonodomainModel(time=time_c, mesh=mesh, M=conductivity(), C_m=C_M,
                                   params={"theta": THETA, "petsc_options": {"ksp_rtol": args.rtol, "ksp_atol": 1e-50, "ksp_max_it": 500,
                                                                              "ksp_guess_order": args.guess_order}})
        ctx = pde._ctx
        ops = pde._ops
        ops.set_preconditioner(args.pc_degree)
        solver = pde._diffusion
        V = g.functionspace(mesh, ("P", 1))
        from beat.models import tp06

        ode = beat.odesolver.DolfinODESolver(v_ode=g.Function(V), v_pde=pde.state, fun=tp06.generalized_rush_larsen,
                                             init_states=ic, parameters=params, num_states=len(ic), v_index=v_index)
        mon = EventMonitor()
        api_solver = beat.MonodomainSplittingSolver(pde=pde, ode=ode, monitor=mon)
        states = ode._dev.states  # the (19, N_local) device array the solver owns
        n_local = plane * slab.nz
    else:
        ctx = Context(local_rank)
        slab = Slab(nz_glob, rank, world)
        n_local = plane * slab.nz
        mass_tab, stiff_tab = _stencil.stencil_tables(3, (H, H, H), conductivity())
        ops = HipOps(ctx, (n, n, slab.nz), slab.lo_phys, slab.hi_phys, mass_tab, stiff_tab)
        ops.set_preconditioner(args.pc_degree)
        ops.set_guess_order(args.guess_order)
        ops.set_timestep(C_M, THETA, DT)
        solver = DiffusionSolver(ops, slab, force_distributed=force_dist)
        states = StateArray(ctx, len(ic), n_local, plane)
    lib = ctx.lib
    init_states(ctx, states, ic, v_index, n, slab, 1234, nz_glob)
    v_field = states.row_field(v_index)  # PDE unknown lives in the V row: no ODE<->PDE copies
    if world > 1:
        # RCCL creates its point-to-point channels on first use (seconds): do that outside any timed step, whatever
        # --warmup says.  The exchange fills the V row's ghost planes with the neighbours' boundary planes, which is
        # what the first right-hand side needs anyway.
        solver.exchange_halo(v_field)
        dist.all_reduce(torch.zeros(2, dtype=torch.float64, device=ctx.device if backend == "nccl" else "cpu"))
        torch.cuda.synchronize()
    import ctypes as C

    p_host = np.ascontiguousarray(params)
    p_ptr = p_host.ctypes.data_as(C.c_void_p)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    clocks = ClockSampler(ctx.device.index if ctx.device.index is not None else 0)

    def timed_run(t, warmup, steps):
        """`warmup` untimed steps, then exactly `steps` timed ones bracketed by barriers; returns the statistics."""
        ev_ode = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        ev_pde_end = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
        if mon is not None:
            mon.events.clear()
        guess_fields, guess_orders = [], []
        iters, pend_counts = [], []  # pend: search directions the timed ionic launches applied for the previous solve

        gap_us = float(os.environ.get("BEAT_BENCH_GAP_US", "0"))  # experiment: extra host time between two steps (GPU idle)

        def step(t, i=None):
            if gap_us > 0.0:
                until = time.perf_counter() + gap_us * 1e-6
                while time.perf_counter() < until:
                    pass
            # the previous solve left its last x += sum alpha_j p_j to this step's ionic kernel (deferred-x PCG)
            pend = ops.pending
            lazy_open = getattr(ops, "open_x", None) is not None  # the previous solve is still open: this step's launch goes behind it
            if i is not None and not lazy_open:
                pend_counts.append(pend[2] if pend else 0)
                gt = ops.guess_traffic()  # increments this launch reads / writes for the initial guess (only if it applies the update)
                guess_fields.append(gt["reads"] + gt["writes"] if (pend and gt["pending"]) else 0)
                guess_orders.append(gt["order"])
            if use_api:
                # (the KSP record is NOT read here: MonodomainSplittingSolver.step leaves its solve open until the next step's
                # ionic launch is queued behind it, and asking for pde.ksp would make the host wait -- the reference's demo
                # loop does not look at the KSP either, demos/niederer_benchmark.py:270-289; every record is taken from the
                # operator's log after the timed region)
                mon.armed = i
                api_solver.step((t, t + DT))
                res = None
                if i is not None and lazy_open:  # what the launch behind the open solve applied (known once that solve is finished)
                    gt = ops.guess_traffic()
                    pend_counts.append(0)  # (set from the log below)
                    guess_fields.append(gt["reads"] + gt["writes"] if gt["pending"] else 0)
                    guess_orders.append(gt["order"])
                if i is not None:
                    ev_ode[i] = mon.events.pop((i, "ode_step"))
                    ev_pde_end[i] = mon.events.pop((i, "pde_step"))[1]
            else:
                if i is not None:
                    ev_ode[i][0].record()
                ops.pending = None
                _hip.check(lib.beat_ode_step_pending(ctx.handle, _hip.MODEL_TP06_GRL1, states.ptr, n_local, states.ld, p_ptr,
                                                     len(p_host), None, 0, t, DT, v_index, None, ops.handle, ops.ring[0].ptr,
                                                     ops.fld, pend[2] if pend else 0))
                if i is not None:
                    ev_ode[i][1].record()
                res = solver.solve(v_field, [], [], v_field, rtol=args.rtol, atol=1e-50, max_it=500, defer_flush=not args.no_defer)
                if i is not None:
                    ev_pde_end[i].record()
            if res is not None:
                if res.converged_reason <= 0:
                    raise SystemExit(f"PCG did not converge: reason {res.converged_reason} after {res.iterations} iterations")
                if i is not None:
                    iters.append(res.iterations)

        for _ in range(warmup):
            step(t)
            t += DT
        barrier()
        if use_api:
            ops.flush_pending()
            ops.ksp_log = []
        clocks.start()
        tic = time.perf_counter()
        for i in range(steps):
            step(t, i)
            t += DT
        ops.flush_pending()  # the potential is complete when the timed region ends
        barrier()
        wall = time.perf_counter() - tic
        clocks.stop()
        if use_api:  # every solve of the timed steps, from the operator's log (one record per finished solve)
            log, ops.ksp_log = ops.ksp_log, None
            if len(log) != steps:
                raise SystemExit(f"{len(log)} KSP records for {steps} timed steps")
            for rec in log:
                if rec.converged_reason <= 0:
                    raise SystemExit(f"PCG did not converge: reason {rec.converged_reason} after {rec.iterations} iterations")
            iters.extend(rec.iterations for rec in log)
            # what each timed ionic launch applied for the solve before it: the directions of that solve's last ring cycle
            ring_len = len(ops.ring)
            counts = [rec.iterations % ring_len for rec in log]
            pend_counts[:] = [0] + counts[:-1]  # (the warm-up's last update was flushed before the timed region)
        if world > 1:
            w = torch.tensor([wall], dtype=torch.float64, device=ctx.device if backend == "nccl" else "cpu")
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
            wall = float(w.item())
        return dict(t=t, wall=wall, iters=iters, pend_counts=pend_counts, guess_fields=guess_fields, guess_orders=guess_orders,
                    clocks=clocks.summary(),
                    ode_ms=float(np.mean([a.elapsed_time(b) for a, b in ev_ode])),
                    pde_ms=float(np.mean([ev_ode[i][1].elapsed_time(ev_pde_end[i]) for i in range(steps)])))

    progress(f"set-up done ({n_local} nodes on this rank, transport "
             f"{solver.libcomm.info()['transport'] if getattr(solver, 'libcomm', None) is not None else 'none'}); timing")
    run = timed_run(0.0, args.warmup, args.steps)
    progress(f"headline timed: {run['wall'] / args.steps * 1e3:.3f} ms/step")
    wall, iters, pend_counts, ode_ms, pde_ms = run["wall"], run["iters"], run["pend_counts"], run["ode_ms"], run["pde_ms"]
    vmin, vmax = v_field.minmax()
    if world > 1:  # extrema over all slabs (NaN-propagating: a non-finite value on any rank shows)
        ext = torch.tensor([-vmin, vmax], dtype=torch.float64, device=ctx.device if backend == "nccl" else "cpu")
        dist.all_reduce(ext, op=dist.ReduceOp.MAX)
        bad = torch.tensor([0.0 if np.isfinite(vmin) and np.isfinite(vmax) else 1.0], dtype=torch.float64, device=ext.device)
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        vmin, vmax = (-float(ext[0]), float(ext[1])) if float(bad[0]) == 0.0 else (float("nan"), float("nan"))
    finite = bool(np.isfinite(vmin) and np.isfinite(vmax))

    # Second regime, same grid, same run: a DEVELOPED depolarisation front (the headline's timed steps follow a smooth
    # bump, whose right-hand side needs about half the PCG iterations a travelling wave needs).  A planar TP06 front
    # from a 1-D device pre-run is extruded over the slab, a few steps let the iteration count settle, then the same
    # number of steps is timed the same way.
    front = None
    if finite and not args.no_front:
        tic = time.perf_counter()
        prof, pre_steps = developed_front_profile(ctx, n, ic, params, v_index, args.rtol)
        for k in range(states.S):
            states.rows[k].view(-1, n).copy_(prof[k][None, :].expand(n_local // n, n))
        del prof
        ops.guess_reset()  # the recorded increments belong to the overwritten state
        setup_s = time.perf_counter() - tic
        progress("developed front prepared; timing")
        fr = timed_run(0.0, max(args.warmup, 5), args.steps)
        progress(f"developed front timed: {fr['wall'] / args.steps * 1e3:.3f} ms/step")
        fmin, fmax = v_field.minmax()
        if world > 1:
            ext = torch.tensor([-fmin, fmax, 0.0 if np.isfinite(fmin) and np.isfinite(fmax) else 1.0], dtype=torch.float64,
                               device=ctx.device if backend == "nccl" else "cpu")
            dist.all_reduce(ext, op=dist.ReduceOp.MAX)
            fmin, fmax = (-float(ext[0]), float(ext[1])) if float(ext[2]) == 0.0 else (float("nan"), float("nan"))
        finite = finite and bool(np.isfinite(fmin) and np.isfinite(fmax))
        kf = float(np.mean(fr["iters"]))
        front = {
            "what": f"planar TP06 depolarisation front normal to x at mid-slab, extruded from a 1-D device pre-run of {pre_steps} "
                    f"steps ({setup_s:.1f} s, untimed); {max(args.warmup, 5)} warm-up + {args.steps} timed steps",
            "value": n * n * nz_glob * args.steps / fr["wall"],
            "unit": "node-updates/s",
            "ms_per_step": fr["wall"] / args.steps * 1e3,
            "pcg_iterations_per_step": kf,
            "ode_ms": fr["ode_ms"],
            "pde_ms": fr["pde_ms"],
            "bytes_per_node_update": 16.0 * len(ic) + 16.0 + 88.0 * kf,
            "frac_of_8TBs_per_gpu": (16.0 * len(ic) + 16.0 + 88.0 * kf) * n * n * nz_glob * args.steps / fr["wall"] / 1e9 / world / HBM_PEAK_GBS,
            "v_min": fmin,
            "v_max": fmax,
            "clocks": fr["clocks"],
        }

    # The headline regime once more through MonodomainSplittingSolver.solve -- the reference's own driver loop
    # (src/beat/monodomain_solver.py:53-66) -- which hands the steps of a one-rank grid to the library's loop (beat_split_steps_big):
    # same kernels and values as the step() calls above, without Python between the steps (the device idles 0.15-0.18 ms per step
    # there, and the ionic kernel that follows an idle gap runs slower: DESIGN.md 7).  State re-initialised, same warm-up, same K steps.
    batched = None
    if world == 1 and use_api and finite and not force_dist and os.environ.get("BEAT_BENCH_BATCHED", "1") == "1":
        saved = (api_solver.monitor, pde.monitor, ode._dev.monitor)
        api_solver.monitor = pde.monitor = ode._dev.monitor = beat.telemetry.NullMonitor()
        try:
            if api_solver._can_batch(None):
                progress("the same steps through MonodomainSplittingSolver.solve")
                ops.flush_pending()
                init_states(ctx, states, ic, v_index, n, slab, 1234, nz_glob)
                ops.set_guess_order(args.guess_order)  # history AND the adaptive order's scores as at the start of the headline run
                api_solver.solve((0.0, args.warmup * DT), DT)
                api_solver.batch_ode_ms = [] if os.environ.get("BEAT_BENCH_BATCHED_EVENTS", "1") == "1" else None
                barrier()
                clocks.start()
                tic = time.perf_counter()
                api_solver.solve((args.warmup * DT, (args.warmup + args.steps) * DT), DT)
                ops.flush_pending()
                barrier()
                wall_b = time.perf_counter() - tic
                clocks.stop()
                bmin, bmax = v_field.minmax()
                if (api_solver.batch_ode_ms is None or len(api_solver.batch_ode_ms) == args.steps) and np.isfinite(bmin) and np.isfinite(bmax):
                    batched = {"what": f"MonodomainSplittingSolver.solve over the same {args.steps} steps after the same {args.warmup} warm-up steps "
                                       "(state re-initialised): the library's own step loop, no Python between the steps",
                               "value": n * n * nz_glob * args.steps / wall_b, "unit": "node-updates/s", "ms_per_step": wall_b / args.steps * 1e3,
                               "pcg_iterations_last_step": int(pde.ksp.iterations),
                               "ode_ms": float(np.mean(api_solver.batch_ode_ms)) if api_solver.batch_ode_ms else None,
                               "v_min": bmin, "v_max": bmax, "clocks": clocks.summary()}
                    progress(f"batched: {batched['ms_per_step']:.3f} ms/step")
                api_solver.batch_ode_ms = None
        except Exception as exc:  # noqa: BLE001 -- the headline is measured: an extra that fails must not cost the line
            batched = {"error": repr(exc)}
        finally:
            api_solver.monitor, pde.monitor, ode._dev.monitor = saved

    # N > 1: what each rank did, and what the communication inside the solve costs.  Profiled on a few EXTRA steps after
    # the timed regions (timing events around every exchange and all-reduce are not free): ms per step the ghost-plane
    # transfers took on their stream, the all-reduces on the compute stream (waiting for the slowest rank included), and
    # the compute stream stood waiting for ghost planes (= the part of the exchange the interior stencil did not hide).
    ranks_info, comm_info = None, None
    libcomm = getattr(solver, "libcomm", None)
    if (world > 1 or force_dist) and finite:
        prof = None
        if libcomm is not None:
            comm_info = libcomm.info()
            if comm_info["transport"] != "callbacks":
                nprof = 5
                libcomm.profile(True)
                t_prof = (fr if front else run)["t"]
                for _ in range(nprof):
                    if use_api:
                        api_solver.step((t_prof, t_prof + DT))
                    else:
                        pend = ops.pending
                        ops.pending = None
                        _hip.check(lib.beat_ode_step_pending(ctx.handle, _hip.MODEL_TP06_GRL1, states.ptr, n_local, states.ld, p_ptr,
                                                             len(p_host), None, 0, t_prof, DT, v_index, None, ops.handle, ops.ring[0].ptr,
                                                             ops.fld, pend[2] if pend else 0))
                        solver.solve(v_field, [], [], v_field, rtol=args.rtol, atol=1e-50, max_it=500, defer_flush=not args.no_defer)
                    t_prof += DT
                ops.flush_pending()
                libcomm.profile(False)
                raw = libcomm.profile_read()
                prof = {"steps": nprof, "halo_ms_per_step": raw["halo_ms"] / nprof, "halo_exchanges_per_step": raw["halo_count"] / nprof,
                        "allreduce_ms_per_step": raw["allreduce_ms"] / nprof, "allreduces_per_step": raw["allreduce_count"] / nprof,
                        "halo_stall_ms_per_step": raw["halo_stall_ms"] / nprof}
        k_pend_r = float(np.mean(run["pend_counts"])) if run["pend_counts"] else 0.0
        mine = {"rank": rank, "device": int(torch.cuda.current_device()), "planes": int(slab.nz), "nodes": int(n_local),
                "ode_ms": run["ode_ms"], "pde_ms": run["pde_ms"], "comm": prof,
                "roofline_ode": {"achieved": (16.0 * len(ic) + 8.0 * k_pend_r) * n_local / (run["ode_ms"] * 1e-3) / 1e9, "unit": "GB/s",
                                 "note": "state rows + pending directions only (the guess's increments are added on rank 0's line)"}}
        if front:
            mine["front_ode_ms"], mine["front_pde_ms"] = fr["ode_ms"], fr["pde_ms"]
        if world > 1:
            ranks_info = [None] * world
            dist.all_gather_object(ranks_info, mine)
        else:
            ranks_info = [mine]
        progress("per-rank figures gathered")

    # What a kernel that only moves bytes reaches on the very array the ionic kernel walks, measured here after the timed regions
    # with the LIBRARY's own loads and stores (beat_stream_probe, csrc/beat_probe.hip; round 4 used torch's x.mul_(1.0), which
    # is 1.2 TB/s below what the memory system gives -- profiles/r05_streaming.md): (i) in place over the flat array, every value
    # read once and written back, the best of three variants (one workgroup per chunk with non-temporal / plain accesses; a
    # grid-stride loop with four accesses in flight); (ii) the ionic kernel's own access pattern -- all S rows of the (S, ld)
    # array read at one node index, then written back: S row streams per wavefront.  The second is the ceiling the kernel can be
    # held against; `roofline.peak` stays the 8 TB/s of the data sheet.
    stream = None
    flat = states.buf[states.base: states.base + states.S * states.ld]  # the rows with their ghost-plane padding, contiguous
    if finite:
        def probe_ms(fn, reps=5):
            fn()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a, b in evs:
                a.record()
                fn()
                b.record()
            torch.cuda.synchronize()
            return sorted(a.elapsed_time(b) for a, b in evs)[reps // 2]

        nflat = flat.numel() & ~1
        fptr = C.c_void_p(flat.data_ptr())
        variants = {"1 WG per chunk, nt loads + stores": (3, 1, 0), "1 WG per chunk, plain": (0, 1, 0),
                    "16384 WGs, 4 in flight, nt loads + stores": (3, 4, 16384)}
        rates = {}
        for name, (pol, unroll, blocks) in variants.items():
            ms = probe_ms(lambda: _hip.check(lib.beat_stream_probe(ctx.handle, fptr, nflat, 0, pol, unroll, blocks, 0, 0)))
            rates[name] = 16.0 * nflat / (ms * 1e-3) / 1e9
        rows_rate = None
        if states.S in (1, 4, 8, 19, 45) and n_local % 2 == 0 and states.ld % 2 == 0:
            ms = probe_ms(lambda: _hip.check(lib.beat_stream_probe(ctx.handle, states.ptr, n_local, 4, 3, 1, 0, states.S, states.ld)))
            rows_rate = 16.0 * states.S * n_local / (ms * 1e-3) / 1e9
        torch_ms = probe_ms(lambda: flat.mul_(1.0))
        stream = {"rates": rates, "rows_rate": rows_rate, "torch_mul_rate": 16.0 * flat.numel() / (torch_ms * 1e-3) / 1e9}

    if rank == 0:
        n_total = n * n * nz_glob
        # HBM bytes and VALU counters per launch of the dominant kernel from the committed rocprofv3 PMC passes of this
        # very command (tools/measure_round.sh -> profiles/r0N_<tag>_pmc.json); only quoted when this run has the
        # configuration the counters were collected on
        traffic, valu, pmc_source = None, None, None
        for tfile in sorted((ROOT / "profiles").glob("r0*_pmc.json"), reverse=True):  # the newest round's file that fits
            tj = json.loads(tfile.read_text())
            cfg = tj.get("config", {})
            kern = next((v for k, v in tj.get("kernels", {}).items() if k.startswith("ode_step_kernel<Tp06Grl1, false, true")), None)
            if (cfg.get("n") == n and nz_glob == n and cfg.get("n_gpus") == world and bool(cfg.get("isotropic", False)) == ISOTROPIC
                    and kern and "hbm_read_bytes" in kern):
                traffic = kern.get("hbm_read_bytes", 0.0) + kern.get("hbm_write_bytes", 0.0)
                valu = kern
                pmc_source = f"profiles/{tfile.name}"
                break
        k_avg = float(np.mean(iters)) if iters else 0.0
        S = len(ic)
        # every state row read once + written once, plus one read per pending search direction of the previous
        # diffusion solve (the launch applies that solve's x += sum alpha_j p_j)
        k_pend = float(np.mean(pend_counts)) if pend_counts else 0.0
        # ... and, with an extrapolated initial guess, the increments it is built from (read) and the new one (written)
        # (counted per launch from the terms of the update it applies, beat_pde_guess_traffic: the adaptive policy moves between orders)
        g_bytes = 8.0 * float(np.mean(run["guess_fields"])) if run["guess_fields"] else 0.0
        ode_bytes = (16.0 * S + 8.0 * k_pend + g_bytes) * n_local
        achieved = ode_bytes / (ode_ms * 1e-3) / 1e9
        step_bytes = (16.0 * S + 16.0 + 88.0 * k_avg) * n_total
        out = {
            "metric": "node_updates_per_sec",
            "value": n_total * args.steps / wall,
            "unit": "node-updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"{n}^3-node " if nz_glob == n else f"{n}x{n}x{nz_glob}-node ") + ("isotropic slab (h=0.1 mm)" if ISOTROPIC else "anisotropic-fibre slab (h=0.1 mm, fibre 30 deg in xy)") + ", TP06 GRL1 ionic step, "
                            f"P1 consistent-mass theta=0.5 diffusion, Godunov splitting, dt=0.01 ms, "
                            f"PCG rtol={args.rtol:g} (x0 = " + ("previous v", "previous v + last increment", "previous v + linear extrapolation of the last two increments",
                                                          "previous v + quadratic extrapolation of the last three increments",
                                                          "previous v + cubic extrapolation of the last four increments",
                                                          "previous v + extrapolation of the last increments, order 1-4 chosen per solve")[args.guess_order] + "), " + ("Jacobi" if args.pc_degree <= 1 else f"Chebyshev-Jacobi polynomial preconditioner, {args.pc_degree} terms"),
                "nodes": n_total,
                "states_per_node": S,
                "driver": ("public API: beat.MonodomainSplittingSolver.step on MonodomainModel + DolfinODESolver" if use_api
                           else "direct C-ABI calls (beat_ode_step_pending + DiffusionSolver.solve)"),
                "parallelism": f"z-slabs x{world}" + (" (forced collective path)" if force_dist else "")
                               + ("" if backend == "nccl" else f" (REHEARSAL on {backend}, ranks share a GPU: not a measurement)"),
                "pcg_iterations_per_step": k_avg,
                "guess_order": args.guess_order,
                "guess_order_used": (float(np.mean(run["guess_orders"])) if run["guess_orders"] else None),
                # where the state array lies decides how fast its rows stream: StateArray allocates a few candidates, times the
                # library's streaming probe of the ionic kernels' pattern on each and keeps the best (beat/_device.py; GB/s)
                "state_placement": getattr(states, "placement", None),
                "work_placement": getattr(ops, "work_placement", None),  # the PCG's work fields, placed the same way (beat/_engine.py)
                "ode_ms": ode_ms,
                "pde_ms": pde_ms,
                "v_min": vmin,
                "v_max": vmax,
                "finite": finite,
                "clocks": run["clocks"],  # rank 0's GPU over the timed steps (ClockSampler)
            },
            "roofline": {
                "bound": "hbm",
                # the yardstick above is the HBM roofline (the path is bandwidth-shaped); THIS kernel sits close to three
                # ceilings at once
                "limiter": "fp64 VALU issue at the power-limited clock (see valu: four waves per SIMD since round 6, profiles/r06_ode_addressing.md; "
                           "package power: profiles/r02_power.md), over the rate its own 19-row access pattern streams at (see inplace_stream.rows_pattern)",
                "kernel": "ode_step_kernel<Tp06Grl1>",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": None if traffic is None else f"{pmc_source} (rocprofv3 PMC passes of this command at this size, committed; not re-measured in this run)",
                "algorithmic_bytes_per_launch": ode_bytes,
                "bytes_per_node": 16.0 * S + 8.0 * k_pend + g_bytes,
                "pending_directions_per_launch": k_pend,
                "valu": None if valu is None else {
                    # a wave walks over several 64-node tiles: instructions per wave x waves / (nodes / 64)
                    "instr_per_node": valu["valu_instr_per_wave"] * valu.get("waves", n_local / 64.0) * 64.0 / n_local,
                    "valu_busy_frac": valu["valu_busy"],            # PMC: SIMD cycles spent issuing VALU
                    # issue time of those instructions at the 2.4 GHz peak clock (4 cycles per wave64 fp64/32-bit
                    # VALU op on a SIMD16) over the kernel time measured live in this run
                    "frac_of_issue_peak": valu["valu_instr_per_wave"] * valu.get("waves", n_local / 64.0) * 4.0 / (1024.0 * 2.4e9) / (ode_ms * 1e-3),
                    "effective_clock_GHz": valu["gui_cycles_per_xcd"] / (valu["avg_us"] * 1e-6) / 1e9,
                    "source": f"{pmc_source} (rocprofv3 SQ/GRBM pass of this command, committed; kernel time measured live)",
                },
                "inplace_stream": None if stream is None else {
                    "what": "beat_stream_probe (the library's own global_load/store_dwordx4 kernels) in place over the same state array, "
                            "16 B per value, median of 5 launches after the timed region; rate = the best variant",
                    "rate": max(stream["rates"].values()),
                    "variants": stream["rates"],
                    "unit": "GB/s",
                    "kernel_frac_of_it": achieved / max(stream["rates"].values()),
                    "rows_pattern": None if stream["rows_rate"] is None else {
                        "what": f"all {S} rows of the (S, ld) array read at one node index, then written back (non-temporal): the ionic kernel's access pattern",
                        "rate": stream["rows_rate"], "kernel_frac_of_it": achieved / stream["rows_rate"]},
                    "torch_mul_rate": stream["torch_mul_rate"],  # x.mul_(1.0): the yardstick of rounds 2-4
                },
                "whole_step": {
                    "bytes_per_node_update": 16.0 * S + 16.0 + 88.0 * k_avg,
                    "achieved": step_bytes * args.steps / wall / 1e9 / world,
                    "frac_of_8TBs_per_gpu": step_bytes * args.steps / wall / 1e9 / world / HBM_PEAK_GBS,
                    "frac_of_6.29TBs_per_gpu": step_bytes * args.steps / wall / 1e9 / world / HBM_COPY_GBS,
                },
            },
        }
        out["developed_front"] = front
        out["batched_solve"] = batched
        if ranks_info is not None:
            out["ranks"] = ranks_info
            out["config"]["comm"] = comm_info
            out["config"]["ordering"] = ordering_of(comm_info)  # of the transport `value` was measured on
            if comm_info is not None:
                out["config"]["rccl_ranks"] = comm_info["rccl_ranks"]
            if external_launch:
                out["config"]["launch"] = {"by": "an external launcher (no watchdog of bench.py's own)",
                                           "first_measurement": "rccl-serial: one communicator, one stream; the overlapped ordering "
                                                                "and the ipc transport are timed afterwards under a deadline (transports, transport_choice)"}
        if world == 1 and args.cpu_sample > 0:
            progress("CPU baseline (oracle port on the host cores)")
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.cpu_steps, args.rtol)
        else:
            out["cpu_baseline"] = None
    else:
        out = None

    # N > 1: the same steps over the OTHER transports of the library (ghost planes as interprocess device copies instead
    # of RCCL send/recv; everything on one communicator and one stream), a few steps each, so that one run on a multi-GPU
    # node says what each costs.  Best effort under a deadline: the headline above is already measured, and if an
    # alternative stops making progress every rank leaves (rank 0 printing the headline line first) instead of hanging.
    if world > 1 and finite and libcomm is not None and os.environ.get("BEAT_BENCH_ALT", "1") == "1":
        import threading

        from beat._engine import LibComm, LibCommUnavailable

        main_name = comm_info["transport"] if comm_info else "?"
        transports = {main_name: {"ms_per_step": wall / args.steps * 1e3, "pcg_iterations_per_step": float(np.mean(iters)), "role": "headline"}}
        deadline = float(os.environ.get("BEAT_BENCH_ALT_DEADLINE_S", "90"))

        # what give_up prints: a snapshot taken HERE, before the alternatives, so that the timer thread never serialises `out`
        # or `transports` while the main thread is updating them (adoption of a faster transport rewrites half of `out`)
        headline_snapshot = json.dumps(out) if rank == 0 else None

        def give_up(reason=None):
            """Leave with the headline that is already measured: rank 0 prints its line, every rank exits 0.  Called by the
            deadline timer, and by a rank on which an alternative raised (a transfer that timed out, a failed HIP call):
            the other ranks are then waiting for this one inside the alternative and leave at their own deadline."""
            if rank == 0:
                line = json.loads(headline_snapshot)
                line["transports"] = {main_name: dict(transports[main_name]),
                                      "error": reason or f"an alternative transport made no progress for {deadline:.0f} s; abandoned"}
                line["alt_failed"] = True
                if parity is not None:  # what was checked before the alternative that did not return
                    line["multi_rank_parity"] = dict(dict(parity), tolerance=bench_parity.TOLERANCE, incomplete=True,
                                                     ok=all(v.get("ok", False) for t in list(parity.values()) for v in t.values() if isinstance(v, dict)))
                print(json.dumps(line), file=result_stream, flush=True)
            print(f"[bench rank {rank}] alternative transports abandoned" + (f": {reason}" if reason else " at the deadline"),
                  file=sys.stderr, flush=True)
            os._exit(0)

        alts = [("ipc", False)] if main_name != "ipc" else []
        if backend == "nccl":
            alts += [("rccl", True)] if main_name != "rccl-serial" else [("rccl", False)]
        t_alt = (fr if front else run)["t"] + 10 * DT
        for name, serial in alts:
            label = name + ("-serial" if serial else "")
            timer = threading.Timer(deadline, give_up)
            timer.daemon = True
            timer.start()
            try:
                progress(f"alternative transport {label}")
                if os.environ.get("BEAT_BENCH_TEST_ALT_RAISE") == str(rank):  # tests: an alternative that fails on one rank only
                    raise RuntimeError("simulated failure inside an alternative transport (BEAT_BENCH_TEST_ALT_RAISE)")
                if parity is not None:  # the same correctness check on this transport, before it is timed
                    hk = parity_hook(name, serial)
                    verdicts = bench_parity.run_cases(dist, rank, world, hook=hk)
                    parity[label] = verdicts
                alt = LibComm(ctx, slab, dist, None, name, serial=serial, plane_doubles=plane)
                solver.libcomm = alt
                ar = timed_run(t_alt, 2, args.steps)
                t_alt = ar["t"]
                transports[label] = {"ms_per_step": ar["wall"] / args.steps * 1e3, "pcg_iterations_per_step": float(np.mean(ar["iters"])),
                                     "ode_ms": ar["ode_ms"], "pde_ms": ar["pde_ms"], "comm": alt.info()}
                ops.flush_pending()
                torch.cuda.synchronize()
                solver.libcomm = libcomm
                alt.close()
            except LibCommUnavailable as exc:  # raised on every rank alike
                transports[label] = {"error": str(exc)}
                solver.libcomm = libcomm
            except Exception as exc:  # noqa: BLE001 -- on this rank only: the ranks are out of step from here on
                transports[label] = {"error": repr(exc)}
                give_up(f"{label} raised on rank {rank}: {exc!r}")
            finally:
                timer.cancel()
        # The same steps on the headline's transport with ONE all-reduce per PCG iteration instead of two (PETSc's
        # -ksp_cg_single_reduction; csrc/beat_pde_rr.hip, beat_rr_udot_part): reported beside the transports, never adopted --
        # `value` stays the classic iteration's.  What it buys depends on what a reduction costs between real GPUs, which is
        # what this line is for.
        single = None
        if os.environ.get("BEAT_BENCH_SINGLE", "1") == "1" and hasattr(ops, "set_single_reduction"):
            timer = threading.Timer(deadline, give_up)
            timer.daemon = True
            timer.start()
            try:
                progress("single-reduction iteration")
                before = int(ctx.lib.beat_comm_merged_solves(libcomm.handle))
                ops.set_single_reduction(True)
                sr = timed_run(t_alt, 2, args.steps)
                t_alt = sr["t"]
                single = {"ms_per_step": sr["wall"] / args.steps * 1e3, "pcg_iterations_per_step": float(np.mean(sr["iters"])),
                          "ode_ms": sr["ode_ms"], "pde_ms": sr["pde_ms"], "transport": main_name,
                          "solves_on_the_single_reduction_iteration": int(ctx.lib.beat_comm_merged_solves(libcomm.handle)) - before,
                          "allreduces_per_solve": "k + 2 (classic: 2 k + 1)"}
                ops.flush_pending()
                torch.cuda.synchronize()
            except Exception as exc:  # noqa: BLE001
                ops.set_single_reduction(None)
                give_up(f"the single-reduction iteration raised on rank {rank}: {exc!r}")
            finally:
                ops.set_single_reduction(None)
                timer.cancel()
        # If an alternative beat the transport the headline was measured on by more than 3 % -- in the regime both were timed
        # in: the alternatives continue from the state of the last timed region -- the headline regime is measured again on
        # it (state re-initialised, same warm-up, same K steps, same barriers) and THAT becomes `value`: the line reports
        # the best this build does on this node, and says so.  Same deadline rule.
        ref_ms = (fr["wall"] if front else wall) / args.steps * 1e3
        timed = {k: v["ms_per_step"] for k, v in transports.items() if isinstance(v, dict) and "ms_per_step" in v and k != main_name}
        best = min(timed, key=timed.get) if timed else None
        choice = {"headline_measured_on": main_name, "compared_in": "developed_front" if front else "headline",
                  "reference_ms_per_step": ref_ms, "adopted": None}
        if best is not None and timed[best] < 0.97 * ref_ms and os.environ.get("BEAT_BENCH_ADOPT", "1") == "1":
            timer = threading.Timer(deadline, give_up)
            timer.daemon = True
            timer.start()
            try:
                progress(f"{best} was {100 * (1 - timed[best] / ref_ms):.0f} % faster: measuring the headline regime on it")
                name, serial = (best[:-7], True) if best.endswith("-serial") else (best, False)
                alt = LibComm(ctx, slab, dist, None, name, serial=serial, plane_doubles=plane)
                solver.libcomm = alt
                ops.flush_pending()
                init_states(ctx, states, ic, v_index, n, slab, 1234, nz_glob)
                ops.set_guess_order(args.guess_order)  # history and the adaptive order's scores as at the start of the first headline run
                solver.exchange_halo(v_field)
                hr = timed_run(0.0, args.warmup, args.steps)
                hmin, hmax = v_field.minmax()
                ext = torch.tensor([-hmin, hmax, 0.0 if np.isfinite(hmin) and np.isfinite(hmax) else 1.0], dtype=torch.float64,
                                   device=ctx.device if backend == "nccl" else "cpu")
                dist.all_reduce(ext, op=dist.ReduceOp.MAX)
                ok = float(ext[2]) == 0.0
                if ok and hr["wall"] < wall:
                    choice["adopted"] = best
                    if rank == 0:
                        transports[main_name]["role"] = "first measurement of the headline"
                        k_new = float(np.mean(hr["iters"]))
                        kp = float(np.mean(hr["pend_counts"])) if hr["pend_counts"] else 0.0
                        gb = 8.0 * float(np.mean(hr["guess_fields"])) if hr["guess_fields"] else 0.0
                        ob = (16.0 * len(ic) + 8.0 * kp + gb) * n_local
                        out.update(value=n * n * nz_glob * args.steps / hr["wall"], ms_per_step=hr["wall"] / args.steps * 1e3)
                        out["config"].update(pcg_iterations_per_step=k_new, ode_ms=hr["ode_ms"], pde_ms=hr["pde_ms"],
                                             v_min=-float(ext[0]), v_max=float(ext[1]), comm=alt.info(), ordering=ordering_of(alt.info()))
                        out["roofline"].update(achieved=ob / (hr["ode_ms"] * 1e-3) / 1e9, frac=ob / (hr["ode_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                               algorithmic_bytes_per_launch=ob, bytes_per_node=16.0 * len(ic) + 8.0 * kp + gb,
                                               pending_directions_per_launch=kp)
                        sb = (16.0 * len(ic) + 16.0 + 88.0 * k_new) * n * n * nz_glob
                        out["roofline"]["whole_step"].update(bytes_per_node_update=16.0 * len(ic) + 16.0 + 88.0 * k_new,
                                                             achieved=sb * args.steps / hr["wall"] / 1e9 / world,
                                                             frac_of_8TBs_per_gpu=sb * args.steps / hr["wall"] / 1e9 / world / HBM_PEAK_GBS)
                        out["roofline"]["whole_step"]["frac_of_6.29TBs_per_gpu"] = sb * args.steps / hr["wall"] / 1e9 / world / HBM_COPY_GBS
                        transports[best + " (headline)"] = {"ms_per_step": hr["wall"] / args.steps * 1e3, "pcg_iterations_per_step": k_new,
                                                           "ode_ms": hr["ode_ms"], "pde_ms": hr["pde_ms"], "role": "headline"}
                ops.flush_pending()
                torch.cuda.synchronize()
                solver.libcomm = libcomm
                alt.close()
            except LibCommUnavailable as exc:
                choice["error"] = str(exc)
                solver.libcomm = libcomm
            except Exception as exc:  # noqa: BLE001
                choice["error"] = repr(exc)
                out["config"]["transport_choice"] = choice
                give_up(f"re-measuring on {best} raised on rank {rank}: {exc!r}")
            finally:
                timer.cancel()
        if rank == 0:
            out["transports"] = transports
            out["config"]["transport_choice"] = choice
            if single is not None:
                single["reference_ms_per_step"] = ref_ms
                out["single_reduction"] = single
    parity_failed = parity is not None and any(not v.get("ok", False) for t in parity.values() for v in t.values() if isinstance(v, dict))
    if rank == 0:
        if parity is not None:
            out["multi_rank_parity"] = dict(parity, tolerance=bench_parity.TOLERANCE, ok=not parity_failed,
                                            what="decomposed on the N ranks against undivided on rank 0, public API, 25 split steps: "
                                                 "max |v_N - v_1| / max |v_1| per case and transport")
        print(json.dumps(out), file=result_stream, flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    if parity_failed:  # (the verdicts are broadcast: every rank sees the same)
        raise SystemExit(f"multi-rank parity failed: {json.dumps(parity)}")
    if not finite:
        raise SystemExit("non-finite membrane potential after the timed steps")  # every rank sees the same verdict


if __name__ == "__main__":
    main()
