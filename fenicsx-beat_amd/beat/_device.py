"""Device plumbing: one HIP context per process, fields as torch CUDA(HIP) tensors.

PyTorch is used for device memory, the stream and ``torch.distributed`` only; all arithmetic
goes through ``libbeat_hip.so``.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _hip


class Context:
    """Wraps ``beat_ctx``: device ordinal + the torch current stream of that device."""

    _default = None

    def __init__(self, device: int | None = None):
        import torch

        if not torch.cuda.is_available():
            raise _hip.BeatHipError(
                "no MI355X/HIP device visible to PyTorch; the beat HIP backend has no CPU fallback"
            )
        self.torch = torch
        self.lib = _hip.load()
        if device is None:
            device = torch.cuda.current_device()
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.current_stream(self.device)
        handle = C.c_void_p()
        _hip.check(self.lib.beat_ctx_create(self.device_index, C.c_void_p(self.stream.cuda_stream), C.byref(handle)))
        self.handle = handle

    @classmethod
    def default(cls) -> "Context":
        if cls._default is None:
            cls._default = cls()
        return cls._default

    def synchronize(self):
        _hip.check(self.lib.beat_ctx_synchronize(self.handle))

    # ------------------------------------------------------------------ allocation
    def zeros(self, n: int, dtype=None):
        torch = self.torch
        return torch.zeros(int(n), dtype=dtype or torch.float64, device=self.device)

    def from_numpy(self, a: np.ndarray):
        torch = self.torch
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.device)

    def field(self, n: int, plane: int) -> "Field":
        return Field(self, n, plane)

    def __del__(self):  # pragma: no cover
        try:
            self.lib.beat_ctx_destroy(self.handle)
        except Exception:
            pass


class Field:
    """An N-vector with one ghost xy-plane on either side (see include/beat_hip.h)."""

    def __init__(self, ctx: Context, n: int, plane: int, buf=None, offset: int | None = None):
        self.ctx = ctx
        self.n = int(n)
        self.plane = int(plane)
        if buf is None:
            buf = ctx.zeros(self.n + 2 * self.plane)
            offset = self.plane
        self.buf = buf
        self.offset = int(offset)
        self.data = buf[self.offset : self.offset + self.n]  # interior view (torch)

    @property
    def ptr(self) -> C.c_void_p:
        return C.c_void_p(self.data.data_ptr())

    @property
    def ghost_lo(self):
        return self.buf[self.offset - self.plane : self.offset]

    @property
    def ghost_hi(self):
        return self.buf[self.offset + self.n : self.offset + self.n + self.plane]

    def numpy(self) -> np.ndarray:
        return self.data.cpu().numpy()

    def set(self, values) -> None:
        torch = self.ctx.torch
        arr = np.ascontiguousarray(np.broadcast_to(np.asarray(values, dtype=np.float64), (self.n,)))
        if not arr.flags.writeable:  # torch.from_numpy wants a writable buffer
            arr = arr.copy()
        self.data.copy_(torch.from_numpy(arr))

    def copy_from(self, other: "Field") -> None:
        _hip.check(self.ctx.lib.beat_copy(self.ctx.handle, self.ptr, other.ptr, self.n))

    def fill(self, value: float) -> None:
        _hip.check(self.ctx.lib.beat_fill(self.ctx.handle, self.ptr, float(value), self.n))

    def minmax(self) -> tuple[float, float]:
        lo, hi = C.c_double(), C.c_double()
        _hip.check(self.ctx.lib.beat_field_minmax(self.ctx.handle, self.ptr, self.n, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value


class StateArray:
    """(S, N) state-major array; every row is a Field-compatible N-vector (ghost planes on
    both sides) so that any row can be handed to the PDE kernels without a copy."""

    def __init__(self, ctx: Context, num_states: int, n: int, plane: int = 0):
        self.ctx = ctx
        self.S = int(num_states)
        self.n = int(n)
        self.plane = int(plane)
        ld = self.n + 2 * self.plane
        self.ld = (ld + 31) // 32 * 32  # 256-byte aligned rows
        # Rows a multiple of 4 KiB apart (512^3 + 2 * 512^2 doubles are 2^30 + 2^22 bytes) put node i of EVERY row on the
        # same memory channels: the ionic kernel's 19 .. 52 row streams then queue on a few channels at a time.  17 * 256 B
        # of padding per row spreads them: the kernel's memory traffic alone (probe build, 256^3) 4.5 -> 5.1 TB/s for 45
        # rows, 4.8 -> 5.3 for 19 (profiles/r03_ode_probes.md).  BEAT_STATE_SKEW=<doubles> overrides (0: none).
        skew = os.environ.get("BEAT_STATE_SKEW", "auto")
        if skew == "auto":
            skew = 544 if (self.ld * 8) % 4096 == 0 and self.n >= 65536 else 0
        self.ld += int(skew) // 32 * 32
        self.base = self.plane  # leading ghost plane of row 0
        self.placement = None  # what _place_buffer found (large arrays)
        self.buf = self._place_buffer(self.base + self.S * self.ld + 32)
        self.rows = self.buf[self.base : self.base + self.S * self.ld].view(self.S, self.ld)[:, : self.n]

    # Round 6: WHERE the driver puts a large state array decides how fast its rows stream.  The S row streams of the ionic kernels
    # (every row read at one node index, written back) ran at 5.1 - 6.3 TB/s on 19-row arrays allocated one after another in ONE
    # process (tools/place_probe3.py, profiles/r06_placement.md: the rate belongs to the allocation -- measured again later it is
    # the same --, the first large allocation of a process was the slowest every time), and consecutive bench processes alternated
    # between two levels of the 512^3 step ~0.4 ms apart that rounds 3 - 5 chased inside the kernel.  So a large array is allocated
    # up to BEAT_STATE_PLACE (default 3; 0 / 1: off) times, each candidate is timed with the library's own streaming probe of that
    # very pattern (beat_stream_probe mode 4: a few launches, ~10 ms each at 20 GB), the best one is kept and the others go back to
    # the driver.  Only when the device has room for the candidates side by side; what was found is in ``self.placement``.
    def _place_buffer(self, numel: int):
        ctx = self.ctx
        torch = ctx.torch
        tries = int(os.environ.get("BEAT_STATE_PLACE", "3"))
        nbytes = 8 * int(numel)
        if tries <= 1 or nbytes < int(os.environ.get("BEAT_STATE_PLACE_MIN_BYTES", str(1 << 30))):  # (the tests lower the threshold)
            return ctx.zeros(numel)
        rows = max(r for r in (1, 4, 8, 19, 45) if r <= self.S)  # (the probe's instances)
        lib = ctx.lib

        def rate(buf) -> float:
            ptr = C.c_void_p(buf.data_ptr() + 8 * self.base)
            times = []
            for it in range(4):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(ctx.stream)
                _hip.check(lib.beat_stream_probe(ctx.handle, ptr, self.n & ~1, 4, 3, 1, 0, rows, self.ld))
                b.record(ctx.stream)
                b.synchronize()
                if it:  # (the first launch loads the kernel)
                    times.append(a.elapsed_time(b))
            return 2.0 * rows * (self.n & ~1) * 8 / (sorted(times)[len(times) // 2] * 1e6)  # GB/s (zeros x 1.0 stay zeros)

        cands, rates = [], []
        for _ in range(tries):
            free_b, _total = torch.cuda.mem_get_info(ctx.device)
            if cands and free_b < 2 * nbytes:  # keep room for what the caller allocates next
                break
            try:
                buf = ctx.zeros(numel)
            except RuntimeError:  # out of memory with the candidates held: what we have is what we choose from
                if not cands:
                    raise
                break
            cands.append(buf)
            try:
                rates.append(rate(buf))
            except _hip.BeatHipError:  # a layout the probe does not take (odd plane: rows not 16-byte aligned): no selection
                self.placement = None
                return cands[0]
        best = max(range(len(cands)), key=lambda j: rates[j])
        buf = cands[best]
        self.placement = {"candidates": [round(r, 1) for r in rates], "chosen": best, "rows_probed": rows, "unit": "GB/s"}
        if len(cands) > 1:
            del cands
            torch.cuda.empty_cache()  # the losers go back to the driver, not into torch's cache (the next allocation would get one)
        return buf

    @property
    def ptr(self) -> C.c_void_p:
        return C.c_void_p(self.buf.data_ptr() + 8 * self.base)

    def row_field(self, k: int) -> Field:
        return Field(self.ctx, self.n, self.plane, buf=self.buf, offset=self.base + k * self.ld)

    def numpy(self) -> np.ndarray:
        return self.rows.cpu().numpy()

    def set(self, values: np.ndarray) -> None:
        torch = self.ctx.torch
        values = np.ascontiguousarray(np.asarray(values, dtype=np.float64))
        assert values.shape == (self.S, self.n), (values.shape, (self.S, self.n))
        self.rows.copy_(torch.from_numpy(values))
