#!/usr/bin/env python3
"""What streaming rates does this box reach?  Read-only, write-only and 1:1 copy over arrays of the size the ionic kernel
walks (19 x 512^3 doubles = 20.4 GB), with PyTorch's kernels, hipMemcpy (through torch) and the library's own copy --
the practical ceilings next to which the 8 TB/s of the roofline should be read."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "fenicsx-beat_amd"))


def main():
    import torch

    from beat import _hip
    from beat._device import Context

    ctx = Context(0)
    lib = ctx.lib
    dev = ctx.device

    def timeit(name, fn, nbytes, reps=8):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in ev)
        med = ts[len(ts) // 2]
        print(f"{name:44s} {med:8.3f} ms  {nbytes / med / 1e9:7.2f} TB/s  (best {nbytes / ts[0] / 1e9:.2f})", flush=True)

    for n in (2**27, 19 * 2**27):
        gb = n * 8 / 1e9
        a = torch.empty(n, dtype=torch.float64, device=dev).normal_()
        b = torch.empty_like(a)
        print(f"--- {gb:.1f} GB per array")
        timeit("torch fill_ (write only)", lambda: b.fill_(1.0), n * 8)
        timeit("torch sum (read only)", lambda: a.sum(), n * 8)
        timeit("torch copy_ (read + write)", lambda: b.copy_(a), 2 * n * 8)
        timeit("torch add_ scalar (read + write in place)", lambda: b.add_(1.0), 2 * n * 8)
        timeit("torch mul out (2 reads + write)", lambda: torch.mul(a, b, out=b), 3 * n * 8)
        pa, pb = C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr())
        timeit("beat_copy (library copy kernel)", lambda: _hip.check(lib.beat_copy(ctx.handle, pb, pa, n)), 2 * n * 8)
        del a, b
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
