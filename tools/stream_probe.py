#!/usr/bin/env python3
"""The HBM streaming ceiling measured with the library's OWN kernels (beat_stream_probe, csrc/beat_probe.hip) on an array
the size of the 512^3 TP06 state array (19 rows of 512^3 + 2 x 512^2 doubles, padded as StateArray pads them: 20.4 GB):

  in place / read only / write only / copy  x  global_load_dwordx4 | buffer_load_dwordx4  x  plain | nt loads | nt stores | both
  x  1 / 2 / 4 accesses in flight per lane  x  grid = 1, 2, 4, 8 x resident (2048 workgroups of 256) or one workgroup per chunk,

and the ionic kernels' own pattern: R rows of the (R, ld) array read at one index and written back (R = 1, 4, 8, 19, 45).
Beside them torch's x.mul_(1.0) (what bench.py's roofline.inplace_stream used until round 4) and the library's beat_copy.

    python tools/stream_probe.py [--gb 20.4] [--reps 5] [--json out.json]       -> markdown table on stdout
"""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "fenicsx-beat_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=20.4)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--json", default="")
    ap.add_argument("--quick", action="store_true", help="plain / both-nt only, unroll 1 and 4")
    args = ap.parse_args()
    import torch

    from beat import _hip
    from beat._device import Context, StateArray

    ctx = Context(0)
    lib = ctx.lib
    n_side = 512
    N, plane = n_side**3, n_side**2
    S = max(1, round(args.gb * 1e9 / (8.0 * (N + 2 * plane))))
    if args.gb < 20.0:  # a smaller array: rows of the size asked for
        S = 19
        N = int(args.gb * 1e9 / 8 / S) & ~255
        plane = 0
    states = StateArray(ctx, S, N, plane)
    ld = states.ld
    flat = states.buf[states.base: states.base + S * ld]
    nflat = flat.numel() & ~1
    flat.fill_(1.25)
    gbytes = nflat * 8 / 1e9
    print(f"array: {S} rows x ld {ld} doubles = {gbytes:.2f} GB (rows {N} long)", flush=True)

    def timed(fn, bytes_moved):
        fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in evs)
        return bytes_moved / ts[len(ts) // 2] / 1e6, ts[len(ts) // 2]  # GB/s (median), ms

    ptr = C.c_void_p(flat.data_ptr())
    results = []

    def probe(mode, policy, unroll, blocks, rows=0):
        def fn():
            if mode == 4:
                _hip.check(lib.beat_stream_probe(ctx.handle, states.ptr, N, 4, policy, 1, blocks, rows, ld))
            else:
                _hip.check(lib.beat_stream_probe(ctx.handle, ptr, nflat, mode, policy, unroll, blocks, 0, 0))
        moved = {0: 2.0, 1: 1.0, 2: 1.0, 3: 1.0}.get(mode, 0.0) * nflat * 8 if mode != 4 else 2.0 * rows * N * 8
        gbs, ms = timed(fn, moved)
        results.append({"mode": mode, "policy": policy, "unroll": unroll, "blocks": blocks, "rows": rows, "GBps": gbs, "ms": ms})
        return gbs

    mode_names = {0: "in place", 1: "read only", 2: "write only", 3: "copy (half -> half)"}
    pol_names = {0: "plain", 1: "nt loads", 2: "nt stores", 3: "nt both", 4: "buffer", 5: "buffer nt loads", 6: "buffer nt stores", 7: "buffer nt both"}
    policies = (0, 3, 4, 7) if args.quick else tuple(range(8))
    unrolls = (1, 4) if args.quick else (1, 2, 4)
    grids = (2048, 4096, 8192, 16384, 0)
    print("\n| mode | instructions / policy | in flight | " + " | ".join(f"{g} WGs" if g else "1 WG / chunk" for g in grids) + " |")
    print("|---|---|---:|" + "---:|" * len(grids))
    best = {}
    for mode in (0, 1, 2, 3):
        for pol in policies:
            if mode == 1 and (pol & 2):
                continue
            if mode == 2 and (pol & 1):
                continue
            for u in unrolls:
                row = []
                for g in grids:
                    gb = probe(mode, pol, u, g)
                    row.append(gb)
                    if gb > best.get(mode, (0,))[0]:
                        best[mode] = (gb, pol, u, g)
                print(f"| {mode_names[mode]} | {pol_names[pol]} | {u} | " + " | ".join(f"{v:.0f}" for v in row) + " |", flush=True)
        if mode == 2:
            flat.fill_(1.25)
    print("\n| rows of the (R, ld) array in place (all R loads of an index, then R stores) | policy | " + " | ".join(f"{g} WGs" if g else "1 WG / chunk" for g in (2048, 8192, 24576, 0)) + " |")
    print("|---|---|" + "---:|" * 4)
    for rows in (1, 4, 8, 19, 45):
        if rows > S:
            continue
        for pol in (0, 1, 2, 3):
            row = [probe(4, pol, 1, g, rows) for g in (2048, 8192, 24576, 0)]
            print(f"| R = {rows} | {pol_names[pol]} | " + " | ".join(f"{v:.0f}" for v in row) + " |", flush=True)
    # the two yardsticks used so far
    t_mul, _ = timed(lambda: flat.mul_(1.0), 2.0 * nflat * 8)
    half = nflat // 2
    dst, src = flat[half:2 * half], flat[:half]
    t_copy_torch, _ = timed(lambda: dst.copy_(src), 2.0 * half * 8)
    t_beat_copy, _ = timed(lambda: _hip.check(lib.beat_copy(ctx.handle, C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), half)), 2.0 * half * 8)
    print(f"\ntorch x.mul_(1.0) in place: {t_mul:.0f} GB/s; torch copy_ half -> half (read + write counted): {t_copy_torch:.0f} GB/s; "
          f"beat_copy (2048 WGs, 16 B per lane and trip): {t_beat_copy:.0f} GB/s")
    print("best per mode (GB/s, policy, in flight, grid): " + "; ".join(
        f"{mode_names[m]} {b[0]:.0f} ({pol_names[b[1]]}, {b[2]}, {b[3] or '1 WG / chunk'})" for m, b in sorted(best.items())))
    print("(copy: bytes read + bytes written)")
    if args.json:
        Path(args.json).write_text(json.dumps({"array_GB": gbytes, "results": results, "torch_mul_GBps": t_mul,
                                               "torch_copy_rw_GBps": t_copy_torch, "beat_copy_rw_GBps": t_beat_copy}, indent=1))


if __name__ == "__main__":
    main()
