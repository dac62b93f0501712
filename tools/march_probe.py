#!/usr/bin/env python3
"""The register-row kernels' access pattern without their arithmetic (beat_stream_probe mode 5): does a wave's misaligned, overlapping
62-node segment (lanes 0 and 63 carry the x-halo) cost bandwidth against whole aligned 64-node pieces?   python tools/march_probe.py [--n 512]"""
import argparse
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "fenicsx-beat_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    args = ap.parse_args()
    import torch

    from beat import _hip
    from beat._device import Context

    ctx = Context.default()
    n = args.n
    buf = torch.rand(2 * n**3, dtype=torch.float64, device=ctx.device)
    lib = ctx.lib
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for ry in (4, 2):
        for segw, shift, tag in ((62, 1, "62 nodes per wave, lane shift -1 (the kernels)"), (64, 0, "64 nodes per wave, aligned"),
                                 (64, 2, "64 aligned + one 12-lane x-halo load per plane")):
            for blocks in (4096, 8192):
                ts = []
                for rep in range(6):
                    ev[0].record()
                    _hip.check(lib.beat_stream_probe(ctx.handle, C.c_void_p(buf.data_ptr()), buf.numel(), 5, shift, ry, blocks, segw, n))
                    ev[1].record()
                    torch.cuda.synchronize()
                    ts.append(ev[0].elapsed_time(ev[1]))
                ms = sorted(ts[1:])[len(ts[1:]) // 2]
                print(f"rows per wave {ry}, {tag:48s} blocks ~{blocks}: {ms * 1e3:7.1f} us = {16.0 * n**3 / ms / 1e9:5.2f} TB/s of 16 B/node", flush=True)


if __name__ == "__main__":
    main()
