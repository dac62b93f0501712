#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel of one translation unit of libbeat_hip, from
``hipcc -Rpass-analysis=kernel-resource-usage`` with the flags the Makefile uses (cross-compiles without a GPU).

    python3 tools/kernel_resources.py beat_ode.hip [--filter ode_step_kernel] [--md out.md]
"""
import argparse
import re
import shutil
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "fenicsx-beat_amd" / "csrc"
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -ffp-contract=off -DBEAT_ODE_WAVES=3".split()
EXTRA = {"beat_ode.hip": ["-mllvm", "-disable-machine-licm"]}


def demangle(names):
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not tool:
        return names
    out = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "")).replace("void ", "") for n in out]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("--filter", default="")
    ap.add_argument("--md", default=None)
    args = ap.parse_args()
    res = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *EXTRA.get(args.source, []), "-Rpass-analysis=kernel-resource-usage", "-c",
                          str(CSRC / args.source), "-o", "/dev/null"], capture_output=True, text=True)
    blocks = re.split(r"remark: [^\n]*Function Name: ", res.stderr)[1:]
    rows = []
    for b in blocks:
        def g(key):
            m = re.search(key + r": (\d+)", b)
            return int(m.group(1)) if m else -1
        rows.append([b.split("\n")[0].split(" [-R")[0].strip(), g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
                     g("TotalSGPRs"), g("SGPRs Spill"), g(r"LDS Size \[bytes/block\]")])
    for r, n in zip(rows, demangle([r[0] for r in rows])):
        r[0] = n
    rows = [r for r in rows if args.filter in r[0]]
    lines = ["| kernel | VGPRs | AGPRs | scratch B/lane | waves/SIMD | SGPRs | spilled SGPRs | LDS B/block |", "|---|---:|---:|---:|---:|---:|---:|---:|"]
    lines += ["| `" + r[0] + "` | " + " | ".join(str(v) for v in r[1:]) + " |" for r in rows]
    text = "\n".join(lines)
    print(text)
    if args.md:
        Path(args.md).write_text(f"# kernel resource usage of {args.source} (hipcc -Rpass-analysis=kernel-resource-usage, gfx950)\n\n" + text + "\n")


if __name__ == "__main__":
    main()
