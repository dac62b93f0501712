#!/bin/bash
# How far apart are the ionic kernel's memory time, its arithmetic time and the time it takes?  Builds two probe
# libraries next to the shipped one (-DBEAT_ODE_PROBE=1: every state read and written back, nothing computed; =2: the
# arithmetic on cache-resident states, nothing stored), runs the plain TP06 step at 512^3 with each, removes them.
# Run on the GPU box from the repo root:  bash tools/ode_probe.sh
set -e
R=$PWD
C=$R/fenicsx-beat_amd/csrc
L=$R/fenicsx-beat_amd/beat/lib
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-function -ffp-contract=off -DBEAT_ODE_WAVES=3 -mllvm -disable-machine-licm"
for m in 1 2; do
  /opt/rocm/bin/hipcc $FL -DBEAT_ODE_PROBE=$m -c $C/beat_ode.hip -o /tmp/beat_ode_probe$m.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $L/libbeat_probe$m.so $C/build/beat_api.o /tmp/beat_ode_probe$m.o $C/build/beat_pde.o \
    $C/build/beat_pde_var.o $C/build/beat_pde_rr.o $C/build/beat_pde_small.o $C/build/beat_dist.o -ldl
done
for l in libbeat_hip libbeat_probe1 libbeat_probe2 libbeat_hip libbeat_probe1 libbeat_probe2; do
  echo -n "$l: "
  BEAT_HIP_LIBRARY=$L/$l.so python tools/bench_kernels.py --n 512 --only "ode_step tp06" 2>&1 | grep ode_step
done
rm -f $L/libbeat_probe1.so $L/libbeat_probe2.so
