#!/bin/bash
# Instrumented builds of the frame kernel for same-process A/B runs (tools/ab_sweep.py libA libB, tools/clock_probe.sh):
#   tools/exp/libexp1.so  PM_EXPERIMENT=1: waves without an intercept skip their NaN stores
#   tools/exp/libexp2.so  PM_EXPERIMENT=2: no plane is stored at all (compute only)
# Never the shipped library; tools/exp/ is git-ignored like every built artefact.
set -e
cd "$(dirname "$0")/../planetmapper_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../tools/exp
for e in 1 2; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DPM_EXPERIMENT=$e -c pm_kernels.hip -o /tmp/pm_kernels_e$e.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/libexp$e.so /tmp/pm_kernels_e$e.o pm_kernels_reproject.o pm_capi.o pm_reproject.o pm_hostpipe.o pm_comm.o -lpthread -ldl
done
ls -la ../../tools/exp
